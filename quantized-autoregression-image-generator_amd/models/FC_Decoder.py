"""Drop-in `models.FC_Decoder.FC_Decoder` (reference models/FC_Decoder.py:12-96)."""
import torch.nn as nn

from ._loading import load_matching
from .layers import ConvLayer, UpsampleConvLayer


class FC_Decoder(nn.Module):
    def __init__(self, num_layers=2, image_channel=3, min_channel=128, max_channel=512,
                 latent_channel=2, hidden_activation_type="silu", use_final_activation=True,
                 final_activation_type="tanh"):
        super().__init__()
        ch = max_channel
        layers = [nn.Sequential(
            ConvLayer(in_channels=latent_channel, out_channels=ch, use_activation=True,
                      activation_type=hidden_activation_type),
            ConvLayer(in_channels=ch, out_channels=ch, use_activation=True,
                      activation_type=hidden_activation_type))]
        for _ in range(num_layers):
            layers.append(ConvLayer(in_channels=ch, out_channels=ch, use_activation=True,
                                    activation_type=hidden_activation_type))
            nxt = ch // 2 if ch // 2 > min_channel else min_channel   # FC_Decoder.py:50-51
            layers.append(UpsampleConvLayer(in_channels=ch, out_channels=nxt,
                                            activation_type=hidden_activation_type))
            ch = nxt
        layers.append(ConvLayer(in_channels=ch, out_channels=image_channel,
                                use_activation=use_final_activation,
                                activation_type=final_activation_type))
        self.fc_decoder_layer = nn.ModuleList(layers)

    def custom_load_state_dict(self, state_dict):
        # accepts Autoencoder checkpoints; anything without "decoder" in its name is
        # skipped (FC_Decoder.py:75-77)
        load_matching(self, state_dict, rename=("fc_decoder.fc_decoder_layer", "fc_decoder_layer"),
                      must_contain="decoder")

    def forward(self, x):
        for layer in self.fc_decoder_layer:
            x = layer(x)
        return x
