"""Drop-in `models.FC_Encoder.FC_Encoder` (reference models/FC_Encoder.py:12-89)."""
import torch.nn as nn

from ._loading import load_matching
from .layers import ConvLayer, DownsampleConvLayer


class FC_Encoder(nn.Module):
    def __init__(self, num_layers=2, image_channel=3, min_channel=128, max_channel=512,
                 latent_channel=2, hidden_activation_type="silu", use_final_activation=True,
                 final_activation_type="tanh"):
        super().__init__()
        ch = min_channel
        layers = [ConvLayer(in_channels=image_channel, out_channels=ch, use_activation=True,
                            activation_type=hidden_activation_type)]
        for _ in range(num_layers):
            layers.append(ConvLayer(in_channels=ch, out_channels=ch, use_activation=True,
                                    activation_type=hidden_activation_type))
            nxt = ch * 2 if ch * 2 < max_channel else max_channel   # FC_Encoder.py:44-45
            layers.append(DownsampleConvLayer(in_channels=ch, out_channels=nxt,
                                              activation_type=hidden_activation_type))
            ch = nxt
        layers.append(ConvLayer(in_channels=ch, out_channels=latent_channel,
                                use_activation=use_final_activation,
                                activation_type=final_activation_type))
        self.fc_encoder_layer = nn.ModuleList(layers)

    def custom_load_state_dict(self, state_dict, ignore_msgs=False):
        # accepts Autoencoder checkpoints: "fc_encoder.fc_encoder_layer" -> "fc_encoder_layer"
        load_matching(self, state_dict, rename=("fc_encoder.fc_encoder_layer", "fc_encoder_layer"),
                      ignore_msgs=ignore_msgs)

    def forward(self, x):
        for layer in self.fc_encoder_layer:
            x = layer(x)
        return x
