"""Drop-in `models.Autoencoder.Autoencoder` (reference models/Autoencoder.py:11-74)."""
import torch.nn as nn

from ._loading import load_matching
from .FC_Decoder import FC_Decoder
from .FC_Encoder import FC_Encoder


class Autoencoder(nn.Module):
    def __init__(self, num_layers=2, image_channel=3, min_channel=128, max_channel=512,
                 latent_channel=2, hidden_activation_type="silu", use_final_enc_activation=True,
                 encoder_activation_type="silu", use_final_dec_activation=True,
                 decoder_activation_type="tanh"):
        super().__init__()
        self.fc_encoder = FC_Encoder(
            num_layers=num_layers, image_channel=image_channel, min_channel=min_channel,
            max_channel=max_channel, latent_channel=latent_channel,
            hidden_activation_type=hidden_activation_type,
            use_final_activation=use_final_enc_activation,
            final_activation_type=encoder_activation_type)
        self.fc_decoder = FC_Decoder(
            num_layers=num_layers, image_channel=image_channel, min_channel=min_channel,
            max_channel=max_channel, latent_channel=latent_channel,
            hidden_activation_type=hidden_activation_type,
            use_final_activation=use_final_dec_activation,
            final_activation_type=decoder_activation_type)

    def custom_load_state_dict(self, state_dict):
        load_matching(self, state_dict)

    def get_latent(self, x):
        return self.fc_encoder(x)

    def recon_image(self, z):
        return self.fc_decoder(z)

    def forward(self, x):
        return self.recon_image(self.get_latent(x))
