"""Unimodal VAE modules with the reference's names / parameters / return conventions (pairedVAE.py:24-187).
Encoder FCNN (sparse first layer), the mu / sigma heads, the fused reparameterisation + KL kernel and the decoder
GEMMs run on the HIP kernels; the decoder's small softmax / softplus heads are torch tensor ops on the GPU."""
from __future__ import annotations

import numpy as np
import torch
from torch import nn
import torch.nn.functional as F

from . import ops
from .utils import FCNN, HipLinear


class VAE_ENCODER(nn.Module):
    """pairedVAE.py:24-74.  forward -> (mu, sigma, z, latent_lib); z is sampled in eval mode too, as upstream.
    The KL term of utils.py:186 is produced by the same kernel and kept in `self.last_kl`."""

    def __init__(self, input_dim: int, input_batch_num: int, hidden_dim, latent_dim: int, layernorm=True,
                 activation=nn.LeakyReLU(), batchnorm=False, dropout_rate=0, infer_library_size=False):
        super().__init__()
        self.infer_library_size = infer_library_size
        self.enc_hidden = FCNN([input_dim + input_batch_num] + hidden_dim, layernorm=layernorm, activation=activation,
                               batchnorm=batchnorm, dropout_rate=dropout_rate)
        if self.infer_library_size == True:  # noqa: E712
            self.enc_lib = nn.Sequential(FCNN([input_dim + input_batch_num] + hidden_dim, layernorm=layernorm,
                                              activation=activation, batchnorm=batchnorm, dropout_rate=dropout_rate),
                                         HipLinear(hidden_dim[-1], 1, bias=True))
        self.latent_mu = HipLinear(hidden_dim[-1], latent_dim, bias=True)
        self.latent_sigma = HipLinear(hidden_dim[-1], latent_dim, bias=True)
        self.last_kl = None
        self.eps_source = None          # optional callable(shape, device) -> noise (replay in parity tests)

    def forward(self, x, x_batch):
        enc_h = self.enc_hidden(x, x_batch)
        mu = self.latent_mu(enc_h)
        sraw = self.latent_sigma(enc_h)
        eps = self.eps_source(sraw.shape, sraw.device) if self.eps_source is not None else torch.randn_like(sraw)
        sigma, z, kl = ops.reparam_kl(mu, sraw, eps)
        self.last_kl = kl
        if self.infer_library_size == True:  # noqa: E712
            latent_lib = F.softplus(self.enc_lib[1](self.enc_lib[0](x, x_batch)))
        else:
            latent_lib = None
        return mu, sigma, z, latent_lib


class VAE_DECODER(nn.Module):
    """pairedVAE.py:78-187."""

    def __init__(self, input_dim: int, input_batch_num: int, hidden_dim, latent_dim: int, layernorm=True,
                 activation=nn.LeakyReLU(), batchnorm=False, dropout_rate=0, infer_library_size=False,
                 sample_factor=True, feature_factor=False, distribution="zinb", dispersion="feature-cell"):
        super().__init__()
        hidden_dim_decoder = list(np.flip(hidden_dim))
        self.feature_factor = feature_factor
        self.infer_library_size = infer_library_size
        self.distribution = distribution
        self.dispersion = dispersion
        self.sample_factor = sample_factor
        if self.feature_factor == True:  # noqa: E712
            self.feature_scalar = torch.nn.Parameter(torch.zeros(input_dim))
        lib_dim = 1 if self.infer_library_size == True else 0  # noqa: E712
        self.dec_hidden = FCNN([latent_dim + input_batch_num + lib_dim] + [int(h) for h in hidden_dim_decoder],
                               layernorm=layernorm, activation=activation, batchnorm=batchnorm,
                               dropout_rate=dropout_rate)
        self.output = HipLinear(int(hidden_dim_decoder[-1]), input_dim, bias=True)
        if self.distribution in ("zinb", "nb"):
            self.scale_function = nn.Softmax(dim=-1)
            if self.dispersion == "feature":
                self.output_inverse_dispersion = torch.nn.Parameter(torch.randn(input_dim))
            elif self.dispersion == "feature-cell":
                self.output_inverse_dispersion = HipLinear(int(hidden_dim_decoder[-1]), input_dim, bias=True)
            if self.distribution == "zinb":
                self.output_dropout = HipLinear(int(hidden_dim_decoder[-1]), input_dim, bias=True)
        self.transformation_function = nn.LeakyReLU()

    def reconstruction(self, dec_h):
        return self.output(dec_h)

    def fused_loss(self, z, x_batch, lib, latent_lib, labels, likelihood):
        """raw_loss(labels, *forward(...)) (utils.py:187-190) for the count likelihoods with dispersion
        'feature-cell' and no feature factor: the three head GEMMs, then softmax / softplus / (ZI)NB likelihood and
        their backward in the fused row kernels.  None for the other combinations (generic path)."""
        if likelihood not in ("nb", "zinb") or self.distribution not in ("nb", "zinb") or \
                self.dispersion != "feature-cell" or self.feature_factor == True:  # noqa: E712
            return None
        if likelihood == "zinb" and self.distribution != "zinb":
            return None
        if x_batch is not None:
            z = torch.cat([z, x_batch], dim=1)
        if latent_lib is not None:
            z = torch.cat([z, latent_lib], dim=1)
        dec_h = self.dec_hidden(z)
        out = self.output(dec_h)
        tz = self.output_inverse_dispersion(dec_h)
        pi = self.output_dropout(dec_h) if likelihood == "zinb" else None
        if self.sample_factor == True:  # noqa: E712
            assert lib is not None
            L = lib
        elif latent_lib is not None:
            L = torch.exp(self.transformation_function(latent_lib))
        else:
            L = torch.ones(out.shape[0], dtype=out.dtype, device=out.device)
        return ops.raw_rows_loss(out, pi, L, labels.to_dense() if hasattr(labels, "to_dense") else labels, tz)

    def forward(self, z, x_batch, lib, latent_lib):
        if x_batch is not None:
            z = torch.cat([z, x_batch], dim=1)
        if latent_lib is not None:
            z = torch.cat([z, latent_lib], dim=1)
        dec_h = self.dec_hidden(z)
        output_temp = self.reconstruction(dec_h)
        if self.distribution == 'gau':
            return output_temp * self.feature_scalar if self.feature_factor == True else output_temp  # noqa: E712
        if self.distribution == 'ber':
            return torch.sigmoid(output_temp * self.feature_scalar if self.feature_factor == True else output_temp)  # noqa: E712
        px_pi = self.output_dropout(dec_h) if self.distribution == 'zinb' else None
        if self.feature_factor == True:  # noqa: E712
            px_mu_scale = self.scale_function(output_temp * torch.sigmoid(self.feature_scalar))
        else:
            px_mu_scale = self.scale_function(output_temp)
        if self.sample_factor == True:  # noqa: E712
            assert lib is not None
            px_mu_rate = lib * px_mu_scale
        elif latent_lib is not None:
            px_mu_rate = torch.exp(self.transformation_function(latent_lib)) * px_mu_scale
        else:
            px_mu_rate = px_mu_scale
        if self.dispersion == "feature":
            px_theta = torch.exp(self.output_inverse_dispersion)
        else:
            px_theta = F.softplus(self.output_inverse_dispersion(dec_h)) + 1e-8
        return px_mu_scale, px_theta, px_mu_rate, px_pi
