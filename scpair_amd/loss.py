"""Objective functions with the reference's names and signatures (loss.py:22-51), on HIP kernels."""
from __future__ import annotations

from . import ops
from .scVI_distribution import NegativeBinomial, ZeroInflatedNegativeBinomial


def scaled_loss(x, mu):
    """loss.py:22-27: mean squared error over the non-NaN entries of x."""
    return ops.mse_masked(x, mu)


def binarized_loss(x, p):
    """loss.py:36-41: BCE (log clamp -100) on probabilities over the non-NaN entries of x."""
    return ops.bce_masked(x, p)


def raw_loss(x, px_mu_rate, px_theta, px_mu_scale, px_pi, likelihood="zinb"):
    """loss.py:46-51."""
    if likelihood == "nb":
        return -NegativeBinomial(mu=px_mu_rate, theta=px_theta, scale=px_mu_scale).log_prob(x).mean()
    elif likelihood == "zinb":
        return -ZeroInflatedNegativeBinomial(mu=px_mu_rate, theta=px_theta, scale=px_mu_scale,
                                             zi_logits=px_pi).log_prob(x).mean()
    raise ValueError("likelihood must be 'nb' or 'zinb'")
