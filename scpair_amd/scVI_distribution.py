"""NB / ZINB likelihoods with the reference's names and call forms (scVI_distribution.py:29-105, 251-433),
evaluated by the HIP kernels.  Only what the reference's hot path uses is provided: log_nb_positive,
log_zinb_positive and the two Distribution-like classes' log_prob / mean (loss.py:48, 50)."""
from __future__ import annotations

import torch

from . import ops


def log_zinb_positive(x, mu, theta, pi, eps=1e-8):
    """scVI_distribution.py:29-71.  eps is fixed at the reference's 1e-8 inside the kernel."""
    if eps != 1e-8:
        raise ValueError("the HIP likelihood kernels implement the reference's eps = 1e-8 only")
    if theta.ndimension() == 1:
        theta = theta.view(1, theta.size(0))          # scVI_distribution.py:51-54
    return ops.nb_loglik(x, mu, theta, pi)


def log_nb_positive(x, mu, theta, eps=1e-8, log_fn=None, lgamma_fn=None):
    """scVI_distribution.py:74-105."""
    if eps != 1e-8 or log_fn is not None or lgamma_fn is not None:
        raise ValueError("the HIP likelihood kernels implement the reference's eps = 1e-8 / torch.log / torch.lgamma only")
    return ops.nb_loglik(x, mu, theta, None)


class NegativeBinomial:
    """(mu, theta) parameterisation only -- the one loss.py:48 uses (scVI_distribution.py:251-342)."""

    def __init__(self, total_count=None, probs=None, logits=None, mu=None, theta=None, scale=None, validate_args=False):
        if mu is None or theta is None or total_count is not None:
            raise ValueError("Please use the (mu, theta) parameterization (the only one the scPair hot path uses).")
        self._eps = 1e-8
        self.mu, self.theta = torch.broadcast_tensors(mu, theta)
        self.scale = scale

    @property
    def mean(self):
        return self.mu

    @property
    def variance(self):
        return self.mean + (self.mean ** 2) / self.theta

    def log_prob(self, value):
        return log_nb_positive(value, mu=self.mu, theta=self.theta, eps=self._eps)


class ZeroInflatedNegativeBinomial(NegativeBinomial):
    """scVI_distribution.py:345-433."""

    def __init__(self, total_count=None, probs=None, logits=None, mu=None, theta=None, zi_logits=None, scale=None,
                 validate_args=False):
        super().__init__(total_count=total_count, probs=probs, logits=logits, mu=mu, theta=theta, scale=scale,
                         validate_args=validate_args)
        self.zi_logits, self.mu, self.theta = torch.broadcast_tensors(zi_logits, self.mu, self.theta)

    @property
    def zi_probs(self):
        return torch.sigmoid(self.zi_logits)

    @property
    def mean(self):
        return (1 - self.zi_probs) * self.mu

    @property
    def variance(self):
        raise NotImplementedError

    def log_prob(self, value):
        return log_zinb_positive(value, self.mu, self.theta, self.zi_logits, eps=1e-08)
