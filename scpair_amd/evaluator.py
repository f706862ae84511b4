"""FOSCTTM / FOSFTTM on the GPU (evaluator.py:24-71), same names and signatures as the reference.

The reference builds the n x n x d difference tensor on the CPU; here the neighbour counts come from
`scp_fos_counts` (tiled direct-form squared distances in fp32, exact integer counts), so 100 k-cell atlases are a few
milliseconds instead of being out of memory.  Inputs: array-likes / tensors [n, d] (moved to the GPU as fp32).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


def _counts(domain1, domain2):
    dev = torch.device("cuda", torch.cuda.current_device())
    x = torch.as_tensor(np.asarray(domain1) if not isinstance(domain1, torch.Tensor) else domain1)
    y = torch.as_tensor(np.asarray(domain2) if not isinstance(domain2, torch.Tensor) else domain2)
    x = x.to(device=dev, dtype=torch.float32).contiguous()
    y = y.to(device=dev, dtype=torch.float32).contiguous()
    if x.ndim != 2 or x.shape != y.shape:
        raise ValueError("FOSCTTM/FOSFTTM need two [n, d] arrays of the same shape")
    n, d = x.shape
    diag = torch.empty(n, dtype=torch.float32, device=dev)
    lt12 = torch.empty(n, dtype=torch.int32, device=dev)
    lt21 = torch.empty(n, dtype=torch.int32, device=dev)
    gt12 = torch.empty(n, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().scp_fos_counts(x.data_ptr(), d, y.data_ptr(), d, n, d, diag.data_ptr(), lt12.data_ptr(),
                                         lt21.data_ptr(), gt12.data_ptr(), _lib.stream()), "scp_fos_counts")
    return n, lt12.cpu().numpy().astype(np.float64), lt21.cpu().numpy().astype(np.float64), \
        gt12.cpu().numpy().astype(np.float64)


def FOSCTTM(domain1, domain2, output_full=False):
    """evaluator.py:24-46: fraction of samples closer than the true match, both directions."""
    n, lt12, lt21, _ = _counts(domain1, domain2)
    fraction_1to2, fraction_2to1 = lt12 / (n - 1), lt21 / (n - 1)
    if output_full:
        return np.sort(fraction_1to2), np.sort(fraction_2to1)
    return (np.sum(fraction_2to1) / n + np.sum(fraction_1to2) / n) / 2


def FOSFTTM(query, reference, output_full=False, sort=False):
    """evaluator.py:49-71: fraction of samples farther than the true match."""
    n, _, _, gt12 = _counts(query, reference)
    fraction = gt12 / (n - 1)
    if output_full:
        return np.sort(fraction) if sort else list(fraction)
    return np.sum(fraction) / n
