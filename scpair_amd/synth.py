"""Synthetic paired RNA+ATAC count matrices (SURVEY.md section 8d generator).

RNA counts  ~ Poisson(g_j * s_i),  g_j ~ Gamma(shape_g, 1),  s_i ~ Gamma(5, 0.2)
ATAC        ~ Bernoulli(p_atac)    stored as CSR with explicit 1.0 values
split       80/10/10 by a seeded permutation

The same generator (same seeds) feeds the golden-fixture script, the CPU oracle leg of
bench.py and the GPU path, so every leg sees identical inputs.  Nothing here touches the GPU.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def make_rna(n_cells: int, n_genes: int, seed: int = 0, shape_g: float = 0.3,
             target_density: float | None = None) -> sp.csr_matrix:
    """Poisson-Gamma counts, canonical CSR (sorted indices, no duplicates), float32 values."""
    rng = np.random.default_rng(seed)
    g = rng.gamma(shape_g, 1.0, size=n_genes)
    s = rng.gamma(5.0, 0.2, size=n_cells)
    if target_density is not None:
        # scale gene rates so that the expected non-zero fraction ~= target_density
        lo, hi = 1e-4, 1e3
        for _ in range(40):
            mid = np.sqrt(lo * hi)
            dens = 1.0 - np.mean(np.exp(-np.outer(s[:256], g * mid)))
            if dens > target_density:
                hi = mid
            else:
                lo = mid
        g = g * np.sqrt(lo * hi)
    rows = []
    chunk = max(1, min(n_cells, int(2e7 // max(n_genes, 1))))
    for c0 in range(0, n_cells, chunk):
        lam = np.outer(s[c0:c0 + chunk], g)
        rows.append(sp.csr_matrix(rng.poisson(lam).astype(np.float32)))
    m = sp.vstack(rows, format="csr")
    m.sort_indices()
    return m


def make_atac(n_cells: int, n_peaks: int, seed: int = 1, p: float = 0.03) -> sp.csr_matrix:
    """Bernoulli(p) accessibility, canonical CSR with explicit 1.0 float32 values."""
    rng = np.random.default_rng(seed)
    indptr = np.zeros(n_cells + 1, dtype=np.int64)
    idx_chunks = []
    nnz_per_row = rng.binomial(n_peaks, p, size=n_cells)
    for i in range(n_cells):
        k = int(nnz_per_row[i])
        cols = np.sort(rng.choice(n_peaks, size=k, replace=False)) if k else np.empty(0, np.int64)
        idx_chunks.append(cols.astype(np.int32))
        indptr[i + 1] = indptr[i] + k
    indices = np.concatenate(idx_chunks) if idx_chunks else np.empty(0, np.int32)
    data = np.ones(indices.shape[0], dtype=np.float32)
    return sp.csr_matrix((data, indices, indptr), shape=(n_cells, n_peaks))


def split_indices(n_cells: int, seed: int = 0, fracs=(0.8, 0.1, 0.1)):
    rng = np.random.default_rng(seed + 12345)
    perm = rng.permutation(n_cells)
    n_tr = int(round(fracs[0] * n_cells))
    n_va = int(round(fracs[1] * n_cells))
    return perm[:n_tr], perm[n_tr:n_tr + n_va], perm[n_tr + n_va:]


def make_paired(n_cells: int, n_genes: int, n_peaks: int, seed: int = 0,
                p_atac: float = 0.03, rna_density: float | None = None):
    """Returns dict: rna/atac CSR (all cells) + train/val/test index arrays."""
    rna = make_rna(n_cells, n_genes, seed=seed, target_density=rna_density)
    atac = make_atac(n_cells, n_peaks, seed=seed + 1, p=p_atac)
    tr, va, te = split_indices(n_cells, seed)
    return {"rna": rna, "atac": atac, "train": tr, "val": va, "test": te}
