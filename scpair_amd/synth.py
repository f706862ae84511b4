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


# ---- device-side generators for benchmark-scale inputs (same distributions, torch RNG on the GPU) ----------
def make_atac_device(n_cells, n_peaks, device, seed=1, p=0.03, chunk=8192):
    """Bernoulli(p) per entry, generated as a Bernoulli process (geometric gaps) directly in GPU memory."""
    import torch
    from .sparse import DeviceCSR
    g = torch.Generator(device=device).manual_seed(seed)
    m = int(n_peaks * p * 1.25) + 64
    idx_chunks, counts = [], []
    log1mp = float(np.log1p(-p))
    for r0 in range(0, n_cells, chunk):
        r = min(chunk, n_cells - r0)
        u = torch.rand((r, m), generator=g, device=device).clamp_min_(1e-12)
        gaps = torch.floor(torch.log(u) / log1mp).to(torch.int64) + 1          # Geometric(p) on {1,2,...}
        cols = torch.cumsum(gaps, 1) - 1
        keep = cols < n_peaks
        counts.append(keep.sum(1))
        idx_chunks.append(cols[keep].to(torch.int32))
    counts = torch.cat(counts)
    indptr = torch.zeros(n_cells + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(counts, 0)
    return DeviceCSR.from_device(indptr, torch.cat(idx_chunks), None, (n_cells, n_peaks))


def make_rna_device(n_cells, n_genes, device, seed=0, shape_g=0.3, target_density=0.1, chunk=4096):
    """Poisson(g_j * s_i) counts generated chunk-wise in GPU memory and compacted to CSR."""
    import torch
    from .sparse import DeviceCSR
    rng = np.random.default_rng(seed)
    gj = rng.gamma(shape_g, 1.0, size=n_genes)
    si = rng.gamma(5.0, 0.2, size=n_cells)
    if target_density is not None:
        lo, hi = 1e-4, 1e3
        for _ in range(40):
            mid = np.sqrt(lo * hi)
            dens = 1.0 - np.mean(np.exp(-np.outer(si[:256], gj * mid)))
            lo, hi = (lo, mid) if dens > target_density else (mid, hi)
        gj = gj * np.sqrt(lo * hi)
    g = torch.Generator(device=device).manual_seed(seed)
    gj_d = torch.tensor(gj, dtype=torch.float32, device=device)
    si_d = torch.tensor(si, dtype=torch.float32, device=device)
    idx_chunks, val_chunks, counts = [], [], []
    for r0 in range(0, n_cells, chunk):
        lam = si_d[r0:r0 + chunk, None] * gj_d[None, :]
        x = torch.poisson(lam, generator=g)
        nz = x > 0
        counts.append(nz.sum(1))
        rc = nz.nonzero()
        idx_chunks.append(rc[:, 1].to(torch.int32))
        val_chunks.append(x[nz])
    counts = torch.cat(counts)
    indptr = torch.zeros(n_cells + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(counts, 0)
    return DeviceCSR.from_device(indptr, torch.cat(idx_chunks), torch.cat(val_chunks), (n_cells, n_genes))
