#!/usr/bin/env python3
"""Entries per (64-cell block, feature tile) of the bench's synthetic minibatches: how often do the per-thread entry
slots of the first-layer kernels overflow (forward: 64 features, 128 threads x EPT; wgrad: 256 or 128 features)?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scpair_amd as S
from scpair_amd import synth
dev = torch.device("cuda:0")
B = 1024
for name, csr, ept in (("rna", synth.make_rna_device(8192, 20000, dev, seed=1, target_density=0.10), 4),
                       ("atac", synth.make_atac_device(8192, 100000, dev, seed=2, p=0.03), 2)):
    rows = np.random.default_rng(0).permutation(8192)[:B]
    t = S.BatchTiles(csr, rows)
    F = csr.shape[1]
    ptr = t.ptr.cpu().numpy().astype(np.int64)
    ncb = B // 64
    ptr = ptr[:ncb * F + 1]
    P = np.concatenate([ptr[:ncb * F].reshape(ncb, F), np.full((ncb, 1), 0)], axis=1)
    for cb in range(ncb):
        P[cb, F] = ptr[cb * F + F] if cb * F + F < len(ptr) else ptr[-1]
    for width, cap, what in ((64, 128 * ept, "fwd K-step"), (256, 512 * ept, "wgrad 256"), (128, 256 * ept, "wgrad 128")):
        edges = np.minimum(np.arange(0, F + width, width), F)
        cnt = P[:, edges[1:]] - P[:, edges[:-1]]
        print(f"{name} {what}: tile {width} features, capacity {cap}: mean {cnt.mean():.0f} max {cnt.max()} "
              f"overflow share {np.mean(cnt > cap):.4f}")
