#!/usr/bin/env python3
"""Diagnostic (needs a library built with -DSPMM_PROF): where a workgroup of the fused weight-gradient + AdamW kernel
(narrow variant, C2 ATAC shapes) spends its lifetime: prologue, K loop, gradient tile to LDS, AdamW epilogue."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scpair_amd as S
from scpair_amd import ops, synth, _lib
dev = torch.device("cuda:0")
S.set_precision("bf16")
B, H, F = 1024, 900, 100_000
csr = synth.make_atac_device(8192, F, dev, seed=1, p=0.03)
rows = np.random.default_rng(0).permutation(8192)[:B]
t = S.BatchTiles(csr, rows)
W = torch.nn.Parameter(torch.randn(H, F, device=dev) * 0.02)
opt = ops.FusedAdamW([W], lr=1e-3, weight_decay=1e-9)
opt.fuse_first_layer(W)
dout = torch.randn(B, H, device=dev)
for _ in range(3):
    out = ops.fcnn_layer(t, None, W, None, None, None, 1.0)
    opt.zero_grad(); out.backward(dout); opt.step()
torch.cuda.synchronize()
L = _lib.lib()
n = 8 * 500
buf = (C.c_ulonglong * n)()
rc = L.scp_spmm_prof_read(buf, n)
a = np.array(buf[:], dtype=np.int64).reshape(-1, 8)
a = a[(a[:, 0] > 0) & (a[:, 5] > 0)]
print("workgroups recorded", len(a), "rc", rc)
for nm, i, j in (("prologue (ranges, zeroing, first loads)", 0, 1), ("K loop (16 steps)", 1, 2), ("gradient tile -> LDS", 2, 3),
                 ("AdamW epilogue (loads, update, stores)", 3, 5), ("whole workgroup", 0, 5)):
    d = a[:, j] - a[:, i]
    print("%-42s mean %8.0f  median %8.0f cycles" % (nm, d.mean(), np.median(d)))
