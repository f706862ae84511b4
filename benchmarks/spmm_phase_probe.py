#!/usr/bin/env python3
"""Diagnostic (needs a library built with -DSPMM_PROF): per-phase cycle counts of one wave of the forward first-layer
kernel at C2 ATAC shapes (900 x 100 k weights, 1024 cells)."""
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
for _ in range(3):
    out = ops.fcnn_layer(t, None, W, None, None, None, 1.0)
torch.cuda.synchronize()
L = _lib.lib()
n = 8 * 300
buf = (C.c_ulonglong * n)()
rc = L.scp_spmm_prof_read(buf, n)
a = np.array(buf[:], dtype=np.int64).reshape(-1, 8)
a = a[(a[:, 0] > 0) & (a[:, 6] > 0)]
print("steps recorded", len(a), "rc", rc)
names = ["fragment reads issued", "stage_out (wait oldest loads, LDS writes)", "mfma 0-1 issued", "prefetch issued", "mfma 2-3 + unscatter issued", "barrier"]
d = np.diff(a[:, :7], axis=1)
sel = slice(6, len(a) - 6)
for i, nm in enumerate(names):
    print("%-44s mean %7.0f  median %7.0f" % (nm, d[sel, i].mean(), np.median(d[sel, i])))
per = np.diff(a[:, 0])
print("step period mean %.0f median %.0f" % (per[sel].mean(), np.median(per[sel])))
