#!/usr/bin/env python3
"""Diagnostic (needs a library built with -DDEC_PROF): per-phase cycle counts of one wave of the Bernoulli decoder's
training kernel at C2 shapes.  Phases are the intervals between the PROF_MARK points of decoder_kernel.h."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scpair_amd as S
from scpair_amd import ops, synth, _lib

dev = torch.device("cuda:0")
B, P, Lat = 1024, 100000, 40
S.set_precision("bf16")
atac = synth.make_atac_device(8192, P, dev, seed=2, p=0.03)
g = torch.Generator(device="cpu").manual_seed(0)
rows = torch.arange(B, device=dev)
h = (torch.randn(B, Lat, generator=g) * 0.5).to(dev).requires_grad_()
Wp = (torch.randn(P, Lat, generator=g) * 0.05).to(dev).requires_grad_()
bp = torch.zeros(P, device=dev).requires_grad_()
lat = torch.randn(B, generator=g).to(dev).requires_grad_()
ta = S.BatchTiles(atac, rows, want_lib=False)
for _ in range(3):
    loss = ops.ber_decoder_loss(h, Wp, bp, None, lat, ta)
    loss.backward()
torch.cuda.synchronize()
L = _lib.lib()
n = 12 * 64
buf = (C.c_ulonglong * n)()
rc = L.scp_dec_prof_read(buf, n)
a = np.array(buf[:], dtype=np.int64).reshape(-1, 12)
a = a[a[:, 0] > 0]
seq = [0, 1, 3, 11, 4, 5, 6, 7, 8, 9, 10]        # order of the marks inside a training iteration (mark 2: eval passes only)
names = ["B1 wait", "stage W/h", "scatter labels", "issue next loads", "B3 wait", "Z mfma", "epilogue", "dz writes", "B4 wait",
         "clear + dW/dh mfma"]
print("iterations recorded", len(a), "rc", rc)
for i, nm in enumerate(names):
    d = a[1:, seq[i + 1]] - a[1:, seq[i]]
    print("%-22s mean %7.0f   even-t %7.0f   odd-t %7.0f" % (nm, d.mean(), d[1::2].mean(), d[0::2].mean()))
tail = a[1:, 0][1:] - a[1:, 10][:-1]
print("%-22s mean %7.0f   (rotate; block end after odd tiles: row reductions, dh slab)" % ("loop end -> next top", tail.mean()))
tot = (a[1:, 0] - a[:-1, 0])
print("iteration period mean", tot.mean(), "even->odd", tot[0::2].mean(), "odd->even (block end)", tot[1::2].mean())
