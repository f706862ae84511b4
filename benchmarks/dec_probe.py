#!/usr/bin/env python3
"""Where does the decoder kernels' time go?  Times the fused NB / Bernoulli decoders at C2 shapes with the real
labels, with labels capped at 1 (no lgamma / digamma work), and with no non-zero labels at all."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scpair_amd as S
from scpair_amd import ops, synth

dev = torch.device("cuda:0")
B, G, P, Lat = 1024, 20000, 100000, 40
S.set_precision(os.environ.get("PREC", "bf16"))


def timeit(fn, rep=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep * 1e3


rna = synth.make_rna_device(8192, G, dev, seed=1, target_density=0.10)
atac = synth.make_atac_device(8192, P, dev, seed=2, p=0.03)
rows = np.random.default_rng(0).permutation(8192)[:B]
g = torch.Generator(device=dev).manual_seed(0)
h = torch.randn(B, Lat, device=dev, generator=g).requires_grad_(True)
Wg = (torch.randn(G, Lat, device=dev, generator=g) * 0.1).requires_grad_(True)
bg = torch.zeros(G, device=dev, requires_grad=True)
t = S.BatchTiles(rna, rows)
print("rna batch: nnz", t.nnz, "max count", float(t.val[:t.nnz].max()), "share > 12:", float((t.val[:t.nnz] > 12).float().mean()))
for name, mod in (("real", None), ("capped at 1", "cap"), ("capped at 12", "cap12")):
    tt = S.BatchTiles(rna, rows)
    if mod == "cap":
        tt.val.clamp_(max=1.0)
    if mod == "cap12":
        tt.val.clamp_(max=12.0)
    us = timeit(lambda: ops.nb_decoder_loss(h, Wg, bg, None, None, tt.lib, None, tt))
    with torch.no_grad():
        us_e = timeit(lambda: ops.nb_decoder_loss(h, Wg, bg, None, None, tt.lib, None, tt))
    print(f"nb decoder, labels {name:14s}: train {us:7.1f} us, eval (stats + loss) {us_e:7.1f} us")
empty = S.DeviceCSR.from_device(torch.zeros(8193, dtype=torch.int64, device=dev), torch.zeros(0, dtype=torch.int32, device=dev),
                                torch.zeros(0, device=dev), (8192, G))
te = S.BatchTiles(empty, rows)
us = timeit(lambda: ops.nb_decoder_loss(h, Wg, bg, None, None, t.lib, None, te))
print(f"nb decoder, no non-zero labels     : train {us:7.1f} us")
Wp = (torch.randn(P, Lat, device=dev, generator=g) * 0.1).requires_grad_(True)
bp = torch.full((P,), -3.0, device=dev).requires_grad_(True)
lat = torch.randn(B, 1, device=dev, generator=g).requires_grad_(True)
ta = S.BatchTiles(atac, rows, want_lib=False)
us = timeit(lambda: ops.ber_decoder_loss(h, Wp, bp, None, lat, ta))
print(f"ber decoder train                  : {us:7.1f} us")
emptya = S.DeviceCSR.from_device(torch.zeros(8193, dtype=torch.int64, device=dev), torch.zeros(0, dtype=torch.int32, device=dev),
                                 None, (8192, P))
tea = S.BatchTiles(emptya, rows, want_lib=False)
us = timeit(lambda: ops.ber_decoder_loss(h, Wp, bp, None, lat, tea))
print(f"ber decoder train, no labels       : {us:7.1f} us")
