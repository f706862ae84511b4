#!/usr/bin/env python3
"""Per-kernel micro-benchmark at the headline shapes (C2: B=1024, G=20k, P=100k, hidden [900,40]).
Prints achieved time / TFLOP/s / GB/s per kernel family.  Usage: python benchmarks/kernels.py [which ...]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scpair_amd as S
from scpair_amd import ops, synth, _lib

dev = torch.device("cuda:0")
which = set(sys.argv[1:]) or {"tiles", "spmm", "ber", "nb", "adamw", "small"}
B, G, P, H, Lat = 1024, 20000, 100000, 900, 40
prec = os.environ.get("PREC", "bf16")
S.set_precision(prec)
REP = int(os.environ.get("REP", "10"))


def timeit(fn, rep=REP, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep * 1e3   # us


ncell = 8192
rna = synth.make_rna_device(ncell, G, dev, seed=1, target_density=0.10)
atac = synth.make_atac_device(ncell, P, dev, seed=2, p=0.03)
rows = np.random.default_rng(0).permutation(ncell)[:B]
t_rna = S.BatchTiles(rna, rows)
t_atac = S.BatchTiles(atac, rows, want_lib=False)
g = torch.Generator(device=dev).manual_seed(0)
L = _lib.lib()

if "tiles" in which:
    print(f"tiles rna  : {timeit(lambda: S.BatchTiles(rna, rows)):8.1f} us  (nnz {t_rna.nnz})")
    print(f"tiles atac : {timeit(lambda: S.BatchTiles(atac, rows, want_lib=False)):8.1f} us  (nnz {t_atac.nnz})")

if "spmm" in which:
    for name, t, F in (("rna", t_rna, G), ("atac", t_atac, P)):
        W = (torch.randn(H, F, device=dev, generator=g) * 0.01).requires_grad_(True)
        b = torch.zeros(H, device=dev, requires_grad=True)
        gam = torch.ones(H, device=dev, requires_grad=True); bet = torch.zeros(H, device=dev, requires_grad=True)
        out = ops.fcnn_layer(t, None, W, b, gam, bet, 0.01)
        dout = torch.randn_like(out)
        us_f = timeit(lambda: ops.fcnn_layer(t, None, W, b, gam, bet, 0.01))
        def fb():
            o = ops.fcnn_layer(t, None, W, b, gam, bet, 0.01)
            o.backward(dout)
        us_fb = timeit(fb)
        fl = 2.0 * B * F * H
        print(f"fcnn layer-1 {name:4s}: fwd {us_f:8.1f} us ({fl/us_f/1e6:6.1f} TF/s dense-equiv) ; fwd+bwd {us_fb:8.1f} us")

if "ber" in which:
    h = torch.randn(B, Lat, device=dev, generator=g).requires_grad_(True)
    W = (torch.randn(P, Lat, device=dev, generator=g) * 0.1).requires_grad_(True)
    b = torch.full((P,), -3.0, device=dev).requires_grad_(True)
    lat = torch.randn(B, 1, device=dev, generator=g).requires_grad_(True)
    us = timeit(lambda: ops.ber_decoder_loss(h, W, b, None, lat, t_atac))
    print(f"ber decoder fused fwd+bwd: {us:8.1f} us ({B*P/us/1e3:6.1f} Gelem/s)")
    with torch.no_grad():
        us = timeit(lambda: ops.ber_decoder_loss(h, W, b, None, lat, t_atac))
    print(f"ber decoder eval         : {us:8.1f} us")

if "nb" in which:
    h = torch.randn(B, Lat, device=dev, generator=g).requires_grad_(True)
    W = (torch.randn(G, Lat, device=dev, generator=g) * 0.1).requires_grad_(True)
    b = torch.zeros(G, device=dev, requires_grad=True)
    us = timeit(lambda: ops.nb_decoder_loss(h, W, b, None, None, t_rna.lib, None, t_rna))
    print(f"nb decoder fused 3 passes: {us:8.1f} us ({B*G/us/1e3:6.1f} Gelem/s)")

if "adamw" in which:
    n = 90_000_000
    p = torch.nn.Parameter(torch.randn(n, device=dev)); p.grad = torch.randn(n, device=dev)
    opt = S.FusedAdamW([p], lr=1e-3, weight_decay=1e-9)
    us = timeit(lambda: opt.step())
    print(f"adamw 90M params         : {us:8.1f} us ({28.0*n/us/1e3:7.1f} GB/s algorithmic)")

if "small" in which:
    x = torch.randn(B, H, device=dev, generator=g).requires_grad_(True)
    W = (torch.randn(Lat, H, device=dev, generator=g) * 0.03).requires_grad_(True)
    bb = torch.zeros(Lat, device=dev, requires_grad=True)
    dy = torch.randn(B, Lat, device=dev, generator=g)
    def fb():
        y = ops.linear(x, W, bb); y.backward(dy)
    print(f"linear 900->40 fwd+bwd   : {timeit(fb):8.1f} us")

if "decdbg" in which:
    import ctypes
    out = (ctypes.c_longlong * 96)()
    L.scp_dec_dbg_read.restype = ctypes.c_int
    L.scp_dec_dbg_read(out)
    v = list(out)
    names = ["top_sync", "stage+zero+sync", "scatter", "issue+sync", "Zmfma", "epilogue+dzstore", "sync", "gradmfma"]
    for mode, nm in enumerate(["BER_TRAIN", "BER_EVAL", "NB_STATS", "NB_LOSS", "NB_GRAD"]):
        n = v[mode * 16 + 8]
        if n:
            print(nm, "iters", n, "total", v[mode * 16 + 9], {names[k]: round(v[mode * 16 + k] / n) for k in range(8)},
                  "sum/iter", round(sum(v[mode * 16:mode * 16 + 8]) / n), "total/iter", round(v[mode * 16 + 9] / n))
