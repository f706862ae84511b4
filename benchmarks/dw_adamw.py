"""Microbenchmark of the first-layer weight-gradient kernels at C2 shapes: two-kernel path (dW then AdamW) against
the kernel with the AdamW step in its epilogue."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scpair_amd as S
from scpair_amd import ops, synth

dev = torch.device("cuda:0")
S.set_precision("bf16")
B, H = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024), 900
which = sys.argv[1] if len(sys.argv) > 1 else "atac"
F = 100_000 if which == "atac" else 20_000
csr = synth.make_atac_device(8192, F, dev, seed=1, p=0.03) if which == "atac" else synth.make_rna_device(8192, F, dev, seed=1, target_density=0.10)
rows = np.random.default_rng(0).permutation(8192)[:B]
t = S.BatchTiles(csr, rows)
for mode in ("plain", "fused"):
    W = torch.nn.Parameter(torch.randn(H, F, device=dev) * 0.02)
    opt = ops.FusedAdamW([W], lr=1e-3, weight_decay=1e-9)
    if mode == "fused":
        opt.fuse_first_layer(W)
    dout = torch.randn(B, H, device=dev)
    def step():
        out = ops.fcnn_layer(t, None, W, None, None, None, 1.0)
        opt.zero_grad()
        out.backward(dout)
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tm = []
    ops.set_fused_timer(tm)
    e0.record()
    for _ in range(20):
        step()
    e1.record(); torch.cuda.synchronize()
    ops.set_fused_timer(None)
    if tm:
        print(f"   fused dW+AdamW kernel alone: {sum(a.elapsed_time(b) for a, b, *_ in tm) / len(tm) * 1e3:.1f} us (B={B})")
    print(f"{which} {mode} stagger={os.environ.get('SCP_DW_STAGGER', '0')}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per fwd+bwd+opt step")
