import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import scpair_amd as S
from scpair_amd import ops, synth
dev = torch.device("cuda:0")
S.set_precision("bf16")
B, F, H1, H2 = 1024, 100000, 900, 40
atac = synth.make_atac_device(4096, F, dev, seed=1, p=0.03)
t = S.BatchTiles(atac, np.arange(B), want_lib=False)
g = torch.Generator(device=dev).manual_seed(0)
def P(n3):
    return dict(W1=torch.randn(H1, F, device=dev, generator=g) * 0.02, b1=torch.zeros(H1, device=dev), g1=torch.ones(H1, device=dev), be1=torch.zeros(H1, device=dev),
                W2=torch.randn(H2, H1, device=dev, generator=g) * 0.03, b2=torch.zeros(H2, device=dev), g2=torch.ones(H2, device=dev), be2=torch.zeros(H2, device=dev),
                W3=torch.randn(n3, H2, device=dev, generator=g) * 0.1, b3=torch.zeros(n3, device=dev), p1=0.1, p2=0.1)
br = [P(40)]
with torch.no_grad():
    for _ in range(3):
        ops.enc_branches(t, br, 0.01)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.enc_branches(t, br, 0.01)
    e1.record(); torch.cuda.synchronize()
print("dbg", os.environ.get("SCP_TAIL_DBG", "0"), "fwd product + tail:", e0.elapsed_time(e1) / 20 * 1e3, "us")
