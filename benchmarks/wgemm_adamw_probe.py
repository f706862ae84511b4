"""Weight gradient of the 20 k x 20 k layer + AdamW: fused (scp_gemm_bf16_adamw) against scp_gemm_bf16 + scp_adamw_multi."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scpair_amd import ops
G = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
M = 1024
dev = torch.device("cuda:0")
ops.set_precision("bf16")
x = torch.randn(M, G, device=dev)
dy = torch.randn(M, G, device=dev) * 0.01
for mode in ("plain", "fused"):
    W = torch.nn.Parameter(torch.randn(G, G, device=dev) * 0.02)
    opt = ops.FusedAdamW([W], lr=1e-3, weight_decay=1e-9)
    if mode == "fused":
        opt.fuse_weight_update(W)
    def step():
        opt.zero_grad()
        ops.linear(x, W).backward(dy)
        opt.step(release_grads=True)
    for _ in range(3): step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): step()
    b.record(); torch.cuda.synchronize()
    print(f"{mode}: {a.elapsed_time(b) / 10:.3f} ms per fwd + bwd + AdamW of a {G} x {G} layer, batch {M}", flush=True)
    del W, opt
    torch.cuda.empty_cache()
