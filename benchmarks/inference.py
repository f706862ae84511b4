#!/usr/bin/env python3
"""Inference paths at C2 shapes: embeddings of 100 k cells through the trained-shape encoders
(reference_embeddings, model.py:527-542) and decoder predictions (predict, model.py:558-591) in chunks, all on the GPU
from resident CSR.  Random weights; prints cells/s."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import scpair_amd as S
from scpair_amd import synth
from torch import nn

dev = torch.device("cuda:0")
S.set_precision(os.environ.get("PREC", "bf16"))
N, G, P, CH = 100_000, 20_000, 100_000, int(os.environ.get("CHUNK", 4096))
rna = synth.make_rna_device(N, G, dev, seed=1, target_density=0.10)
atac = synth.make_atac_device(N, P, dev, seed=2, p=0.03)
S.set_seed(0)
enc_r = S.Input_Module(input_dim=G, input_batch_num=0, hidden_layer=[900, 40], activation=nn.LeakyReLU(), layernorm=True,
                       batchnorm=False, dropout_rate=0.1, add_linear_layer=True, infer_library_size=True).to(dev).eval()
enc_a = S.Input_Module(input_dim=P, input_batch_num=0, hidden_layer=[900, 40], activation=nn.LeakyReLU(), layernorm=True,
                       batchnorm=False, dropout_rate=0.1, add_linear_layer=True, infer_library_size=False).to(dev).eval()
dec_a = S.Output_Module_Ber(output_dim=P, output_batch_num=0, hidden_layer=[900, 40], infer_library_size=True,
                            sample_factor=False, feature_factor=False).to(dev).eval()


@torch.no_grad()
def embed(enc, csr):
    out = []
    for r0 in range(0, N, CH):
        rows = np.arange(r0, min(N, r0 + CH))
        out.append(enc(S.BatchTiles(csr, rows, want_lib=False), None)[0])
    return torch.cat(out, 0)


for name, enc, csr in (("RNA", enc_r, rna), ("ATAC", enc_a, atac)):
    embed(enc, csr); torch.cuda.synchronize()
    t0 = time.perf_counter(); e = embed(enc, csr); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"embeddings {name}: {N} cells in {dt*1e3:.1f} ms ({N/dt:,.0f} cells/s), chunk {CH}")

with torch.no_grad():
    emb = embed(enc_r, rna)
    lat = torch.zeros(N, 1, device=dev)
    one = torch.ones(1, device=dev)
    n_pred = 8 * CH
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for r0 in range(0, n_pred, CH):
            p = dec_a(emb[r0:r0 + CH], None, one, lat[r0:r0 + CH])[0]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"predict RNA->ATAC probabilities: {n_pred} cells x {P} peaks in {dt*1e3:.1f} ms ({n_pred/dt:,.0f} cells/s, "
          f"{n_pred*P*4/dt/1e9:.0f} GB/s of fp32 output written to HBM; the copy to the host is not included)")
