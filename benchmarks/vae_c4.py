#!/usr/bin/env python3
"""C4-shaped check of the unimodal VAE step (pairedVAE.py, utils.py:179-202): 20 k genes, hidden [900], latent 40,
NB likelihood, batch 1024, bf16 mode; cells resident as CSR.  Prints ms per step (one cell through one VAE step)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import scpair_amd as S
from scpair_amd import synth
dev = torch.device("cuda:0")
S.set_precision(os.environ.get("PREC", "bf16"))
G, B, N = 20000, 1024, 50000
lik = os.environ.get("LIK", "nb")
rna = synth.make_rna_device(N, G, dev, seed=1, target_density=0.10)
S.set_seed(0)
enc = S.VAE_ENCODER(input_dim=G, input_batch_num=0, hidden_dim=[900], latent_dim=40, layernorm=True,
                    activation=torch.nn.LeakyReLU(), batchnorm=False, dropout_rate=0.1, infer_library_size=False).to(dev)
dec = S.VAE_DECODER(input_dim=G, input_batch_num=0, hidden_dim=[900], latent_dim=40, layernorm=True,
                    activation=torch.nn.LeakyReLU(), batchnorm=False, dropout_rate=0.1, infer_library_size=False,
                    sample_factor=True, feature_factor=False, distribution=lik,
                    dispersion="feature-cell").to(dev)
opt = S.FusedAdamW(list(enc.parameters()) + list(dec.parameters()), lr=1e-3, weight_decay=0.0)
if os.environ.get("FUSE", "1") == "1":
    opt.fuse_first_layer(enc.enc_hidden[0].weight)      # AdamW step of the 900 x 20 k encoder weights in their dW kernel
    if os.environ.get("FUSE_WIDE", "0") == "1":         # experiment: also the decoder's 900 -> 20 k heads (measured slower)
        opt.fuse_weight_update(*[m.weight for m in (getattr(dec, n_, None) for n_ in ("output", "output_inverse_dispersion", "output_dropout"))
                                 if isinstance(m, torch.nn.Linear)])
enc.train(); dec.train()
rng = np.random.default_rng(0)
def step():
    rows = rng.permutation(N)[:B]
    t = S.BatchTiles(rna, rows, want_lib=True)
    return S.train_VAE(t, t.lib.reshape(-1, 1), enc, dec, opt, kl_weight=1.0, likelihood_type=lik)
for _ in range(3):
    l = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = int(os.environ.get("STEPS", 10))
for _ in range(K):
    l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print(f"VAE {lik} step: {dt*1e3:.3f} ms  ({B/dt:,.0f} cells/s)  loss {float(l):.4f}")
