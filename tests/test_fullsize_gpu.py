"""Parity at BASELINE.json's C2 sizes (batch 1024, 20 k genes, 100 k peaks, hidden 900, latent 40).

The CPU oracle needs minutes per step at these sizes, so the checks here are size-independent properties and
device-side references built from plain torch fp32/fp64 tensor ops on the same inputs (a floating-point reference
for floating-point kernels; the oracle itself is pinned at small sizes in test_kernels_gpu.py):

  * tile list: exact round trip CSR rows -> tile list -> dense, conservation of entries, library sizes
  * first-layer GEMMs (bf16 mode): against matmuls of the bf16-rounded operands; linear in the sparse operand
  * fused decoders (fp32-equivalent mode): loss and every gradient against torch autograd of the reference formula
    (model.py:164-190 / 229-259, loss.py:36-48, scVI_distribution.py:74-105); bitwise reproducible run to run
  * fused AdamW on 90 M parameters against the update formula (torch.optim.AdamW semantics, model.py:698-701)

Tolerances: bf16 mode 2e-3 of the output scale against references with identically rounded operands (fp32
accumulation order); fp32-equivalent mode 1e-4 (loss) / 1e-3 of scale (gradients, fp32 torch reference noise).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B, G, P, H, LAT = 1024, 20_000, 100_000, 900, 40
NCELL = 4096


@pytest.fixture(scope="module")
def D():
    import scpair_amd as S
    from scpair_amd import ops, synth
    dev = torch.device("cuda:0")
    rna = synth.make_rna_device(NCELL, G, dev, seed=11, target_density=0.10)
    atac = synth.make_atac_device(NCELL, P, dev, seed=12, p=0.03)
    rows = np.random.default_rng(3).permutation(NCELL)[:B]
    return type("D", (), dict(S=S, ops=ops, dev=dev, rna=rna, atac=atac, rows=rows,
                              t_rna=S.BatchTiles(rna, rows), t_atac=S.BatchTiles(atac, rows, want_lib=True)))


def dense_rows(csr, rows, dev):
    """Reference densification with torch indexing only."""
    indptr = csr.indptr.cpu().numpy(); rows = np.asarray(rows)
    out = torch.zeros((len(rows), csr.n_cols), dtype=torch.float32, device=dev)
    lens = indptr[rows + 1] - indptr[rows]
    src = np.concatenate([np.arange(indptr[r], indptr[r + 1]) for r in rows]) if lens.sum() else np.zeros(0, np.int64)
    src = torch.from_numpy(src).to(dev)
    r_of = torch.repeat_interleave(torch.arange(len(rows), device=dev), torch.from_numpy(lens).to(dev))
    cols = csr.indices[src].long()
    vals = torch.ones(len(src), device=dev) if csr.values is None else csr.values[src]
    out[r_of, cols] = vals
    return out


def relmax(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def test_tile_list_round_trip(D):
    for csr, t in ((D.rna, D.t_rna), (D.atac, D.t_atac)):
        ref = dense_rows(csr, D.rows, D.dev)
        got = t.to_dense()
        assert torch.equal(got, ref)                                   # exact: integer-valued fp32 / byte work
        ptr = t.ptr[: ((B + 63) // 64) * t.F + 1].long()
        assert bool((ptr[1:] >= ptr[:-1]).all()) and int(ptr[0]) == 0 and int(ptr[-1]) == t.nnz
        assert int((ref != 0).sum()) == t.nnz
        assert torch.equal(t.lib, ref.sum(1))                          # counts: exact in fp32
        # every entry sits in the bucket its key names
        key = t.key[: t.nnz].long()
        e = torch.arange(t.nnz, device=D.dev)
        bucket = torch.searchsorted(ptr, e, right=True) - 1
        assert torch.equal(bucket % t.F, key >> 6)


@pytest.mark.parametrize("which", ["rna", "atac"])
def test_first_layer_bf16_full_size(D, which):
    ops = D.ops
    ops.set_precision("bf16")
    t, F = (D.t_rna, G) if which == "rna" else (D.t_atac, P)
    csr = D.rna if which == "rna" else D.atac
    g = torch.Generator(device=D.dev).manual_seed(1)
    W = (torch.randn(H, F, device=D.dev, generator=g) * 0.02).requires_grad_(True)
    b = (torch.randn(H, device=D.dev, generator=g) * 0.1).requires_grad_(True)
    dout = torch.randn(B, H, device=D.dev, generator=g)
    out = ops.fcnn_layer(t, None, W, b, None, None, 1.0)              # slope 1: the bare linear layer
    out.backward(dout)
    X = dense_rows(csr, D.rows, D.dev)
    Xb, Wb = X.bfloat16().float(), W.detach().bfloat16().float()
    ref = Xb @ Wb.t() + b.detach()
    assert relmax(out, ref) < 2e-3
    ref_dW = dout.bfloat16().float().t() @ Xb
    assert relmax(W.grad, ref_dW) < 2e-3
    assert relmax(b.grad, dout.sum(0)) < 1e-5
    # linearity in the sparse operand: rows of (t) processed alone give the same rows (cell blocks are independent)
    sub = D.S.BatchTiles(csr, D.rows[:200])
    out2 = ops.fcnn_layer(sub, None, W.detach(), b.detach(), None, None, 1.0)
    assert relmax(out2, out[:200].detach()) < 1e-5
    ops.set_precision("fp32")


def test_ber_decoder_full_size(D):
    ops = D.ops
    ops.set_precision("fp32")
    g = torch.Generator(device=D.dev).manual_seed(2)
    h = (torch.randn(B, LAT, device=D.dev, generator=g) * 0.5).requires_grad_(True)
    W = (torch.randn(P, LAT, device=D.dev, generator=g) * 0.2).requires_grad_(True)
    b = (torch.randn(P, device=D.dev, generator=g) * 0.5 - 3.0).requires_grad_(True)
    lat = torch.randn(B, 1, device=D.dev, generator=g).requires_grad_(True)
    l = ops.ber_decoder_loss(h, W, b, None, lat, D.t_atac)
    l.backward()
    got = [l.detach().clone(), h.grad.clone(), W.grad.clone(), b.grad.clone(), lat.grad.clone()]
    # reference formula, torch autograd in fp64 over feature chunks
    Y = dense_rows(D.atac, D.rows, D.dev)
    hr, Wr, br, lr = [x.detach().double().requires_grad_(True) for x in (h, W, b, lat)]
    tot = torch.zeros((), dtype=torch.float64, device=D.dev)
    for f0 in range(0, P, 25_000):
        sl = slice(f0, f0 + 25_000)
        q = torch.sigmoid(hr @ Wr[sl].t() + br[sl]) * torch.sigmoid(lr)
        y = Y[:, sl].double()
        tot = tot - (y * torch.log(q).clamp_min(-100) + (1 - y) * torch.log(1 - q).clamp_min(-100)).sum()
    ref = tot / (B * P)
    ref.backward()
    assert abs(float(got[0]) - float(ref)) <= 1e-4 * abs(float(ref))
    for nm, a, r_ in (("dh", got[1], hr.grad), ("dW", got[2], Wr.grad), ("db", got[3], br.grad), ("dl", got[4], lr.grad)):
        assert relmax(a, r_) < 1e-3, nm
    # bitwise reproducible (fixed-order reductions, no float atomics)
    for x in (h, W, b, lat):
        x.grad = None
    l2 = ops.ber_decoder_loss(h, W, b, None, lat, D.t_atac)
    l2.backward()
    assert torch.equal(l2.detach(), got[0]) and torch.equal(h.grad, got[1]) and torch.equal(W.grad, got[2])
    # bf16 mode stays within its tolerance of the same reference
    ops.set_precision("bf16")
    with torch.no_grad():
        lb = ops.ber_decoder_loss(h, W, b, None, lat, D.t_atac)
    assert abs(float(lb) - float(ref)) <= 1e-2 * abs(float(ref))
    ops.set_precision("fp32")


def test_nb_decoder_full_size(D):
    ops = D.ops
    ops.set_precision("fp32")
    g = torch.Generator(device=D.dev).manual_seed(3)
    h = (torch.randn(B, LAT, device=D.dev, generator=g) * 0.5).requires_grad_(True)
    W = (torch.randn(G, LAT, device=D.dev, generator=g) * 0.2).requires_grad_(True)
    b = (torch.randn(G, device=D.dev, generator=g) * 0.3 + 0.5).requires_grad_(True)
    lib = D.t_rna.lib
    l = ops.nb_decoder_loss(h, W, b, None, None, lib, None, D.t_rna)
    l.backward()
    Y = dense_rows(D.rna, D.rows, D.dev).double()
    hr, Wr, br = [x.detach().double().requires_grad_(True) for x in (h, W, b)]
    z = hr @ Wr.t() + br
    a = torch.relu(z)
    mu = lib.double().reshape(-1, 1) * torch.softmax(a, dim=1)
    th = torch.exp(a)                                                  # dispersion feature-cell: exp(px_mu), px_mu = relu(z)
    eps = 1e-8
    ll = th * (torch.log(th + eps) - torch.log(th + mu + eps)) + Y * (torch.log(mu + eps) - torch.log(th + mu + eps)) \
        + torch.lgamma(Y + th) - torch.lgamma(th) - torch.lgamma(Y + 1)
    ref = -ll.mean()
    ref.backward()
    assert abs(float(l) - float(ref)) <= 1e-4 * abs(float(ref))
    for nm, a_, r_ in (("dh", h.grad, hr.grad), ("dW", W.grad, Wr.grad), ("db", b.grad, br.grad)):
        assert relmax(a_, r_) < 2e-3, nm


def test_adamw_full_size(D):
    ops = D.ops
    n_rows, n_cols = 900, 100_000
    g = torch.Generator(device=D.dev).manual_seed(4)
    p = torch.nn.Parameter(torch.randn(n_rows, n_cols, device=D.dev, generator=g) * 0.05)
    p0 = p.detach().clone()
    opt = ops.FusedAdamW([{"params": [p], "weight_decay": 1e-2, "lr": 1e-3}])
    m = torch.zeros_like(p0); v = torch.zeros_like(p0); ref = p0.clone()
    for t in (1, 2):
        grad = torch.randn(n_rows, n_cols, device=D.dev, generator=g)
        p.grad = grad.clone()
        opt.step()
        ref.mul_(1 - 1e-3 * 1e-2)
        m.lerp_(grad, 0.1); v.mul_(0.999).addcmul_(grad, grad, value=0.001)
        ref.addcdiv_(m / (1 - 0.9 ** t), (v / (1 - 0.999 ** t)).sqrt() + 1e-8, value=-1e-3)
    assert float((p.detach() - ref).abs().max()) < 2e-6
