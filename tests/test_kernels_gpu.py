"""GPU parity tests of every C-ABI kernel against the CPU oracle / golden fixtures.

All calls go through libscpair_hip.so (ctypes, C ABI).  Tolerances:
  precision 'fp32' (ns=3): values rel 2e-5 + abs 1e-6 unless stated (fp32 summation-order noise)
  precision 'bf16' (ns=1): rel 2e-2 of the output scale (bf16 operand rounding, fp32 accumulate)
"""
import json

import numpy as np
import pytest
import torch

from tests import util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import scpair_amd as pkg
    import scpair_amd.loss, scpair_amd.model  # noqa: F401,E401
    import scpair_amd.ops as ops
    import scpair_amd.sparse as sparse
    from oracle import ref_cpu as O
    return type("S", (), dict(pkg=pkg, ops=ops, sparse=sparse, O=O, dev=torch.device("cuda:0")))


def relerr(a, b):
    a = a.detach().double().cpu(); b = torch.as_tensor(b).double().cpu().reshape(a.shape)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def close(a, b, rtol=2e-5, atol=1e-6, what=""):
    a = a.detach().double().cpu(); b = torch.as_tensor(b).double().cpu().reshape(a.shape)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), f"{what}: max err {float(err.max())} (scale {float(b.abs().max())}), worst tol ratio {float((err / tol).max())}"


def rand_csr(rng, n, f, density, binary=False, nan_frac=0.0):
    import scipy.sparse as sp
    m = sp.random(n, f, density=density, format="csr", random_state=np.random.RandomState(int(rng.integers(1 << 30))),
                  data_rvs=lambda k: rng.integers(1, 30, size=k).astype(np.float32))
    m.sort_indices()
    if binary:
        m.data[:] = 1.0
    if nan_frac > 0:
        m.data[rng.uniform(size=m.nnz) < nan_frac] = np.nan
    return m


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,f,b,density", [(300, 97, 130, 0.2), (1000, 2051, 257, 0.03), (64, 5, 64, 0.5), (10, 40, 3, 0.0)])
def test_tiles_build(S, n, f, b, density):
    rng = np.random.default_rng(0)
    m = rand_csr(rng, n, f, density, nan_frac=0.05 if density > 0.1 else 0)
    csr = S.sparse.DeviceCSR.from_scipy(m, S.dev)
    rows = rng.permutation(n)[:b]
    t = S.sparse.BatchTiles(csr, rows)
    dense = np.asarray(m[rows].todense(), dtype=np.float32)
    got = t.to_dense().cpu().numpy()
    np.testing.assert_array_equal(np.isnan(got), np.isnan(dense))
    np.testing.assert_array_equal(np.nan_to_num(got), np.nan_to_num(dense))
    np.testing.assert_allclose(t.lib.cpu().numpy(), np.nansum(dense, 1), rtol=1e-6)
    assert int(t.nnan.item()) == int(np.isnan(dense).sum())
    ptr = t.ptr.cpu().numpy()
    assert ptr[0] == 0 and ptr[((b + 63) // 64) * f] == t.nnz
    assert np.all(np.diff(ptr[: ((b + 63) // 64) * f + 1]) >= 0)


def test_tiles_implicit_ones(S):
    rng = np.random.default_rng(1)
    m = rand_csr(rng, 200, 333, 0.05, binary=True)
    csr = S.sparse.DeviceCSR.from_scipy(m, S.dev, binary=True)
    assert csr.values is None
    rows = np.arange(200)[::-1].copy()
    t = S.sparse.BatchTiles(csr, rows)
    np.testing.assert_array_equal(t.to_dense().cpu().numpy(), np.asarray(m[rows].todense(), dtype=np.float32))


@pytest.mark.parametrize("M,N,K", [(1024, 40, 900), (130, 900, 2000), (37, 5, 19), (64, 1, 40), (257, 130, 64), (8, 8, 8)])
@pytest.mark.parametrize("ns", [3, 1])
def test_gemm_nt(S, M, N, K, ns):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g); Bm = torch.randn(N, K, generator=g); bias = torch.randn(N, generator=g)
    Ad, Bd = A.to(S.dev), Bm.to(S.dev)
    C = torch.empty(M, N, device=S.dev)
    S.ops.gemm_nt(M, N, K, Ad, K, 1, Bd, K, 1, C, N, bias=bias.to(S.dev), ns=ns)
    if ns == 3:
        ref = A.double() @ Bm.double().t() + bias.double()
        close(C, ref, rtol=1e-5, atol=1e-5 * float(ref.abs().max()), what="gemm ns3")
    else:
        ref = A.bfloat16().double() @ Bm.bfloat16().double().t() + bias.double()
        close(C, ref, rtol=1e-5, atol=2e-5 * float(ref.abs().max()), what="gemm ns1 vs bf16-rounded inputs")
    # split-K slabs + reduce
    nsplit = S.ops._lib.lib().scp_gemm_split_count(K, 4)
    slabs = torch.empty(nsplit, M, N, device=S.dev)
    S.ops.gemm_nt(M, N, K, Ad, K, 1, Bd, K, 1, slabs, N, nsplit=nsplit, ns=3)
    ref = A.double() @ Bm.double().t()
    close(slabs.sum(0), ref, rtol=1e-5, atol=1e-5 * float(ref.abs().max()), what="gemm split-K")


@pytest.mark.parametrize("M,K,N", [(130, 77, 40), (1024, 900, 40), (50, 40, 1)])
def test_linear_autograd(S, M, K, N):
    S.ops.set_precision("fp32")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) * 0.1; b = torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g)
    xr, Wr, br = [t.clone().double().requires_grad_(True) for t in (x, W, b)]
    yr = xr @ Wr.t() + br
    gx, gW, gb = torch.autograd.grad((yr * dy.double()).sum(), [xr, Wr, br])
    xd, Wd, bd = [t.to(S.dev).requires_grad_(True) for t in (x, W, b)]
    y = S.ops.linear(xd, Wd, bd)
    (y * dy.to(S.dev)).sum().backward()
    s = lambda t: 1e-5 * float(t.abs().max())
    close(y, yr, rtol=1e-5, atol=s(yr)); close(xd.grad, gx, rtol=1e-5, atol=s(gx))
    close(Wd.grad, gW, rtol=1e-5, atol=s(gW)); close(bd.grad, gb, rtol=1e-5, atol=s(gb))


@pytest.mark.parametrize("B,F,H,C,ln,p_sparse", [(130, 333, 16, 0, True, True), (257, 2051, 900, 3, True, True),
                                                 (200, 1500, 132, 0, True, True),
                                                 (64, 64, 8, 0, False, True), (100, 50, 40, 4, True, False)])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_fcnn_layer(S, B, F, H, C, ln, p_sparse, prec):
    """Linear -> [LayerNorm] -> LeakyReLU on sparse (tile list) or dense input, fwd + all grads, vs torch fp64."""
    S.ops.set_precision(prec)
    rng = np.random.default_rng(B + F)
    m = rand_csr(rng, B + 20, F, 0.08)
    rows = rng.permutation(B + 20)[:B]
    xdense = torch.tensor(np.asarray(m[rows].todense(), dtype=np.float32))
    g = torch.Generator().manual_seed(3)
    W = torch.randn(H, F + C, generator=g) * (1.0 / np.sqrt(F)); b = torch.randn(H, generator=g) * 0.1
    gamma = 1 + 0.1 * torch.randn(H, generator=g); beta = 0.1 * torch.randn(H, generator=g)
    cov = torch.randn(B, C, generator=g) if C else None
    dout = torch.randn(B, H, generator=g)
    # reference in fp64 (bf16 mode: operands rounded like the kernel does)
    rnd = (lambda t: t.bfloat16().double()) if prec == "bf16" else (lambda t: t.double())
    Wr = W.double().requires_grad_(True); br = b.double().requires_grad_(True)
    gr = gamma.double().requires_grad_(True); ber = beta.double().requires_grad_(True)
    xin = xdense if cov is None else torch.cat([xdense, cov], 1)
    Wq = Wr if prec == "fp32" else (Wr + (rnd(W) - W.double()).detach())
    y = rnd(xin) @ Wq.t() + br
    if ln:
        y = torch.nn.functional.layer_norm(y, (H,), gr, ber)
    out_r = torch.nn.functional.leaky_relu(y, 0.01)
    params = [Wr, br] + ([gr, ber] if ln else [])
    grads_r = torch.autograd.grad((out_r * dout.double()).sum(), params)
    # device
    Wd, bd = W.to(S.dev).requires_grad_(True), b.to(S.dev).requires_grad_(True)
    gd = gamma.to(S.dev).requires_grad_(True) if ln else None
    bed = beta.to(S.dev).requires_grad_(True) if ln else None
    if p_sparse:
        x_in = S.sparse.BatchTiles(S.sparse.DeviceCSR.from_scipy(m, S.dev), rows)
    else:
        x_in = xdense.to(S.dev)
    out = S.ops.fcnn_layer(x_in, None if cov is None else cov.to(S.dev), Wd, bd, gd, bed, 0.01)
    (out * dout.to(S.dev)).sum().backward()
    tol = dict(rtol=2e-4, atol=2e-5) if prec == "fp32" else dict(rtol=2e-2, atol=2e-2)
    sc = lambda t: tol["atol"] * max(1.0, float(t.abs().max()))
    close(out, out_r, rtol=tol["rtol"], atol=sc(out_r), what="fcnn out")
    got = [Wd.grad, bd.grad] + ([gd.grad, bed.grad] if ln else [])
    for name, a, r_ in zip(["dW", "db", "dgamma", "dbeta"], got, grads_r):
        close(a, r_, rtol=tol["rtol"], atol=sc(r_), what=f"fcnn {name}")


def test_fcnn_layer_dropout_mask_consistency(S):
    """Dropout: kept fraction ~ 1-p, scale 1/(1-p), and backward uses the same mask as forward."""
    S.ops.set_precision("fp32")
    B, F, H, p = 256, 64, 128, 0.3
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, F, generator=g).to(S.dev).requires_grad_(True)
    W = (torch.randn(H, F, generator=g) * 0.2).to(S.dev).requires_grad_(True)
    b = torch.zeros(H, device=S.dev, requires_grad=True)
    out0 = S.ops.fcnn_layer(x, None, W, b, None, None, 1.0, 0.0, 7)         # slope 1 => identity activation
    out = S.ops.fcnn_layer(x, None, W, b, None, None, 1.0, p, 7)
    kept = (out != 0)
    frac = float(kept.float().mean())
    assert abs(frac - (1 - p)) < 0.02, frac
    close(out[kept], (out0 / (1 - p))[kept], rtol=1e-5, atol=1e-6)
    out.sum().backward()
    # d out / d b = column count of kept entries / (1-p)
    close(b.grad, kept.float().sum(0) / (1 - p), rtol=1e-5, atol=1e-4)


# ------------------------------------------------------------------------------------------
def _ber_case(S, d, tag, prec):
    c = U.cfg(d, f"{tag}/cfg")
    sd = U.sub(d, f"{tag}/sd")
    h = torch.tensor(d[f"{tag}/h"]); cov = U.opt(d, f"{tag}/cov")
    y = torch.tensor(d[f"{tag}/y"])
    lib, latlib = U.opt(d, f"{tag}/lib"), U.opt(d, f"{tag}/latlib")
    lrow = lib if c["sf"] else latlib
    hin = h if cov is None else torch.cat([h, cov], 1)
    hd = hin.to(S.dev).requires_grad_(True)
    Wd = sd["output_layer.weight"].to(S.dev).requires_grad_(True)
    bd = sd["output_layer.bias"].to(S.dev).requires_grad_(True)
    fsd = sd["feature_scalar"].to(S.dev).requires_grad_(True) if c["ff"] else None
    ld = lrow.to(S.dev).requires_grad_(True) if lrow is not None else None
    tiles = S.sparse.BatchTiles(S.sparse.DeviceCSR.from_dense(y, S.dev), np.arange(y.shape[0]))
    S.ops.set_precision(prec)
    loss = S.ops.ber_decoder_loss(hd, Wd, bd, fsd, ld, tiles)
    loss.backward()
    return c, loss, hd, Wd, bd, fsd, ld


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_ber_decoder_golden(S, prec):
    d = U.load("mod_ber")
    tol = dict(rtol=1e-4, atol=1e-7) if prec == "fp32" else dict(rtol=3e-2, atol=3e-4)
    for tag in json.loads(str(d["cases"])):
        c, loss, hd, Wd, bd, fsd, ld = _ber_case(S, d, tag, prec)
        close(loss, d[f"{tag}/loss"], rtol=1e-5 if prec == "fp32" else 5e-3, atol=0, what=f"{tag} loss")
        n_lat = d[f"{tag}/h"].shape[1]
        sc = lambda a: tol["atol"] + tol["rtol"] * float(np.abs(a).max())
        close(hd.grad[:, :n_lat], d[f"{tag}/dh"], rtol=tol["rtol"], atol=sc(d[f"{tag}/dh"]), what=f"{tag} dh")
        close(Wd.grad, d[f"{tag}/grad/output_layer.weight"], rtol=tol["rtol"], atol=sc(d[f"{tag}/grad/output_layer.weight"]), what=f"{tag} dW")
        close(bd.grad, d[f"{tag}/grad/output_layer.bias"], rtol=tol["rtol"], atol=sc(d[f"{tag}/grad/output_layer.bias"]), what=f"{tag} db")
        if c["ff"]:
            close(fsd.grad, d[f"{tag}/grad/feature_scalar"], rtol=tol["rtol"], atol=sc(d[f"{tag}/grad/feature_scalar"]), what=f"{tag} dfs")
        if f"{tag}/dlatlib" in d.files:
            close(ld.grad, d[f"{tag}/dlatlib"], rtol=tol["rtol"], atol=sc(d[f"{tag}/dlatlib"]), what=f"{tag} dlatlib")


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_nb_decoder_golden(S, prec):
    d = U.load("mod_raw")
    tol = dict(rtol=2e-4, atol=1e-7) if prec == "fp32" else dict(rtol=3e-2, atol=3e-4)
    n_done = 0
    for tag in json.loads(str(d["cases"])):
        c = U.cfg(d, f"{tag}/cfg")
        if c["lik"] != "nb":
            continue
        sd = U.sub(d, f"{tag}/sd")
        h = torch.tensor(d[f"{tag}/h"]); cov = U.opt(d, f"{tag}/cov"); y = torch.tensor(d[f"{tag}/y"])
        hin = h if cov is None else torch.cat([h, cov], 1)
        hd = hin.to(S.dev).requires_grad_(True)
        Wd = sd["output_mean.0.weight"].to(S.dev).requires_grad_(True)
        bd = sd["output_mean.0.bias"].to(S.dev).requires_grad_(True)
        fsd = sd["feature_scalar"].to(S.dev).requires_grad_(True) if c["ff"] else None
        thd = sd["output_inverse_dispersion"].to(S.dev).requires_grad_(True) if c["disp"] == "feature" else None
        if c["sf"]:
            L = torch.tensor(d[f"{tag}/lib"]).to(S.dev); lat = None
        else:
            lat = torch.tensor(d[f"{tag}/latlib"]).to(S.dev).requires_grad_(True)
            L = torch.exp(torch.nn.functional.leaky_relu(lat.detach(), 0.01))
        tiles = S.sparse.BatchTiles(S.sparse.DeviceCSR.from_dense(y, S.dev), np.arange(y.shape[0]))
        S.ops.set_precision(prec)
        loss = S.ops.nb_decoder_loss(hd, Wd, bd, fsd, thd, L, lat, tiles)
        loss.backward()
        close(loss, d[f"{tag}/loss"], rtol=2e-5 if prec == "fp32" else 5e-3, atol=0, what=f"{tag} loss")
        sc = lambda a: tol["atol"] + tol["rtol"] * float(np.abs(a).max())
        n_lat = h.shape[1]
        close(hd.grad[:, :n_lat], d[f"{tag}/dh"], rtol=tol["rtol"], atol=sc(d[f"{tag}/dh"]), what=f"{tag} dh")
        close(Wd.grad, d[f"{tag}/grad/output_mean.0.weight"], rtol=tol["rtol"], atol=sc(d[f"{tag}/grad/output_mean.0.weight"]), what=f"{tag} dW")
        close(bd.grad, d[f"{tag}/grad/output_mean.0.bias"], rtol=tol["rtol"], atol=sc(d[f"{tag}/grad/output_mean.0.bias"]), what=f"{tag} db")
        if c["ff"]:
            close(fsd.grad, d[f"{tag}/grad/feature_scalar"], rtol=tol["rtol"], atol=sc(d[f"{tag}/grad/feature_scalar"]), what=f"{tag} dfs")
        if c["disp"] == "feature":
            close(thd.grad, d[f"{tag}/grad/output_inverse_dispersion"], rtol=tol["rtol"], atol=sc(d[f"{tag}/grad/output_inverse_dispersion"]), what=f"{tag} dtheta")
        if not c["sf"]:
            close(lat.grad, d[f"{tag}/dlatlib"], rtol=tol["rtol"], atol=sc(d[f"{tag}/dlatlib"]), what=f"{tag} dlat")
        n_done += 1
    assert n_done >= 6


@pytest.mark.parametrize("B,F,K", [(130, 2051, 40), (257, 333, 11)])
def test_decoders_larger_vs_oracle(S, B, F, K):
    """Ragged sizes (cells not a multiple of 64, features not a multiple of 256) against the CPU oracle."""
    O = S.O
    S.ops.set_precision("fp32")
    rng = np.random.default_rng(F)
    g = torch.Generator().manual_seed(F)
    h = torch.randn(B, K, generator=g) * 0.5
    W = torch.randn(F, K, generator=g) * 0.3; b = torch.randn(F, generator=g) * 0.5 - 1.0
    lat = torch.randn(B, 1, generator=g)
    # Bernoulli with latent_lib
    yb = rand_csr(rng, B, F, 0.05, binary=True)
    ybd = torch.tensor(np.asarray(yb.todense(), dtype=np.float32))
    hr, Wr, br, lr = [t.clone().requires_grad_(True) for t in (h, W, b, lat)]
    res = O.output_ber({"output_layer.weight": Wr, "output_layer.bias": br}, hr, None, None, lr, sample_factor=False)
    lo = O.bce_masked(ybd, res[0]); lo.backward()
    hd, Wd, bd, ld = [t.clone().to(S.dev).requires_grad_(True) for t in (h, W, b, lat)]
    tiles = S.sparse.BatchTiles(S.sparse.DeviceCSR.from_scipy(yb, S.dev, binary=True), np.arange(B))
    l = S.ops.ber_decoder_loss(hd, Wd, bd, None, ld, tiles); l.backward()
    close(l, lo, rtol=2e-5, atol=0, what="ber loss")
    for nm, a, r_ in [("dh", hd.grad, hr.grad), ("dW", Wd.grad, Wr.grad), ("db", bd.grad, br.grad), ("dl", ld.grad, lr.grad)]:
        close(a, r_, rtol=2e-4, atol=2e-4 * float(r_.abs().max()), what=f"ber {nm}")
    # NB with library size
    yn = rand_csr(rng, B, F, 0.1)
    ynd = torch.tensor(np.asarray(yn.todense(), dtype=np.float32))
    lib = ynd.sum(1, keepdim=True)
    hr, Wr, br = [t.clone().requires_grad_(True) for t in (h, W, b + 1.0)]
    scale, theta, rate, _ = O.output_raw({"output_mean.0.weight": Wr, "output_mean.0.bias": br}, hr, None, lib, None, True)
    lo = O.raw_loss(ynd, rate, theta, None, "nb"); lo.backward()
    hd, Wd, bd = [t.clone().to(S.dev).requires_grad_(True) for t in (h, W, b + 1.0)]
    tiles = S.sparse.BatchTiles(S.sparse.DeviceCSR.from_scipy(yn, S.dev), np.arange(B))
    close(tiles.lib, lib, rtol=1e-6, atol=0)
    l = S.ops.nb_decoder_loss(hd, Wd, bd, None, None, tiles.lib, None, tiles); l.backward()
    close(l, lo, rtol=5e-5, atol=0, what="nb loss")
    for nm, a, r_ in [("dh", hd.grad, hr.grad), ("dW", Wd.grad, Wr.grad), ("db", bd.grad, br.grad)]:
        close(a, r_, rtol=5e-4, atol=5e-4 * float(r_.abs().max()), what=f"nb {nm}")


@pytest.mark.parametrize("B,F,K,sf", [(130, 2051, 40, True), (67, 333, 11, False), (5, 1000, 8, True)])
def test_zinb_row_loss_vs_oracle(S, B, F, K, sf):
    """ZINB predictor head (pre-activations materialised, fused row kernels) against the CPU oracle: loss and the
    gradients of the mean layer, the G x G zero-inflation layer, the latent input and the latent library size."""
    O = S.O
    S.ops.set_precision("fp32")
    rng = np.random.default_rng(F + 1)
    g = torch.Generator().manual_seed(F + 1)
    h = torch.randn(B, K, generator=g) * 0.5
    W = torch.randn(F, K, generator=g) * 0.3; b = torch.randn(F, generator=g) * 0.5
    Wp = torch.randn(F, F, generator=g) * (0.5 / F ** 0.5); bp = torch.randn(F, generator=g) * 0.5
    lat = torch.randn(B, 1, generator=g) + 4.0
    yn = rand_csr(rng, B, F, 0.1)
    ynd = torch.tensor(np.asarray(yn.todense(), dtype=np.float32))
    lib = ynd.sum(1, keepdim=True)
    ref = [t.clone().requires_grad_(True) for t in (h, W, b, Wp, bp, lat)]
    sd = {"output_mean.0.weight": ref[1], "output_mean.0.bias": ref[2], "output_dropout.weight": ref[3],
          "output_dropout.bias": ref[4]}
    _, theta, rate, pi = O.output_raw(sd, ref[0], None, lib if sf else None, None if sf else ref[5], sf)
    lo = O.raw_loss(ynd, rate, theta, pi, "zinb"); lo.backward()
    dec = S.pkg.model.Output_Module_Raw(F, 0, [K], infer_library_size=not sf, sample_factor=sf).to(S.dev)
    with torch.no_grad():
        dec.output_mean[0].weight.copy_(W); dec.output_mean[0].bias.copy_(b)
        dec.output_dropout.weight.copy_(Wp); dec.output_dropout.bias.copy_(bp)
    hd, ld = h.clone().to(S.dev).requires_grad_(True), lat.clone().to(S.dev).requires_grad_(True)
    tiles = S.sparse.BatchTiles(S.sparse.DeviceCSR.from_scipy(yn, S.dev), np.arange(B))
    l = dec.fused_loss(hd, None, tiles.lib.reshape(-1, 1) if sf else None, None if sf else ld, tiles, "zinb")
    assert l is not None
    l.backward()
    close(l, lo, rtol=5e-5, atol=0, what="zinb loss")
    pairs = [("dh", hd.grad, ref[0].grad), ("dW", dec.output_mean[0].weight.grad, ref[1].grad),
             ("db", dec.output_mean[0].bias.grad, ref[2].grad), ("dWpi", dec.output_dropout.weight.grad, ref[3].grad),
             ("dbpi", dec.output_dropout.bias.grad, ref[4].grad)]
    if not sf:
        pairs.append(("dlat", ld.grad, ref[5].grad))
    for nm, a, r_ in pairs:
        close(a, r_, rtol=5e-4, atol=5e-4 * float(r_.abs().max()), what=f"zinb {nm}")
    # the generic path (forward() + raw_loss) gives the same value
    with torch.no_grad():
        outs = dec(hd, None, tiles.lib.reshape(-1, 1) if sf else None, None if sf else ld)
        lg = S.pkg.loss.raw_loss(x=tiles.to_dense(), px_mu_rate=outs[2], px_theta=outs[1], px_mu_scale=outs[0],
                                 px_pi=outs[3], likelihood="zinb")
    close(lg, lo, rtol=5e-5, atol=0, what="zinb loss (generic path)")
    # precision mode 'bf16' (bf16 GEMM operands, hardware exp / log in the row kernels): loss within 1e-2 relative
    S.ops.set_precision("bf16")
    try:
        for prm in dec.parameters():
            prm.grad = None
        hb = h.clone().to(S.dev).requires_grad_(True)
        lb = dec.fused_loss(hb, None, tiles.lib.reshape(-1, 1) if sf else None, None if sf else ld.detach(), tiles, "zinb")
        lb.backward()
        close(lb, lo, rtol=1e-2, atol=0, what="zinb loss (bf16 mode)")
        gb, gr = dec.output_mean[0].bias.grad, ref[2].grad
        assert relerr(gb, gr) < 5e-2
    finally:
        S.ops.set_precision("fp32")


@pytest.mark.parametrize("ff,disp", [(True, "feature-cell"), (False, "feature"), (True, "feature")])
@pytest.mark.parametrize("sf", [True, False])
def test_zinb_row_loss_feature_parameters_vs_oracle(S, ff, disp, sf):
    """ZINB predictor head with feature_factor=True and / or dispersion='feature' (model.py:239-252) on the fused row
    kernels against the CPU oracle: loss, gradients of both Linear layers, of feature_scalar and of the per-feature
    dispersion (column sums taken in fixed row chunks), of the latent input and the latent library size.  Tolerance:
    loss 5e-5 relative, gradients 5e-4 of their largest element (fp32 mode; accumulation order differs)."""
    O = S.O
    S.ops.set_precision("fp32")
    B, F, K = 70, 517, 12
    rng = np.random.default_rng(F + 7)
    g = torch.Generator().manual_seed(F + 7 + int(ff) + 2 * int(sf))
    h = torch.randn(B, K, generator=g) * 0.5
    W = torch.randn(F, K, generator=g) * 0.3; b = torch.randn(F, generator=g) * 0.5
    Wp = torch.randn(F, F, generator=g) * (0.5 / F ** 0.5); bp = torch.randn(F, generator=g) * 0.5
    fs0 = torch.randn(F, generator=g); th0 = torch.randn(F, generator=g) * 0.5
    lat = torch.randn(B, 1, generator=g) + 4.0
    yn = rand_csr(rng, B, F, 0.1)
    ynd = torch.tensor(np.asarray(yn.todense(), dtype=np.float32))
    lib = ynd.sum(1, keepdim=True)
    ref = [t.clone().requires_grad_(True) for t in (h, W, b, Wp, bp, lat, fs0, th0)]
    sd = {"output_mean.0.weight": ref[1], "output_mean.0.bias": ref[2], "output_dropout.weight": ref[3],
          "output_dropout.bias": ref[4]}
    if ff:
        sd["feature_scalar"] = ref[6]
    if disp == "feature":
        sd["output_inverse_dispersion"] = ref[7]
    _, theta, rate, pi = O.output_raw(sd, ref[0], None, lib if sf else None, None if sf else ref[5], sf, dispersion=disp)
    lo = O.raw_loss(ynd, rate, theta, pi, "zinb"); lo.backward()
    dec = S.pkg.model.Output_Module_Raw(F, 0, [K], infer_library_size=not sf, sample_factor=sf, feature_factor=ff,
                                        dispersion=disp).to(S.dev)
    with torch.no_grad():
        dec.output_mean[0].weight.copy_(W); dec.output_mean[0].bias.copy_(b)
        dec.output_dropout.weight.copy_(Wp); dec.output_dropout.bias.copy_(bp)
        if ff:
            dec.feature_scalar.copy_(fs0)
        if disp == "feature":
            dec.output_inverse_dispersion.copy_(th0)
    hd, ld = h.clone().to(S.dev).requires_grad_(True), lat.clone().to(S.dev).requires_grad_(True)
    tiles = S.sparse.BatchTiles(S.sparse.DeviceCSR.from_scipy(yn, S.dev), np.arange(B))
    l = dec.fused_loss(hd, None, tiles.lib.reshape(-1, 1) if sf else None, None if sf else ld, tiles, "zinb")
    assert l is not None, "the fused ZINB head must cover feature factor / feature dispersion"
    l.backward()
    close(l, lo, rtol=5e-5, atol=0, what="zinb loss")
    pairs = [("dh", hd.grad, ref[0].grad), ("dW", dec.output_mean[0].weight.grad, ref[1].grad),
             ("db", dec.output_mean[0].bias.grad, ref[2].grad), ("dWpi", dec.output_dropout.weight.grad, ref[3].grad),
             ("dbpi", dec.output_dropout.bias.grad, ref[4].grad)]
    if not sf:
        pairs.append(("dlat", ld.grad, ref[5].grad))
    if ff:
        pairs.append(("dfs", dec.feature_scalar.grad, ref[6].grad))
    if disp == "feature":
        pairs.append(("dtheta", dec.output_inverse_dispersion.grad, ref[7].grad))
    for nm, a, r_ in pairs:
        close(a, r_, rtol=5e-4, atol=5e-4 * float(r_.abs().max()), what=f"zinb {nm}")
    # the generic path (forward() + raw_loss) gives the same value, and a second backward the same bits
    with torch.no_grad():
        outs = dec(hd, None, tiles.lib.reshape(-1, 1) if sf else None, None if sf else ld)
        lg = S.pkg.loss.raw_loss(x=tiles.to_dense(), px_mu_rate=outs[2], px_theta=outs[1], px_mu_scale=outs[0],
                                 px_pi=outs[3], likelihood="zinb")
    close(lg, lo, rtol=5e-5, atol=0, what="zinb loss (generic path)")
    first = [a.clone() for _, a, _ in pairs]
    for prm in dec.parameters():
        prm.grad = None
    hd.grad = None; ld.grad = None
    dec.fused_loss(hd, None, tiles.lib.reshape(-1, 1) if sf else None, None if sf else ld, tiles, "zinb").backward()
    again = [hd.grad, dec.output_mean[0].weight.grad, dec.output_mean[0].bias.grad, dec.output_dropout.weight.grad,
             dec.output_dropout.bias.grad] + ([] if sf else [ld.grad]) + ([dec.feature_scalar.grad] if ff else []) + \
            ([dec.output_inverse_dispersion.grad] if disp == "feature" else [])
    for a_, b_ in zip(first, again):
        assert torch.equal(a_, b_), "not bitwise reproducible"
    # precision mode 'bf16': loss within 1e-2 relative
    S.ops.set_precision("bf16")
    try:
        hb = h.clone().to(S.dev).requires_grad_(True)
        lb = dec.fused_loss(hb, None, tiles.lib.reshape(-1, 1) if sf else None, None if sf else ld.detach(), tiles, "zinb")
        lb.backward()
        close(lb, lo, rtol=1e-2, atol=0, what="zinb loss (bf16 mode)")
    finally:
        S.ops.set_precision("fp32")


def test_ber_eval_mode_matches_train_loss(S):
    S.ops.set_precision("fp32")
    d = U.load("mod_ber")
    tag = json.loads(str(d["cases"]))[0]
    sd = U.sub(d, f"{tag}/sd")
    y = torch.tensor(d[f"{tag}/y"])
    tiles = S.sparse.BatchTiles(S.sparse.DeviceCSR.from_dense(y, S.dev), np.arange(y.shape[0]))
    with torch.no_grad():
        l = S.ops.ber_decoder_loss(torch.tensor(d[f"{tag}/h"]).to(S.dev), sd["output_layer.weight"].to(S.dev),
                                   sd["output_layer.bias"].to(S.dev), None, None, tiles)
    close(l, d[f"{tag}/loss"], rtol=1e-5, atol=0)


# ------------------------------------------------------------------------------------------
def test_mse_and_reparam(S):
    b = U.load("kat_bce_mse")
    x = torch.tensor(b["mse_x"]).to(S.dev); mu = torch.tensor(b["mse_mu"]).to(S.dev).requires_grad_(True)
    l = S.ops.mse_masked(x, mu); l.backward()
    close(l, b["mse_loss"], rtol=1e-6); close(mu.grad, b["mse_dmu"], rtol=1e-5, atol=1e-8)
    g = torch.Generator().manual_seed(0)
    m_ = torch.randn(33, 7, generator=g); s_ = torch.randn(33, 7, generator=g) * 3; e_ = torch.randn(33, 7, generator=g)
    s_[0, 0] = 25.0; s_[0, 1] = -30.0
    w = torch.randn(33, 7, generator=g)
    mr, sr = m_.clone().double().requires_grad_(True), s_.clone().double().requires_grad_(True)
    sig = torch.nn.functional.softplus(sr) + 1e-7
    z = mr + e_.double() * sig
    kl = S.O.kl_normal(mr, sig)
    (0.7 * kl + (z * w.double()).sum()).backward()
    md, sd_ = m_.to(S.dev).requires_grad_(True), s_.to(S.dev).requires_grad_(True)
    sigma, zd, kld = S.ops.reparam_kl(md, sd_, e_.to(S.dev))
    (0.7 * kld + (zd * w.to(S.dev)).sum()).backward()
    close(sigma, sig, rtol=1e-5, atol=1e-9); close(zd, z, rtol=1e-5, atol=1e-6); close(kld, kl, rtol=1e-5)
    close(md.grad, mr.grad, rtol=1e-5, atol=1e-6); close(sd_.grad, sr.grad, rtol=2e-5, atol=1e-6)


def test_adamw_golden(S):
    d = U.load("adamw_traj")
    pa = torch.nn.Parameter(torch.tensor(d["init_0"]).to(S.dev)); pb = torch.nn.Parameter(torch.tensor(d["init_1"]).to(S.dev))
    opt = S.ops.FusedAdamW([{"params": [pa], "weight_decay": 1e-2, "lr": 1e-3}, {"params": [pb], "weight_decay": 0, "lr": 1e-3}])
    for s in range(5):
        pa.grad = torch.tensor(d[f"g{s}_0"]).to(S.dev); pb.grad = torch.tensor(d[f"g{s}_1"]).to(S.dev)
        opt.step()
        close(pa, d[f"p{s}_0"], rtol=1e-6, atol=1e-7); close(pb, d[f"p{s}_1"], rtol=1e-6, atol=1e-7)
    close(opt.state[pa]["exp_avg"], d["m_final_0"], rtol=1e-5, atol=1e-8)
    close(opt.state[pa]["exp_avg_sq"], d["v_final_0"], rtol=5e-5, atol=1e-10)
    # large, oddly sized tensor vs torch.optim.AdamW on the CPU
    g = torch.Generator().manual_seed(1)
    p0 = torch.randn(100003, generator=g); gr = torch.randn(100003, generator=g)
    pc = torch.nn.Parameter(p0.clone()); oc = torch.optim.AdamW([pc], lr=1e-3, weight_decay=1e-9)
    pd = torch.nn.Parameter(p0.clone().to(S.dev)); od = S.ops.FusedAdamW([pd], lr=1e-3, weight_decay=1e-9)
    for _ in range(3):
        pc.grad = gr.clone(); oc.step()
        pd.grad = gr.clone().to(S.dev); od.step()
    close(pd, pc.detach(), rtol=1e-6, atol=1e-7)


def test_error_reporting(S):
    L = S.ops._lib.lib()
    rc = L.scp_gemm_nt(0, 1, 1, None, 1, 1, None, 1, 1, None, 1, None, 0, 1, 3, None)
    assert rc == -1 and b"positive" in L.scp_last_error()
    with pytest.raises(RuntimeError):
        S.ops.linear(torch.randn(4, 4), torch.randn(4, 4), None)        # CPU tensors are refused, no fallback


def test_bf16_shadow_tracks_weights(S):
    """The bf16 shadow used by the bf16 first-layer kernels equals bf16(W) after FusedAdamW steps and after torch
    modifies W in place (version counter)."""
    S.ops.set_precision("bf16")
    rng = np.random.default_rng(7)
    B, F, H = 130, 333, 24                       # F not a multiple of 8: padded leading dimension
    m = rand_csr(rng, B, F, 0.1)
    tiles = S.sparse.BatchTiles(S.sparse.DeviceCSR.from_scipy(m, S.dev), np.arange(B))
    W = torch.nn.Parameter((torch.randn(H, F) * 0.1).to(S.dev)); b = torch.nn.Parameter(torch.zeros(H, device=S.dev))
    opt = S.ops.FusedAdamW([W, b], lr=1e-2, weight_decay=0.0)
    for _ in range(3):
        out = S.ops.fcnn_layer(tiles, None, W, b, None, None, 0.01)
        opt.zero_grad(); out.square().mean().backward(); opt.step()
        sh = S.ops.shadow_of(W, create=False)
        assert sh is not None and sh.shape == (H, 384)       # rows padded to 128 bytes
        assert torch.equal(sh[:, :F], W.detach().bfloat16()) and bool((sh[:, F:] == 0).all())
    with torch.no_grad():
        W.mul_(0.5)
    out = S.ops.fcnn_layer(tiles, None, W, b, None, None, 0.01)
    assert torch.equal(S.ops.shadow_of(W, create=False)[:, :F], W.detach().bfloat16())
    ref = torch.nn.functional.leaky_relu(tiles.to_dense().bfloat16().double() @ W.detach().bfloat16().double().t() + b.detach().double(), 0.01)
    close(out, ref, rtol=1e-4, atol=1e-4 * float(ref.abs().max()))
    S.ops.set_precision("fp32")


@pytest.mark.parametrize("relu_in", [False, True])
def test_linear_wide_gemm_bf16(S, relu_in):
    """Wide plain nn.Linear in bf16 mode (M*N*K >= 2^29: ZINB G x G layer model.py:228,255-256, VAE 900<->G layers
    pairedVAE.py:111-119) runs on the hand-written wide MFMA GEMM (scp_gemm_bf16): forward (bias in the epilogue),
    dgrad (relu mask in the epilogue when the layer's input is relu(x), model.py:240), wgrad and bias gradient against
    fp64 matmuls of the bf16-rounded operands (tolerance 1e-4 relative to the largest element: fp32 accumulation
    order); the shadow stays equal to bf16(W) across a FusedAdamW step."""
    S.ops.set_precision("bf16")
    g = torch.Generator().manual_seed(5)
    M, N, K = 300, 2050, 1801                      # not multiples of 8: padded leading dimensions
    assert M * N * K >= S.ops._LIB_GEMM_MIN
    x = torch.randn(M, K, generator=g).to(S.dev).requires_grad_(True)
    W = torch.nn.Parameter((torch.randn(N, K, generator=g) * 0.05).to(S.dev))
    b = torch.nn.Parameter(torch.randn(N, generator=g).to(S.dev))
    dy = torch.randn(M, N, generator=g).to(S.dev)
    y = S.ops.linear(x, W, b, relu_input=relu_in)
    y.backward(dy)
    xin = torch.relu(x.detach()) if relu_in else x.detach()
    xb, Wb, dyb = xin.bfloat16().double(), W.detach().bfloat16().double(), dy.bfloat16().double()
    sc = lambda t: 1e-4 * float(t.abs().max())
    close(y, xb @ Wb.t() + b.detach().double(), rtol=1e-4, atol=sc(y), what="wide gemm fwd")
    dx_ref = dyb @ Wb
    if relu_in:
        dx_ref = dx_ref * (x.detach().bfloat16() > 0).double()
        assert bool((x.grad[x.detach() <= 0] == 0).all())
    close(x.grad, dx_ref, rtol=1e-4, atol=sc(x.grad), what="wide gemm dgrad")
    close(W.grad, dyb.t() @ xb, rtol=1e-4, atol=sc(W.grad), what="wide gemm wgrad")
    close(b.grad, dy.double().sum(0), rtol=1e-5, atol=1e-4, what="wide gemm dbias")
    opt = S.ops.FusedAdamW([W, b], lr=1e-2, weight_decay=0.0)
    opt.step()
    sh = S.ops.shadow_of(W, create=False)
    assert sh is not None and torch.equal(sh[:, :K], W.detach().bfloat16())
    S.ops.set_precision("fp32")


def test_linear_wide_gemm_fused_adamw_equals_two_kernel_path(S):
    """FusedAdamW.fuse_weight_update on a wide nn.Linear (scp_gemm_bf16_adamw: AdamW step in the epilogue of the
    weight-gradient GEMM) against the ordinary path (scp_gemm_bf16 + scp_adamw_multi) over three steps: weights and
    moments within 2e-6 relative to the largest weight (hardware sqrt / reciprocal in the fused update, ~1 ulp each),
    the shadow exactly bf16(W), the same input gradient, and no .grad on the fused weight."""
    ops = S.ops
    ops.set_precision("bf16")
    g = torch.Generator().manual_seed(9)
    M, N, K = 300, 2052, 1804                      # ragged tiles in both dimensions, K % 4 == 0
    assert M * N * K >= ops._LIB_GEMM_MIN
    W0 = torch.randn(N, K, generator=g) * 0.05
    b0 = torch.randn(N, generator=g) * 0.1
    xs = [torch.randn(M, K, generator=g).to(S.dev) for _ in range(3)]
    dys = [torch.randn(M, N, generator=g).to(S.dev) * 0.01 for _ in range(3)]
    res = {}
    for mode in ("plain", "fused"):
        W = torch.nn.Parameter(W0.clone().to(S.dev)); b = torch.nn.Parameter(b0.clone().to(S.dev))
        opt = ops.FusedAdamW([{"params": [W], "weight_decay": 1e-2, "lr": 1e-2}, {"params": [b], "weight_decay": 0.0, "lr": 1e-2}])
        if mode == "fused":
            opt.fuse_weight_update(W)
        dxs = []
        for x, dy in zip(xs, dys):
            xr = x.clone().requires_grad_(True)
            opt.zero_grad()
            ops.linear(xr, W, b).backward(dy)
            if mode == "fused":
                assert W.grad is None
            dxs.append(xr.grad.clone())
            opt.step()
        st = opt.state[W]
        res[mode] = (W.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(), ops.shadow_of(W, create=False).clone(),
                     dxs, b.detach().clone(), st["step"])
    scale = float(W0.abs().max())
    for i, what in enumerate(("W", "exp_avg", "exp_avg_sq")):
        a_, r_ = res["fused"][i], res["plain"][i]
        tol = 2e-6 * max(scale if i == 0 else float(r_.abs().max()), 1e-30)
        assert float((a_ - r_).abs().max()) <= tol, (what, float((a_ - r_).abs().max()), tol)
    assert torch.equal(res["fused"][3][:, :K], res["fused"][0].bfloat16())
    # the second and third products read weights that differ by the update's rounding: bf16 shadows may flip one ulp
    assert torch.equal(res["fused"][4][0], res["plain"][4][0])
    for a_, r_ in zip(res["fused"][4][1:], res["plain"][4][1:]):
        close(a_, r_.double(), rtol=1e-3, atol=1e-3 * float(r_.abs().max()), what="dx after fused steps")
    assert torch.equal(res["fused"][5], res["plain"][5]) and res["fused"][6] == res["plain"][6] == 3
    assert float((res["fused"][0] - W0.to(S.dev)).abs().max()) > 1e-3
    ops.set_precision("fp32")


def test_linear_relu_input_small_shapes_equal_torch(S):
    """relu_input on the small-shape path (hand-written split-bf16 GEMM, fp32-equivalent) against torch autograd."""
    g = torch.Generator().manual_seed(6)
    x = torch.randn(50, 70, generator=g).to(S.dev).requires_grad_(True)
    W = torch.nn.Parameter((torch.randn(30, 70, generator=g) * 0.1).to(S.dev))
    b = torch.nn.Parameter(torch.randn(30, generator=g).to(S.dev))
    dy = torch.randn(50, 30, generator=g).to(S.dev)
    S.ops.linear(x, W, b, relu_input=True).backward(dy)
    got = [x.grad.clone(), W.grad.clone(), b.grad.clone()]
    x.grad = W.grad = b.grad = None
    torch.nn.functional.linear(torch.relu(x), W, b).backward(dy)
    for a_, r_ in zip(got, [x.grad, W.grad, b.grad]):
        close(a_, r_.double(), rtol=1e-4, atol=1e-5)


def test_tiles_to_dense(S):
    rng = np.random.default_rng(11)
    m = rand_csr(rng, 200, 517, 0.07)
    rows = rng.permutation(200)[:130]
    t = S.sparse.BatchTiles(S.sparse.DeviceCSR.from_scipy(m, S.dev), rows)
    np.testing.assert_array_equal(t.to_dense().cpu().numpy(), np.asarray(m[rows].todense(), dtype=np.float32))


@pytest.mark.parametrize("B,H", [(130, 77), (64, 900)])
def test_batch_norm_kernel(S, B, H):
    """scp_bn_fwd / scp_bn_bwd against the oracle's BatchNorm1d restatement (training statistics, running-statistics
    update, eval mode) and torch autograd of it on the CPU."""
    O = S.O
    g = torch.Generator().manual_seed(H)
    x = torch.randn(B, H, generator=g) * 2 + 0.5
    w = torch.rand(H, generator=g) + 0.5; b = torch.randn(H, generator=g)
    dout = torch.randn(B, H, generator=g)
    rm0 = torch.randn(H, generator=g) * 0.1; rv0 = torch.rand(H, generator=g) + 0.5
    for training in (True, False):
        xr, wr, br = [t.clone().requires_grad_(True) for t in (x, w, b)]
        sd = {"bn.weight": wr, "bn.bias": br, "bn.running_mean": rm0.clone(), "bn.running_var": rv0.clone(),
              "bn.num_batches_tracked": torch.tensor(0)}
        yr = O.batch_norm(xr, sd, "bn", training)
        (yr * dout).sum().backward()
        bn = torch.nn.BatchNorm1d(H).to(S.dev)
        with torch.no_grad():
            bn.weight.copy_(w); bn.bias.copy_(b); bn.running_mean.copy_(rm0); bn.running_var.copy_(rv0)
        xd = x.clone().to(S.dev).requires_grad_(True)
        yd = S.ops.batch_norm(xd, bn, training)
        (yd * dout.to(S.dev)).sum().backward()
        close(yd, yr, rtol=1e-5, atol=1e-5, what="bn y")
        close(xd.grad, xr.grad, rtol=1e-4, atol=2e-5, what="bn dx")
        close(bn.weight.grad, wr.grad, rtol=1e-4, atol=1e-4, what="bn dgamma")
        close(bn.bias.grad, br.grad, rtol=1e-4, atol=1e-4, what="bn dbeta")
        close(bn.running_mean, sd["bn.running_mean"], rtol=1e-5, atol=1e-6, what="bn running_mean")
        close(bn.running_var, sd["bn.running_var"], rtol=1e-5, atol=1e-6, what="bn running_var")
        assert int(bn.num_batches_tracked) == int(sd["bn.num_batches_tracked"])
    # dropout fused behind it: the backward recomputes the same mask
    bn = torch.nn.BatchNorm1d(H).to(S.dev)
    xd = x.clone().to(S.dev).requires_grad_(True)
    yd = S.ops.batch_norm(xd, bn, True, 0.3, 1234)
    keep = (yd != 0).float()
    assert 0.55 < float(keep.mean()) < 0.85
    y0 = S.ops.batch_norm(x.clone().to(S.dev), torch.nn.BatchNorm1d(H).to(S.dev), True)
    close(yd, y0 * keep / 0.7, rtol=1e-5, atol=1e-6, what="bn dropout scale")


@pytest.mark.parametrize("binary", [False, True])
@pytest.mark.parametrize("density", [0.35, 0.9])
def test_first_layer_bf16_dense_blocks(S, binary, density):
    """Blocks far denser than the per-thread entry slots of the bf16 first-layer kernels (overflow path: extra
    entries scattered by the slow loop, the LDS buffer cleared wholesale instead of un-scattered), mixed with sparse
    feature ranges, forward and weight gradient, several K-steps per slab."""
    S.ops.set_precision("bf16")
    rng = np.random.default_rng(int(density * 100) + binary)
    B, F, H = 300, 1500, 72
    import scipy.sparse as sp
    dense = rand_csr(rng, B, 700, density, binary=binary)
    thin = rand_csr(rng, B, F - 700, 0.02, binary=binary)
    m = sp.hstack([thin[:, :400], dense, thin[:, 400:]]).tocsr(); m.sort_indices()
    X = torch.tensor(np.asarray(m.todense(), dtype=np.float32))
    tiles = S.sparse.BatchTiles(S.sparse.DeviceCSR.from_scipy(m, S.dev, binary=binary), np.arange(B))
    g = torch.Generator().manual_seed(1)
    W = (torch.randn(H, F, generator=g) * 0.05).to(S.dev).requires_grad_(True)
    b = torch.zeros(H, device=S.dev, requires_grad=True)
    dout = torch.randn(B, H, generator=g).to(S.dev)
    for _ in range(2):                            # twice: buffers / tags of a previous launch must not leak
        W.grad = None
        out = S.ops.fcnn_layer(tiles, None, W, b, None, None, 1.0)
        out.backward(dout)
        Xb, Wb = X.bfloat16().double(), W.detach().cpu().bfloat16().double()
        ref = Xb @ Wb.t()
        close(out, ref, rtol=1e-4, atol=1e-4 * float(ref.abs().max()), what="dense-block fwd")
        ref_dW = dout.cpu().bfloat16().double().t() @ Xb
        close(W.grad, ref_dW, rtol=1e-4, atol=1e-4 * float(ref_dW.abs().max()), what="dense-block dW")
    S.ops.set_precision("fp32")


def test_foscttm_fosfttm_gpu(S):
    """scp_fos_counts behind scpair_amd.evaluator against the reference's evaluator outputs (golden) and, on a
    ragged larger case, against the oracle; counts are integers, so away from exact distance ties the results agree
    to the last bit of the final division."""
    from scpair_amd import evaluator as E
    g = U.load("eval_fos")
    assert abs(E.FOSCTTM(g["x"], g["y"]) - float(g["foscttm"])) < 1e-9
    f12, f21 = E.FOSCTTM(g["x"], g["y"], output_full=True)
    np.testing.assert_allclose(f12, g["f12_sorted"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(f21, g["f21_sorted"], rtol=0, atol=1e-12)
    assert abs(E.FOSFTTM(g["x"], g["y"]) - float(g["fosfttm"])) < 1e-9
    np.testing.assert_allclose(E.FOSFTTM(g["x"], g["y"], output_full=True), g["fosfttm_full"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(E.FOSFTTM(g["x"], g["y"], output_full=True, sort=True), g["fosfttm_sorted"], rtol=0, atol=1e-12)
    rng = np.random.default_rng(1)
    x = rng.normal(size=(1000, 70)).astype(np.float32); y = (x + 5 * rng.normal(size=x.shape)).astype(np.float32)
    av, _, _ = S.O.foscttm(x, y)
    assert abs(E.FOSCTTM(x, y) - av) < 2e-6       # fp32 summation order can flip a handful of near-ties in 1e6 pairs


@pytest.mark.parametrize("B,F,H,binary", [(300, 1000, 72, False), (1024, 4100, 200, True), (64, 256, 128, False),
                                          (300, 1000, 132, False), (2200, 900, 264, True)])
def test_first_layer_wgrad_with_adamw_epilogue(S, B, F, H, binary):
    """scp_spmm_dw_adamw_bf16 (weight gradient + AdamW step in one kernel) against the two-kernel path
    (scp_spmm_dw_bf16 then scp_adamw_multi) over three steps: weights, both moments and the bf16 shadow."""
    ops = S.ops
    ops.set_precision("bf16")
    try:
        rng = np.random.default_rng(F)
        x = rand_csr(rng, B, F, 0.08, binary=binary)
        tiles = S.sparse.BatchTiles(S.sparse.DeviceCSR.from_scipy(x, S.dev, binary=binary), np.arange(B))
        g = torch.Generator().manual_seed(F)
        W0 = torch.randn(H, F, generator=g) * 0.05
        b0 = torch.randn(H, generator=g) * 0.1
        res = {}
        for mode in ("plain", "fused"):
            W = torch.nn.Parameter(W0.clone().to(S.dev)); b = torch.nn.Parameter(b0.clone().to(S.dev))
            opt = ops.FusedAdamW([{"params": [W], "weight_decay": 1e-2, "lr": 1e-2}, {"params": [b], "weight_decay": 0.0, "lr": 1e-2}])
            if mode == "fused":
                opt.fuse_first_layer(W)
            gd = torch.Generator(device=S.dev).manual_seed(7)
            for step in range(3):
                out = ops.fcnn_layer(tiles, None, W, b, None, None, 0.01)
                dout = torch.randn(B, H, device=S.dev, generator=gd)
                opt.zero_grad()
                out.backward(dout)
                assert (W.grad is None) == (mode == "fused")
                opt.step()
            st = opt.state[W]
            assert st["step"] == 3
            res[mode] = [W.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(),
                         ops.shadow_of(W, create=False).clone(), b.detach().clone()]
        for nm, a, r_ in zip(("W", "exp_avg", "exp_avg_sq", "shadow", "bias"), res["fused"], res["plain"]):
            a, r_ = a.float(), r_.float()
            # fp32 state: same arithmetic, up to the compiler's fma contraction in the two kernels; the shadow may
            # then differ by one bf16 rounding step where a weight sits on a rounding boundary
            tol = 2.0 ** -7 if nm == "shadow" else 1e-6
            assert float((a - r_).abs().max()) <= tol * float(r_.abs().max()) + 1e-12, nm
        # the shadow is exactly the bf16 rounding of the master weights it was written with
        assert torch.equal(res["fused"][3][:, :F], res["fused"][0].bfloat16())
        # the weights really moved, and the guard against a second backward without step() works
        assert float((res["fused"][0] - W0.to(S.dev)).abs().max()) > 1e-3
        W = torch.nn.Parameter(W0.clone().to(S.dev))
        opt = ops.FusedAdamW([W], lr=1e-3)
        opt.fuse_first_layer(W)
        ops.fcnn_layer(tiles, None, W, None, None, None, 1.0).sum().backward()
        with pytest.raises(RuntimeError):
            ops.fcnn_layer(tiles, None, W, None, None, None, 1.0).sum().backward()
    finally:
        ops.set_precision("fp32")


@pytest.mark.parametrize("mode,scale", [("log1p", 1.0), ("log1p_norm", 1e4), ("log1p_norm", 37.5)])
def test_input_transform_on_tile_list(S, mode, scale):
    """Opt-in log1p / library-size normalisation of encoder inputs, applied to the minibatch tile list: against the
    oracle's formula on the dense rows; the resident CSR, the library sizes and the default (no transform) are
    unchanged; binary matrices with implicit ones are refused."""
    O = S.O
    rng = np.random.default_rng(5)
    B, F = 200, 777
    x = rand_csr(rng, 260, F, 0.07)
    x = x.tolil(); x[3, :] = 0; x = x.tocsr(); x.eliminate_zeros()          # an empty cell
    csr = S.sparse.DeviceCSR.from_scipy(x, S.dev)
    rows = rng.permutation(260)[:B]
    dense = torch.tensor(np.asarray(x.todense(), dtype=np.float32))[rows]
    raw = S.sparse.BatchTiles(csr, rows, want_lib=True)
    t = S.sparse.BatchTiles(csr.with_transform(mode, scale), rows, want_lib=True)
    close(t.to_dense(), O.input_transform(dense, mode, scale), rtol=2e-6, atol=1e-7, what=mode)
    assert torch.equal(t.lib.cpu(), dense.sum(1)) and torch.equal(raw.to_dense().cpu(), dense)
    nptr = ((B + 63) // 64) * F + 1
    assert torch.equal(t.ptr[:nptr], raw.ptr[:nptr])       # same buckets (entries inside a bucket carry no order)
    # feeds the first layer like any other tile list
    S.ops.set_precision("fp32")
    W = torch.randn(16, F) * 0.1
    out = S.ops.fcnn_layer(t, None, W.to(S.dev), None, None, None, 1.0)
    close(out, O.input_transform(dense, mode, scale) @ W.t(), rtol=1e-4, atol=1e-4, what="first layer on transformed input")
    with pytest.raises(ValueError):
        S.sparse.DeviceCSR.from_scipy(x, S.dev, binary=True).with_transform(mode, scale)
    with pytest.raises(ValueError):
        csr.with_transform("sqrt")


@pytest.mark.parametrize("lik,libmode", [("zinb", "lib"), ("nb", "latent"), ("zinb", "none")])
def test_vae_decoder_fused_loss_vs_oracle(S, lik, libmode):
    """VAE decoder heads (softmax mean, softplus dispersion, zero inflation; pairedVAE.py:147-187) through the fused row
    kernels against the CPU oracle: reconstruction loss and the gradient of every decoder parameter and input."""
    import scpair_amd as pkg
    O = S.O
    S.ops.set_precision("fp32")
    B, G, H, LAT = 70, 515, 24, 6
    rng = np.random.default_rng(11)
    torch.manual_seed(3)
    sf = libmode == "lib"
    dec = pkg.VAE_DECODER(G, 0, [H], LAT, infer_library_size=(libmode == "latent"), sample_factor=sf,
                          distribution="zinb", dispersion="feature-cell")
    with torch.no_grad():
        for prm in dec.parameters():
            if prm.ndim == 2:
                prm.mul_(0.5)
            else:
                prm.add_(torch.randn_like(prm) * 0.1)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in dec.state_dict().items()}
    dec = dec.to(S.dev).train()
    yn = rand_csr(rng, B, G, 0.1)
    ynd = torch.tensor(np.asarray(yn.todense(), dtype=np.float32))
    lib = ynd.sum(1, keepdim=True)
    z = torch.randn(B, LAT)
    lat = torch.randn(B, 1) * 0.5 + 3.0
    zr, lr = z.clone().requires_grad_(True), lat.clone().requires_grad_(True)
    _, theta, rate, pi = O.vae_decoder(sd, zr, None, lib if sf else None, lr if libmode == "latent" else None,
                                       distribution="zinb", sample_factor=sf)
    lo = O.raw_loss(ynd, rate, theta, pi if lik == "zinb" else None, lik); lo.backward()
    zd, ld = z.clone().to(S.dev).requires_grad_(True), lat.clone().to(S.dev).requires_grad_(True)
    tiles = S.sparse.BatchTiles(S.sparse.DeviceCSR.from_scipy(yn, S.dev), np.arange(B))
    l = dec.fused_loss(zd, None, tiles.lib.reshape(-1, 1) if sf else None, ld if libmode == "latent" else None, tiles, lik)
    assert l is not None
    l.backward()
    close(l, lo, rtol=5e-5, atol=0, what="vae recon loss")
    close(zd.grad, zr.grad, rtol=5e-4, atol=5e-4 * float(zr.grad.abs().max()), what="dz")
    if libmode == "latent":
        close(ld.grad, lr.grad, rtol=5e-4, atol=5e-4 * float(lr.grad.abs().max()), what="dlat")
    for k, prm in dec.named_parameters():
        ref = sd[k].grad
        if ref is None:
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, k
            continue
        close(prm.grad, ref, rtol=5e-4, atol=5e-4 * float(ref.abs().max()) + 1e-9, what=k)
    # same value as the generic path (forward() + raw_loss)
    with torch.no_grad():
        outs = dec(zd, None, tiles.lib.reshape(-1, 1) if sf else None, ld if libmode == "latent" else None)
        lg = pkg.loss.raw_loss(x=tiles.to_dense(), px_mu_rate=outs[2], px_theta=outs[1], px_mu_scale=outs[0],
                               px_pi=outs[3], likelihood=lik)
    close(lg, lo, rtol=5e-5, atol=0, what="generic path")


@pytest.mark.parametrize("B,F,H,binary", [(300, 1000, 72, False), (512, 4100, 200, True), (300, 1000, 132, True)])
def test_first_layer_wgrad_into_bf16_exchange_buffer(S, B, F, H, binary):
    """Data-parallel path in precision 'bf16': scp_spmm_dw_bf16_g16 writes dW as bf16 into the exchange buffer and
    scp_adamw_multi reads it from there (g_bf16) — against the fp32-gradient path with the gradient rounded to bf16
    before the optimiser step (what the cast into the bf16 bucket and the copy back did)."""
    ops = S.ops
    ops.set_precision("bf16")
    try:
        rng = np.random.default_rng(F + 2)
        x = rand_csr(rng, B, F, 0.08, binary=binary)
        tiles = S.sparse.BatchTiles(S.sparse.DeviceCSR.from_scipy(x, S.dev, binary=binary), np.arange(B))
        g = torch.Generator().manual_seed(F)
        W0 = torch.randn(H, F, generator=g) * 0.05
        res = {}
        for mode in ("plain", "sink"):
            W = torch.nn.Parameter(W0.clone().to(S.dev))
            opt = ops.FusedAdamW([W], lr=1e-2, weight_decay=1e-2)
            sink = None
            if mode == "sink":
                buf = torch.zeros(H * F, dtype=torch.bfloat16, device=S.dev)
                sink = ops.set_grad_sink(W, buf.view(H, F))
            gd = torch.Generator(device=S.dev).manual_seed(7)
            for step in range(3):
                out = ops.fcnn_layer(tiles, None, W, None, None, None, 0.01)
                dout = torch.randn(B, H, device=S.dev, generator=gd)
                opt.zero_grad()
                out.backward(dout)
                if mode == "sink":
                    assert W.grad is None and sink.written
                    gref = res["plain_g"][step]
                    assert torch.equal(sink.view, gref), "bf16 gradient in the exchange buffer"
                else:
                    res.setdefault("plain_g", []).append(W.grad.bfloat16())
                    W.grad = W.grad.bfloat16().float()
                opt.step(grad_scale=0.5)
                if mode == "sink":
                    assert not sink.written
            st = opt.state[W]
            res[mode] = [W.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone()]
            ops.set_grad_sink(W, None)
        for nm, a, r_ in zip(("W", "exp_avg", "exp_avg_sq"), res["sink"], res["plain"]):
            assert float((a - r_).abs().max()) <= 1e-6 * float(r_.abs().max()) + 1e-12, nm
    finally:
        ops.set_precision("fp32")


@pytest.mark.parametrize("B,F,H,binary", [(300, 1001, 132, False), (2200, 900, 264, True), (130, 515, 900, False)])
def test_first_layer_wgrad_tail_rows(S, B, F, H, binary):
    """Hidden widths of the form 128 k + r, r <= 8 (the reference's default 900 = 7 x 128 + 4): the last 128-row tile of
    the MFMA weight-gradient kernels is almost empty.  Every row of dW against the fp32 matmul of the bf16-rounded
    operands (the products are exact in fp32; 2e-5 of the scale is stated for the summation order), the last rows
    separately, and the result is bitwise reproducible.  (A variant that gave those rows to a vector-unit kernel and
    the MFMA launch only the full tiles passed this test and measured slower, 487 + 32 us against 502 us at C2: the
    launch is bound by the HBM stream of its epilogues, which the nearly empty tiles do not add to.)"""
    ops = S.ops
    ops.set_precision("bf16")
    try:
        rng = np.random.default_rng(H)
        x = rand_csr(rng, B, F, 0.08, binary=binary)
        tiles = S.sparse.BatchTiles(S.sparse.DeviceCSR.from_scipy(x, S.dev, binary=binary), np.arange(B))
        X = tiles.to_dense().bfloat16().float()
        g = torch.Generator(device=S.dev).manual_seed(H)
        W = (torch.randn(H, F, device=S.dev, generator=g) * 0.05).requires_grad_(True)
        dY = torch.randn(B, H, device=S.dev, generator=g)
        out = ops.fcnn_layer(tiles, None, W, None, None, None, 1.0)
        (g1,) = torch.autograd.grad(out, W, dY, retain_graph=True)
        (g2,) = torch.autograd.grad(out, W, dY)
        ref = dY.bfloat16().float().t() @ X
        err = float((g1 - ref).abs().max()) / float(ref.abs().max())
        assert err < 2e-5, err
        tail = H % 128
        assert 0 < tail <= 8
        errt = float((g1[H - tail:] - ref[H - tail:]).abs().max()) / float(ref.abs().max())
        assert errt < 2e-5, errt
        assert torch.equal(g1, g2)
    finally:
        ops.set_precision("fp32")


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_decoders_with_many_covariate_levels_take_the_materialised_path(S, prec):
    """Contraction depth K = latent width + one-hot covariate columns beyond what the fused decoders hold (64 with bf16
    MFMA inputs, 48 in the fp32-equivalent mode): Output_Module_Ber / Output_Module_Raw.fused_loss() decline and the
    step goes through the materialised forward + the dense likelihood kernels, against the CPU oracle."""
    O = S.O
    import scpair_amd as A
    S.ops.set_precision(prec)
    try:
        B, F, lat_w, C = 70, 300, 40, 60                       # K = 100
        rng = np.random.default_rng(4)
        g = torch.Generator().manual_seed(4)
        z = torch.randn(B, lat_w, generator=g) * 0.5
        cov = torch.nn.functional.one_hot(torch.randint(0, C, (B,), generator=g), C).float()
        assert not S.ops.decoder_depth_supported(lat_w + C) and S.ops.decoder_depth_supported(40)
        tol = 2e-5 if prec == "fp32" else 2e-2
        # Bernoulli
        dec = A.Output_Module_Ber(F, C, [900, lat_w], infer_library_size=False, sample_factor=False, feature_factor=False).to(S.dev)
        yb = rand_csr(rng, B, F, 0.05, binary=True)
        ybd = torch.tensor(np.asarray(yb.todense(), dtype=np.float32))
        tiles = S.sparse.BatchTiles(S.sparse.DeviceCSR.from_scipy(yb, S.dev, binary=True), np.arange(B))
        zd = z.clone().to(S.dev).requires_grad_(True)
        assert dec.fused_loss(zd, cov.to(S.dev), None, None, tiles, "ber") is None
        enc = lambda x, c: (zd, None)
        loss = A.utils._predictor_loss(None, tiles, None, enc, dec, "ber", True, None, cov.to(S.dev))
        loss.backward()
        sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in dec.state_dict().items()}
        zr = z.clone().requires_grad_(True)
        res = O.output_ber(sd, zr, cov, None, None, sample_factor=False)
        lo = O.bce_masked(ybd, res[0]); lo.backward()
        close(loss, lo, rtol=tol, atol=0, what="ber loss, K = 100")
        close(zd.grad, zr.grad, rtol=10 * tol, atol=10 * tol * float(zr.grad.abs().max()), what="ber dz")
        close(dec.output_layer.weight.grad, sd["output_layer.weight"].grad, rtol=10 * tol,
              atol=10 * tol * float(sd["output_layer.weight"].grad.abs().max()), what="ber dW")
        # NB with the observed library size
        decn = A.Output_Module_Raw(F, C, [900, lat_w], infer_library_size=False, sample_factor=True, zero_inflated=False,
                                   dispersion="feature").to(S.dev)
        yn = rand_csr(rng, B, F, 0.1)
        ynd = torch.tensor(np.asarray(yn.todense(), dtype=np.float32))
        tn = S.sparse.BatchTiles(S.sparse.DeviceCSR.from_scipy(yn, S.dev), np.arange(B))
        zd = z.clone().to(S.dev).requires_grad_(True)
        assert decn.fused_loss(zd, cov.to(S.dev), tn.lib.reshape(-1, 1), None, tn, "nb") is None
        loss = A.utils._predictor_loss(None, tn, tn.lib.reshape(-1, 1), enc, decn, "nb", True, None, cov.to(S.dev))
        loss.backward()
        sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in decn.state_dict().items()}
        zr = z.clone().requires_grad_(True)
        scale, theta, rate, _ = O.output_raw(sd, zr, cov, ynd.sum(1, keepdim=True), None, True, dispersion="feature")
        lo = O.raw_loss(ynd, rate, theta, None, "nb"); lo.backward()
        close(loss, lo, rtol=max(tol, 5e-5), atol=0, what="nb loss, K = 100")
        close(zd.grad, zr.grad, rtol=10 * tol, atol=10 * tol * float(zr.grad.abs().max()), what="nb dz")
    finally:
        S.ops.set_precision("fp32")
