"""The fused encoder branch (first-layer product on the tile list + scp_enc_tail_fwd / scp_enc_tail_bwd) against an fp64
torch restatement of the same modules (utils.py:54-64 FCNN layers + the Linear behind them, model.py:77-92, 124-143):
output and the gradient of every parameter, at the bench's hidden sizes and at ragged ones, for the three kinds of
branch (extra_linear_rep: N3 = H2; library head: N3 = 1; no final Linear), in both precision modes; dropout masks are
checked for forward / backward consistency by directional derivatives of the kernel itself."""
import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _csr(rng, n, f, density):
    import scipy.sparse as sp
    m = sp.random(n, f, density=density, format="csr", random_state=np.random.RandomState(int(rng.integers(1 << 30))),
                  data_rvs=lambda k: rng.integers(1, 9, size=k).astype(np.float32))
    m.sort_indices()
    return m


def _params(g, F, H1, H2, N3):
    P = dict(W1=torch.randn(H1, F, generator=g) / np.sqrt(F) * 3, b1=torch.randn(H1, generator=g) * 0.1,
             g1=1 + 0.1 * torch.randn(H1, generator=g), be1=0.1 * torch.randn(H1, generator=g),
             W2=torch.randn(H2, H1, generator=g) / np.sqrt(H1), b2=torch.randn(H2, generator=g) * 0.1,
             g2=1 + 0.1 * torch.randn(H2, generator=g), be2=0.1 * torch.randn(H2, generator=g))
    if N3:
        P["W3"] = torch.randn(N3, H2, generator=g) / np.sqrt(H2)
        P["b3"] = torch.randn(N3, generator=g) * 0.1
    return P


def _ref(P, x, rnd, keep1=None, keep2=None, cov=None):
    """fp64 forward (cov: covariate block the reference concatenates to the input, model.py:128; its weight columns
    stay exact fp32 in both precision modes); `rnd` rounds the first-layer operands the way the precision mode does; keep1 / keep2: dropout
    factors (0 or 1/(1-p)) of the two layers.  Also returns the smallest |pre-activation|: an element closer to the
    LeakyReLU kink than the fp32 / fp64 difference can sit on the other side of it in the kernel, which changes its
    gradient by a factor 100 -- the tests draw their parameters so that no element is that close."""
    q = {k: v.double().requires_grad_(True) for k, v in P.items()}
    F = x.shape[1]
    W1 = q["W1"][:, :F] + (rnd(P["W1"][:, :F]) - P["W1"][:, :F].double()).detach()
    y = rnd(x) @ W1.t() + q["b1"]
    if cov is not None:
        y = y + cov.double() @ q["W1"][:, F:].t()
    z1 = Fn.layer_norm(y, (y.shape[1],), q["g1"], q["be1"])
    a1 = Fn.leaky_relu(z1, 0.01)
    if keep1 is not None:
        a1 = a1 * keep1
    z2 = Fn.layer_norm(a1 @ q["W2"].t() + q["b2"], (q["W2"].shape[0],), q["g2"], q["be2"])
    a2 = Fn.leaky_relu(z2, 0.01)
    if keep2 is not None:
        a2 = a2 * keep2
    out = a2 @ q["W3"].t() + q["b3"] if "W3" in q else a2
    return q, out, min(float(z1.detach().abs().min()), float(z2.detach().abs().min()))


def _draw(F, H1, H2, N3, x, rnd, seed0, keep1=None, keep2=None, cov=None):
    g = torch.Generator().manual_seed(seed0)
    P = _params(g, F + (0 if cov is None else cov.shape[1]), H1, H2, N3)
    q, out_r, zmin = _ref(P, x, rnd, keep1, keep2, cov)
    return g, P, q, out_r


def _check(name, got, ref, rt):
    """Every element within rt * scale -- except the few a LeakyReLU kink explains: a pre-activation within fp32 noise
    of zero (a handful among ~1e6) can sit on the other side of the kink in the kernel, which changes that one
    element's gradient by a factor 100 and with it one column of the per-column gradients / one row of dW1.  Such
    outliers are allowed for at most 0.1 % of the elements (3 of a short vector) and must stay below 30 % of the scale; an error of the
    kernel shows everywhere."""
    got, ref = got.detach().double().cpu().reshape(-1), ref.detach().double().reshape(-1)
    scale = max(1e-2, float(ref.abs().max()))
    err = (got - ref).abs()
    bad = int((err > rt * scale).sum())
    assert bad <= max(3, 1e-3 * err.numel()) and float(err.max()) <= 0.3 * scale, (name, bad, float(err.max()), scale)


@pytest.mark.parametrize("B,F,H1,H2,N3", [(1024, 2000, 900, 40, 40), (1024, 2000, 900, 40, 1), (130, 333, 16, 8, 0),
                                          (67, 515, 1024, 64, 64), (9, 100, 4, 1, 1), (257, 700, 800, 30, 30)])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_encoder_branch_vs_fp64(B, F, H1, H2, N3, prec):
    import scpair_amd as S
    from scpair_amd import ops
    S.set_precision(prec)
    try:
        rng = np.random.default_rng(B + F)
        m = _csr(rng, B + 11, F, 0.06)
        rows = rng.permutation(B + 11)[:B]
        x = torch.tensor(np.asarray(m[rows].todense(), dtype=np.float32))
        rnd = (lambda t: t.bfloat16().double()) if prec == "bf16" else (lambda t: t.double())
        g, P, q, out_r = _draw(F, H1, H2, N3, x, rnd, H1 + N3)
        wout = torch.randn(B, N3 if N3 else H2, generator=g)
        (out_r * wout.double()).sum().backward()
        d = {k: v.clone().to(DEV).requires_grad_(True) for k, v in P.items()}
        tiles = S.BatchTiles(S.DeviceCSR.from_scipy(m, DEV), rows)
        assert ops.enc_branch_supported(tiles, None, d["W1"], d["W2"], d.get("W3"), True, True)
        out = ops.enc_branch(tiles, d["W1"], d["b1"], d["g1"], d["be1"], d["W2"], d["b2"], d["g2"], d["be2"], d.get("W3"),
                             d.get("b3"), 0.01)
        (out * wout.to(DEV)).sum().backward()
        # fp32 mode: the tail is exact fp32 and the first layer fp32-equivalent; bf16 mode: only the first-layer
        # operands are rounded (the reference above rounds them the same way)
        rt = 2e-4 if prec == "fp32" else 2e-3
        _check("out", out, out_r, rt)
        for k in P:
            assert d[k].grad is not None, k
            _check(k, d[k].grad, q[k].grad, rt * (10 if (k == "W1" and prec == "bf16") else 1))   # bf16: dY^T is rounded too
        # inference (no saved tensors) gives the same output
        with torch.no_grad():
            out_e = ops.enc_branch(tiles, d["W1"], d["b1"], d["g1"], d["be1"], d["W2"], d["b2"], d["g2"], d["be2"],
                                   d.get("W3"), d.get("b3"), 0.01)
        assert torch.equal(out_e, out.detach())
    finally:
        S.set_precision("fp32")


@pytest.mark.parametrize("B,F,H1,H2,N3,C", [(1024, 2000, 900, 40, 40, 0), (203, 515, 128, 8, 1, 0), (67, 333, 16, 8, 0, 5),
                                            (9, 100, 4, 1, 1, 0), (300, 1001, 800, 30, 30, 4)])
def test_binary_input_fp32_mode_three_term_path(B, F, H1, H2, N3, C, monkeypatch):
    """fp32-equivalent mode with implicit-ones data (binarised peaks): the first layer runs three launches of the one-term
    bf16 kernel on the hi / mid / lo planes of the weights (scp_split3_bf16) and its weight gradient one launch over the
    minibatch repeated three times with dY split the same way (scp_split3_transpose_bf16) -- against the fp64 formula at
    the fp32 tolerance of test_encoder_branch_vs_fp64, and against the general 6-product kernel (SCPAIR_FP32_GENERAL=1)."""
    import scipy.sparse as sp
    import scpair_amd as S
    from scpair_amd import ops
    S.set_precision("fp32")
    rng = np.random.default_rng(B + F)
    m = sp.random(B + 11, F, density=0.06, format="csr", random_state=np.random.RandomState(5), dtype=np.float32)
    m.data[:] = 1.0
    m.sort_indices()
    rows = rng.permutation(B + 11)[:B]
    x = torch.tensor(np.asarray(m[rows].todense(), dtype=np.float32))
    gc = torch.Generator().manual_seed(3)
    cov = torch.randn(B, C, generator=gc) if C else None
    g, P, q, out_r = _draw(F, H1, H2, N3, x, lambda t: t.double(), H1 + N3, cov=cov)
    wout = torch.randn(B, N3 if N3 else H2, generator=g)
    (out_r * wout.double()).sum().backward()
    csr = S.DeviceCSR.from_scipy(m, DEV, binary=True)
    assert csr.values is None
    res = {}
    for mode in ("three-term", "general"):
        if mode == "general":
            monkeypatch.setenv("SCPAIR_FP32_GENERAL", "1")
        d = {k: v.clone().to(DEV).requires_grad_(True) for k, v in P.items()}
        tiles = S.BatchTiles(csr, rows)
        br = dict(W1=d["W1"], b1=d["b1"], g1=d["g1"], be1=d["be1"], W2=d["W2"], b2=d["b2"], g2=d["g2"], be2=d["be2"],
                  W3=d.get("W3"), b3=d.get("b3"))
        out = ops.enc_branches(tiles, [br], 0.01, cov=None if cov is None else cov.to(DEV))[0]
        (out * wout.to(DEV)).sum().backward()
        _check("out/" + mode, out, out_r, 2e-4)
        for k in P:
            _check(k + "/" + mode, d[k].grad, q[k].grad, 2e-4)
        with torch.no_grad():
            out_e = ops.enc_branches(tiles, [br], 0.01, cov=None if cov is None else cov.to(DEV))[0]
        assert torch.equal(out_e, out.detach())
        res[mode] = (out.detach(), d["W1"].grad.detach())
    _check("out three-term vs general", res["three-term"][0], res["general"][0].cpu(), 2e-5)
    _check("dW1 three-term vs general", res["three-term"][1], res["general"][1].cpu(), 2e-5)


def _mask(ops, seed, B, H, p):
    """The dropout factors keep_scale(seed, row * H + col, p) of a [B, H] layer, read off the generic FCNN layer kernel
    (same counter-based hash): input 1, weight 1, no LayerNorm, identity activation."""
    one = torch.ones(B, 1, device=DEV)
    return ops.fcnn_layer(one, None, torch.ones(H, 1, device=DEV), None, None, None, 1.0, p, seed).double().cpu()


@pytest.mark.parametrize("N3", [5, 0])
def test_encoder_branch_with_dropout(N3):
    """Dropout on both layers: forward and every gradient against the fp64 restatement that applies the very masks the
    kernels draw (the backward recomputes them from the seeds); same seeds give the same output, other seeds another."""
    import scpair_amd as S
    from scpair_amd import ops
    S.set_precision("fp32")
    B, F, H1, H2 = 203, 400, 132, 16
    p1, p2, seed1, seed2 = 0.3, 0.2, 1234, 5678
    rng = np.random.default_rng(1)
    m = _csr(rng, B, F, 0.1)
    x = torch.tensor(np.asarray(m.todense(), dtype=np.float32))
    tiles = S.BatchTiles(S.DeviceCSR.from_scipy(m, DEV), np.arange(B))
    keep1, keep2 = _mask(ops, seed1, B, H1, p1), _mask(ops, seed2, B, H2, p2)
    assert 0.65 < float((keep1 > 0).double().mean()) < 0.75 and 0.74 < float((keep2 > 0).double().mean()) < 0.86
    g, P, q, out_r = _draw(F, H1, H2, N3, x, lambda t: t.double(), 2, keep1, keep2)
    wout = torch.randn(B, N3 if N3 else H2, generator=g)
    (out_r * wout.double()).sum().backward()

    def run(s1, s2, grad):
        d = {k: v.clone().to(DEV).requires_grad_(grad) for k, v in P.items()}
        out = ops.enc_branch(tiles, d["W1"], d["b1"], d["g1"], d["be1"], d["W2"], d["b2"], d["g2"], d["be2"],
                             d.get("W3"), d.get("b3"), 0.01, p1, p2, seeds=(s1, s2))
        return d, out
    d, out = run(seed1, seed2, True)
    (out * wout.to(DEV)).sum().backward()
    _check("out", out, out_r, 2e-4)
    for k in P:
        _check(k, d[k].grad, q[k].grad, 2e-4)
    assert torch.equal(run(seed1, seed2, False)[1], out.detach())
    assert not torch.equal(run(99, 98, False)[1], out.detach())


@pytest.mark.parametrize("prec,pack", [("fp32", False), ("bf16", False), ("bf16", True)])
@pytest.mark.parametrize("B", [1024, 203])
def test_two_branches_in_one_launch(prec, pack, B):
    """hidden_rep (+ extra_linear_rep, N3 = H2) and hidden_lib (+ head, N3 = 1) of one encoder through ops.enc_branches:
    one tail launch each way for both; with packed first-layer weights (precision 'bf16') also one first-layer product
    and one weight-gradient launch.  Every output and gradient against the fp64 restatement of each branch."""
    import scpair_amd as S
    from scpair_amd import ops
    S.set_precision(prec)
    try:
        F, H1, H2 = 1000, 900, 40
        rng = np.random.default_rng(B)
        m = _csr(rng, B + 5, F, 0.06)
        rows = rng.permutation(B + 5)[:B]
        x = torch.tensor(np.asarray(m[rows].todense(), dtype=np.float32))
        rnd = (lambda t: t.bfloat16().double()) if prec == "bf16" else (lambda t: t.double())
        refs, devs, wouts = [], [], []
        for bi, N3 in enumerate((H2, 1)):
            g, P, q, out_r = _draw(F, H1, H2, N3, x, rnd, 50 + bi)
            wout = torch.randn(B, N3, generator=g)
            (out_r * wout.double()).sum().backward()
            refs.append((P, q, out_r))
            wouts.append(wout)
            devs.append({k: v.clone().to(DEV).requires_grad_(True) for k, v in P.items()})
        if pack:
            assert ops.pack_first_layers(devs[0]["W1"], devs[1]["W1"])
            assert devs[1]["W1"].data_ptr() == devs[0]["W1"].data_ptr() + 4 * H1 * devs[0]["W1"].stride(0)
            assert torch.equal(devs[1]["W1"].detach().cpu(), refs[1][0]["W1"])
        tiles = S.BatchTiles(S.DeviceCSR.from_scipy(m, DEV), rows)
        outs = ops.enc_branches(tiles, devs, 0.01)
        sum((o * w.to(DEV)).sum() for o, w in zip(outs, wouts)).backward()
        rt = 2e-4 if prec == "fp32" else 2e-3
        for bi in range(2):
            P, q, out_r = refs[bi]
            _check(f"out{bi}", outs[bi], out_r, rt)
            for k in P:
                assert devs[bi][k].grad is not None, (bi, k)
                _check(f"{k}{bi}", devs[bi][k].grad, q[k].grad, rt * (10 if (k == "W1" and prec == "bf16") else 1))
    finally:
        S.set_precision("fp32")


def test_packed_pair_fused_adamw_matches_separate_weights():
    """Two steps of the fused weight-gradient + AdamW launch over a packed pair against the same two steps with
    unpacked weights (two launches): identical parameters (same kernel arithmetic, other tiling of the rows)."""
    import scpair_amd as S
    from scpair_amd import ops
    S.set_precision("bf16")
    try:
        B, F, H1, H2 = 256, 600, 132, 16
        rng = np.random.default_rng(9)
        m = _csr(rng, B, F, 0.1)
        tiles = S.BatchTiles(S.DeviceCSR.from_scipy(m, DEV), np.arange(B))
        x = torch.tensor(np.asarray(m.todense(), dtype=np.float32))
        res = []
        for pack in (False, True):
            devs = []
            for bi, N3 in enumerate((H2, 1)):
                g, P, q, out_r = _draw(F, H1, H2, N3, x, lambda t: t.double(), 70 + bi)
                devs.append({k: torch.nn.Parameter(v.clone().to(DEV)) for k, v in P.items()})
            if pack:
                assert ops.pack_first_layers(devs[0]["W1"], devs[1]["W1"])
            opt = S.FusedAdamW([{"params": [p for d in devs for p in d.values()], "weight_decay": 1e-2, "lr": 1e-2}])
            opt.fuse_first_layer(devs[0]["W1"], devs[1]["W1"])
            for _ in range(2):
                opt.zero_grad()
                outs = ops.enc_branches(tiles, devs, 0.01)
                sum((o * o).sum() for o in outs).backward()
                opt.step()
            res.append([{k: v.detach().clone() for k, v in d.items()} for d in devs])
        for a, b in zip(*res):
            for k in a:
                assert torch.allclose(a[k], b[k], rtol=1e-5, atol=1e-7), k
    finally:
        S.set_precision("fp32")


@pytest.mark.parametrize("B,F,H1,H2,N3,C", [(1024, 2000, 900, 40, 40, 4), (130, 333, 16, 8, 0, 7), (257, 700, 800, 30, 1, 64),
                                            (67, 516, 1024, 64, 64, 1)])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_encoder_branch_with_covariates(B, F, H1, H2, N3, C, prec):
    """The reference's default encoder input torch.cat([x, cov]) (model.py:128, cov=['labels'] is the constructor's
    default): sparse x on the tile list, the covariate block as a dense [B, C] operand of the tail kernel, W1 = [H1, F + C].
    Output and every gradient (the covariate columns of dW1 included) against the fp64 restatement."""
    import scpair_amd as S
    from scpair_amd import ops
    S.set_precision(prec)
    try:
        rng = np.random.default_rng(B + F + C)
        m = _csr(rng, B + 3, F, 0.06)
        rows = rng.permutation(B + 3)[:B]
        x = torch.tensor(np.asarray(m[rows].todense(), dtype=np.float32))
        gc = torch.Generator().manual_seed(C)
        cov = torch.randn(B, C, generator=gc)
        cov[:, 0] = (torch.rand(B, generator=gc) < 0.5).float()          # a one-hot-like column
        rnd = (lambda t: t.bfloat16().double()) if prec == "bf16" else (lambda t: t.double())
        g, P, q, out_r = _draw(F, H1, H2, N3, x, rnd, H1 + N3 + C, cov=cov)
        wout = torch.randn(B, N3 if N3 else H2, generator=g)
        (out_r * wout.double()).sum().backward()
        d = {k: v.clone().to(DEV).requires_grad_(True) for k, v in P.items()}
        tiles = S.BatchTiles(S.DeviceCSR.from_scipy(m, DEV), rows)
        covd = cov.to(DEV)
        assert ops.enc_branch_supported(tiles, covd, d["W1"], d["W2"], d.get("W3"), True, True)
        spec = dict(d)
        spec.setdefault("W3", None); spec.setdefault("b3", None)
        out, = ops.enc_branches(tiles, [spec], 0.01, cov=covd)
        (out * wout.to(DEV)).sum().backward()
        rt = 2e-4 if prec == "fp32" else 2e-3
        _check("out", out, out_r, rt)
        for k in P:
            assert d[k].grad is not None, k
            if k == "W1":
                _check("W1[:, :F]", d[k].grad[:, :F], q[k].grad[:, :F], rt * (10 if prec == "bf16" else 1))
                _check("W1[:, F:]", d[k].grad[:, F:], q[k].grad[:, F:], rt)
            else:
                _check(k, d[k].grad, q[k].grad, rt)
        with torch.no_grad():
            out_e, = ops.enc_branches(tiles, [spec], 0.01, cov=covd)
        assert torch.equal(out_e, out.detach())
    finally:
        S.set_precision("fp32")


@pytest.mark.parametrize("C", [4, 8])
def test_covariate_columns_fused_adamw_matches_separate_step(C):
    """Packed pair of branches with a covariate block, precision 'bf16': two steps with the first layer's AdamW step
    fused into the weight-gradient kernel (sparse columns) and into the tail's reduction (covariate columns) against the
    same two steps through the plain gradient + multi-tensor AdamW: same parameters up to the hardware sqrt / rcp of the
    fused update."""
    import scpair_amd as S
    from scpair_amd import ops
    S.set_precision("bf16")
    try:
        B, F, H1, H2 = 256, 600, 132, 16
        rng = np.random.default_rng(9 + C)
        m = _csr(rng, B, F, 0.1)
        tiles = S.BatchTiles(S.DeviceCSR.from_scipy(m, DEV), np.arange(B))
        x = torch.tensor(np.asarray(m.todense(), dtype=np.float32))
        cov = torch.randn(B, C, generator=torch.Generator().manual_seed(3))
        covd = cov.to(DEV)
        res = []
        for fuse in (False, True):
            devs = []
            for bi, N3 in enumerate((H2, 1)):
                g, P, q, out_r = _draw(F, H1, H2, N3, x, lambda t: t.double(), 70 + bi, cov=cov)
                devs.append({k: torch.nn.Parameter(v.clone().to(DEV)) for k, v in P.items()})
            assert ops.pack_first_layers(devs[0]["W1"], devs[1]["W1"])
            opt = S.FusedAdamW([{"params": [p for d in devs for p in d.values()], "weight_decay": 1e-2, "lr": 1e-2}])
            if fuse:
                opt.fuse_first_layer(devs[0]["W1"], devs[1]["W1"])
            for _ in range(2):
                opt.zero_grad()
                outs = ops.enc_branches(tiles, devs, 0.01, cov=covd)
                sum((o * o).sum() for o in outs).backward()
                if fuse:
                    assert devs[0]["W1"].grad is None and devs[1]["W1"].grad is None
                opt.step()
            res.append([{k: v.detach().clone() for k, v in d.items()} for d in devs])
            # the bf16 shadow follows the master weights, covariate columns included
            for d in devs:
                sh = ops.shadow_of(d["W1"], create=False)
                assert sh is not None and torch.equal(sh[:, :F + C].float(), d["W1"].detach().bfloat16().float())
        for a, b in zip(*res):
            for k in a:
                assert torch.allclose(a[k], b[k], rtol=2e-5, atol=2e-6), (k, float((a[k] - b[k]).abs().max()))
    finally:
        S.set_precision("fp32")


@pytest.mark.parametrize("F,C", [(601, 0), (601, 3), (1001, 4), (640, 0)])
def test_homed_first_layers_any_feature_count(F, C):
    """ops.home_first_layers: the first-layer weights of an encoder's branches as row blocks of one tensor whose rows
    start on 128-byte boundaries.  Feature counts that are no multiple of 4 / 32 (and covariate blocks behind them) keep
    the fused weight-gradient + AdamW launch; two steps of it against the same steps through plain gradients + the
    multi-tensor AdamW kernel (which addresses the padded rows through ld_p); shapes, values, state_dict and deepcopy are
    those of ordinary parameters."""
    import copy
    import scpair_amd as S
    from scpair_amd import ops
    S.set_precision("bf16")
    try:
        B, H1, H2 = 192, 132, 16
        Ft = F + C
        rng = np.random.default_rng(F + C)
        m = _csr(rng, B, F, 0.1)
        tiles = S.BatchTiles(S.DeviceCSR.from_scipy(m, DEV), np.arange(B))
        x = torch.tensor(np.asarray(m.todense(), dtype=np.float32))
        cov = torch.randn(B, C, generator=torch.Generator().manual_seed(3)) if C else None
        covd = None if cov is None else cov.to(DEV)
        res = []
        for fuse in (False, True):
            devs = []
            for bi, N3 in enumerate((H2, 1)):
                g, P, q, out_r = _draw(F, H1, H2, N3, x, lambda t: t.double(), 70 + bi, cov=cov)
                devs.append({k: torch.nn.Parameter(v.clone().to(DEV)) for k, v in P.items()})
            before = [d["W1"].detach().clone() for d in devs]
            assert ops.home_first_layers(devs[0]["W1"], devs[1]["W1"])
            ld = (Ft + 31) // 32 * 32
            for d, w0 in zip(devs, before):
                assert tuple(d["W1"].shape) == (H1, Ft) and d["W1"].stride() == (ld, 1) and d["W1"].data_ptr() % 128 == 0
                assert torch.equal(d["W1"].detach(), w0)
            assert devs[1]["W1"].data_ptr() == devs[0]["W1"].data_ptr() + 4 * H1 * ld
            assert ops.home_first_layers(devs[0]["W1"], devs[1]["W1"])          # idempotent
            opt = S.FusedAdamW([{"params": [p for d in devs for p in d.values()], "weight_decay": 1e-2, "lr": 1e-2}])
            if fuse:
                opt.fuse_first_layer(devs[0]["W1"], devs[1]["W1"])
            for _ in range(2):
                opt.zero_grad()
                outs = ops.enc_branches(tiles, devs, 0.01, cov=covd)
                sum((o * o).sum() for o in outs).backward()
                if fuse:
                    assert devs[0]["W1"].grad is None and devs[1]["W1"].grad is None, "the fused path must have taken the step"
                opt.step()
            for d in devs:
                sh = ops.shadow_of(d["W1"], create=False)
                assert sh is not None and torch.equal(sh[:, :Ft].float(), d["W1"].detach().bfloat16().float())
                assert float(sh[:, Ft:].float().abs().max()) == 0.0 if sh.shape[1] > Ft else True
                st = opt.state[d["W1"]]
                assert st["exp_avg"].stride() == (ld, 1) and st["exp_avg_sq"].stride() == (ld, 1)
            # an ordinary parameter to its users
            sd = {k: v.detach().clone() for k, v in devs[0].items()}
            cp = copy.deepcopy(devs[0]["W1"])
            assert cp.is_contiguous() and torch.equal(cp.detach(), devs[0]["W1"].detach())
            with torch.no_grad():
                devs[0]["W1"].copy_(sd["W1"])
            res.append([{k: v.detach().clone() for k, v in d.items()} for d in devs])
        for a, b in zip(*res):
            for k in a:
                assert torch.allclose(a[k], b[k], rtol=2e-5, atol=2e-6), (k, float((a[k] - b[k]).abs().max()))
    finally:
        S.set_precision("fp32")


def _ref_bn(P, BN, x, rnd, training, cov=None, keep1=None, keep2=None):
    """fp64 forward of an FCNN built with batchnorm=True (utils.py:54-64: Linear -> LayerNorm -> LeakyReLU -> BatchNorm1d
    -> Dropout per layer) + the Linear behind it; BN: dict with w1, b1, rm1, rv1, w2, b2, rm2, rv2 (fp32 tensors), eps,
    momentum.  Returns the leaf dict, the output and the updated running statistics."""
    q = {k: v.double().requires_grad_(True) for k, v in P.items()}
    qb = {k: BN[k].double().requires_grad_(True) for k in ("w1", "b1", "w2", "b2")}
    F = x.shape[1]
    W1 = q["W1"][:, :F] + (rnd(P["W1"][:, :F]) - P["W1"][:, :F].double()).detach()
    y = rnd(x) @ W1.t() + q["b1"]
    if cov is not None:
        y = y + cov.double() @ q["W1"][:, F:].t()
    run = {}

    def bn(a, k):
        rm, rv = BN["rm" + k].double().clone(), BN["rv" + k].double().clone()
        o = Fn.batch_norm(a, rm, rv, qb["w" + k], qb["b" + k], training, BN["momentum"], BN["eps"])
        run["rm" + k], run["rv" + k] = rm, rv
        return o
    a1 = bn(Fn.leaky_relu(Fn.layer_norm(y, (y.shape[1],), q["g1"], q["be1"]), 0.01), "1")
    if keep1 is not None:
        a1 = a1 * keep1
    z2 = Fn.layer_norm(a1 @ q["W2"].t() + q["b2"], (q["W2"].shape[0],), q["g2"], q["be2"])
    a2 = bn(Fn.leaky_relu(z2, 0.01), "2")
    if keep2 is not None:
        a2 = a2 * keep2
    out = a2 @ q["W3"].t() + q["b3"] if "W3" in q else a2
    q.update({"bn" + k: v for k, v in qb.items()})
    return q, out, run


@pytest.mark.parametrize("B,F,H1,H2,N3,C", [(1024, 2000, 900, 40, 40, 0), (257, 700, 800, 30, 1, 4), (130, 333, 16, 8, 0, 0),
                                            (67, 515, 1024, 64, 64, 3), (9, 100, 4, 1, 1, 0), (3, 64, 260, 33, 7, 1)])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_encoder_branch_with_batchnorm(B, F, H1, H2, N3, C, prec):
    """batchnorm=True encoders (the tutorial's configuration) on the fused tail: three forward segments with the
    column-statistics kernel between them, three backward segments with the column-sum kernel between them.  Training
    mode (batch statistics, running statistics updated in place with the unbiased variance) and eval mode (running
    statistics), output and the gradient of every parameter incl. the BatchNorm affine ones, against fp64."""
    import scpair_amd as S
    from scpair_amd import ops
    S.set_precision(prec)
    try:
        rng = np.random.default_rng(B + F + 1)
        m = _csr(rng, B + 11, F, 0.06)
        rows = rng.permutation(B + 11)[:B]
        x = torch.tensor(np.asarray(m[rows].todense(), dtype=np.float32))
        rnd = (lambda t: t.bfloat16().double()) if prec == "bf16" else (lambda t: t.double())
        g = torch.Generator().manual_seed(H1 + N3 + 5)
        P = _params(g, F + C, H1, H2, N3)
        cov = torch.randn(B, C, generator=g) if C else None
        BN = dict(w1=1 + 0.2 * torch.randn(H1, generator=g), b1=0.1 * torch.randn(H1, generator=g),
                  rm1=0.1 * torch.randn(H1, generator=g), rv1=0.5 + torch.rand(H1, generator=g),
                  w2=1 + 0.2 * torch.randn(H2, generator=g), b2=0.1 * torch.randn(H2, generator=g),
                  rm2=0.1 * torch.randn(H2, generator=g), rv2=0.5 + torch.rand(H2, generator=g), eps=1e-5, momentum=0.1)
        wout = torch.randn(B, N3 if N3 else H2, generator=g)
        tiles = S.BatchTiles(S.DeviceCSR.from_scipy(m, DEV), rows)
        covd = None if cov is None else cov.to(DEV)
        for training in (True, False):
            q, out_r, run = _ref_bn(P, BN, x, rnd, training, cov)
            (out_r * wout.double()).sum().backward()
            d = {k: v.clone().to(DEV).requires_grad_(True) for k, v in P.items()}
            db = {k: BN[k].clone().to(DEV).requires_grad_(True) for k in ("w1", "b1", "w2", "b2")}
            rs = {k: BN[k].clone().to(DEV) for k in ("rm1", "rv1", "rm2", "rv2")}
            spec = dict(W1=d["W1"], b1=d["b1"], g1=d["g1"], be1=d["be1"], W2=d["W2"], b2=d["b2"], g2=d["g2"], be2=d["be2"],
                        W3=d.get("W3"), b3=d.get("b3"), bn1_w=db["w1"], bn1_b=db["b1"], bn2_w=db["w2"], bn2_b=db["b2"],
                        bn=dict(train=training, rm1=rs["rm1"], rv1=rs["rv1"], rm2=rs["rm2"], rv2=rs["rv2"], eps1=1e-5, eps2=1e-5,
                                mom1=0.1, mom2=0.1))
            out = ops.enc_branches(tiles, [spec], 0.01, cov=covd)[0]
            (out * wout.to(DEV)).sum().backward()
            rt = 2e-4 if prec == "fp32" else 2e-3
            _check(f"out train={training}", out, out_r, rt)
            for k in P:
                assert d[k].grad is not None, k
                _check(f"{k} train={training}", d[k].grad, q[k].grad, rt * (10 if (k == "W1" and prec == "bf16") else 1))
            for k in db:
                _check(f"bn {k} train={training}", db[k].grad, q["bn" + k].grad, rt)
            for k in rs:                                # running statistics: updated in training mode, untouched in eval mode
                _check(f"{k} train={training}", rs[k], run[k], 1e-5 if prec == "fp32" else 2e-3)
            if not training:
                with torch.no_grad():
                    out_e = ops.enc_branches(tiles, [spec], 0.01, cov=covd)[0]
                assert torch.equal(out_e, out.detach())
    finally:
        S.set_precision("fp32")


def test_two_batchnorm_branches_with_dropout():
    """Two BatchNorm branches of one encoder in one call (hidden_rep + extra Linear, hidden_lib + library head), dropout
    behind each BatchNorm: forward and every gradient against fp64 with the very masks the kernels draw."""
    import scpair_amd as S
    from scpair_amd import ops
    S.set_precision("fp32")
    B, F, H2 = 203, 400, 16
    rng = np.random.default_rng(3)
    m = _csr(rng, B, F, 0.1)
    x = torch.tensor(np.asarray(m.todense(), dtype=np.float32))
    tiles = S.BatchTiles(S.DeviceCSR.from_scipy(m, DEV), np.arange(B))
    g = torch.Generator().manual_seed(11)
    specs, refs, wouts, leaves = [], [], [], []
    for bi, (H1, N3, p1, p2) in enumerate([(132, 5, 0.3, 0.2), (64, 1, 0.25, 0.0)]):
        seeds = (100 + bi, 200 + bi)
        keep1 = _mask(ops, seeds[0], B, H1, p1)
        keep2 = _mask(ops, seeds[1], B, H2, p2) if p2 > 0 else None
        P = _params(g, F, H1, H2, N3)
        BN = dict(w1=1 + 0.2 * torch.randn(H1, generator=g), b1=0.1 * torch.randn(H1, generator=g), rm1=torch.zeros(H1),
                  rv1=torch.ones(H1), w2=1 + 0.2 * torch.randn(H2, generator=g), b2=0.1 * torch.randn(H2, generator=g),
                  rm2=torch.zeros(H2), rv2=torch.ones(H2), eps=1e-5, momentum=0.1)
        q, out_r, run = _ref_bn(P, BN, x, lambda t: t.double(), True, None, keep1, keep2)
        wout = torch.randn(B, N3, generator=g)
        (out_r * wout.double()).sum().backward()
        d = {k: v.clone().to(DEV).requires_grad_(True) for k, v in P.items()}
        db = {k: BN[k].clone().to(DEV).requires_grad_(True) for k in ("w1", "b1", "w2", "b2")}
        rs = {k: BN[k].clone().to(DEV) for k in ("rm1", "rv1", "rm2", "rv2")}
        specs.append(dict(W1=d["W1"], b1=d["b1"], g1=d["g1"], be1=d["be1"], W2=d["W2"], b2=d["b2"], g2=d["g2"], be2=d["be2"],
                          W3=d["W3"], b3=d["b3"], p1=p1, p2=p2, bn1_w=db["w1"], bn1_b=db["b1"], bn2_w=db["w2"], bn2_b=db["b2"],
                          bn=dict(train=True, rm1=rs["rm1"], rv1=rs["rv1"], rm2=rs["rm2"], rv2=rs["rv2"], eps1=1e-5, eps2=1e-5,
                                  mom1=0.1, mom2=0.1)))
        refs.append((q, out_r, run, P, d, db, rs))
        wouts.append(wout)
        leaves.append(seeds)
    outs = ops.enc_branches(tiles, specs, 0.01, seeds=leaves)
    sum((o * w.to(DEV)).sum() for o, w in zip(outs, wouts)).backward()
    for bi, (q, out_r, run, P, d, db, rs) in enumerate(refs):
        _check(f"out{bi}", outs[bi], out_r, 2e-4)
        for k in P:
            _check(f"{k}{bi}", d[k].grad, q[k].grad, 2e-4)
        for k in db:
            _check(f"bn {k}{bi}", db[k].grad, q["bn" + k].grad, 2e-4)
        for k in rs:
            _check(f"{k}{bi}", rs[k], run[k], 1e-5)
