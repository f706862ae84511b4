"""Full-size shapes of BASELINE.json on the GPU (C2: 900 x 100 000 first layer, batch 1024; C5: 250 000 peaks, batch
4096; C4: unimodal VAE with 20 000 genes and a 900-wide hidden layer), checked against the defining formulas evaluated
with plain dense torch math ON THE SAME DEVICE (fp32 matmuls of the bf16-rounded operands, fp64 where it is cheap) --
the CPU oracle would take minutes at these sizes -- and against size-independent properties (linearity in dY, mean
over all label elements, agreement of the eval and train passes).

Tolerances: products of bf16-rounded operands are exact in fp32, so the first-layer kernels differ from the dense
formula only by fp32 summation order over K = batch (1024 / 4096) or K = features terms: 2e-5 of the output scale is
stated (measured <= 4e-6).  AdamW moments after two steps: 5e-5 of their scale (measured 1.3e-5: the gradient's
summation-order noise, same update formula)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _scale_err(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)


def test_k2prime_c2_shape_against_the_formula():
    """scp_spmm_dw_adamw_bf16 at 900 x 100 000, batch 1024 (the launch the bench's roofline is quoted on): two steps
    against  g = bf16(dY)^T bf16(X)  (dense fp32 matmul)  followed by the AdamW formulas (torch.optim.AdamW semantics:
    decoupled weight decay, bias-corrected moments), not against another kernel of this library."""
    import scpair_amd as S
    from scpair_amd import ops, synth
    S.set_precision("bf16")
    try:
        B, F, H = 1024, 100_000, 900
        lr, wd, b1, b2, eps = 1e-2, 1e-2, 0.9, 0.999, 1e-8
        atac = synth.make_atac_device(4096, F, DEV, seed=5, p=0.03)
        rows = np.random.default_rng(0).permutation(4096)[:B]
        t = S.BatchTiles(atac, rows, want_lib=False)
        X = t.to_dense()                                            # [B, F] of 0 / 1: exact in bf16
        g_ = torch.Generator(device=DEV).manual_seed(1)
        W = torch.nn.Parameter(torch.randn(H, F, device=DEV, generator=g_) * 0.02)
        Wr = W.detach().clone()
        mr, vr = torch.zeros_like(Wr), torch.zeros_like(Wr)
        opt = ops.FusedAdamW([{"params": [W], "weight_decay": wd, "lr": lr}], betas=(b1, b2), eps=eps)
        opt.fuse_first_layer(W)
        for step in (1, 2):
            dY = torch.randn(B, H, device=DEV, generator=g_) * (0.5 if step == 1 else 2.0)
            out = ops.fcnn_layer(t, None, W, None, None, None, 1.0)
            # forward at this size too: Y = bf16(X) bf16(W)^T, summed over 100 000 features in fp32
            y_ref = X @ W.detach().bfloat16().float().t()    # the kernel's own weights: this line checks the product
            assert _scale_err(out, y_ref) < 2e-5, ("fwd", step, _scale_err(out, y_ref))
            opt.zero_grad()
            out.backward(dY)
            opt.step()
            g = dY.bfloat16().float().t() @ X                     # [H, F]
            Wr.mul_(1 - lr * wd)
            mr.mul_(b1).add_(g, alpha=1 - b1)
            vr.mul_(b2).addcmul_(g, g, value=1 - b2)
            Wr.addcdiv_(mr / (1 - b1 ** step), (vr / (1 - b2 ** step)).sqrt_().add_(eps), value=-lr)
            del g, y_ref, out
        st = opt.state[W]
        assert st["step"] == 2
        assert _scale_err(st["exp_avg"], mr) < 5e-5, _scale_err(st["exp_avg"], mr)            # measured 1.3e-5
        assert _scale_err(st["exp_avg_sq"], vr) < 5e-5, _scale_err(st["exp_avg_sq"], vr)
        # the update is lr * m_hat / (sqrt(v_hat) + eps): where v_hat is tiny the quotient amplifies summation-order
        # noise of g, so the weights are compared in units of the step size lr
        assert float((W.detach() - Wr).abs().max()) < 2e-3 * lr, float((W.detach() - Wr).abs().max()) / lr
        assert torch.equal(ops.shadow_of(W, create=False)[:, :F], W.detach().bfloat16())
    finally:
        S.set_precision("fp32")


def test_first_layer_c5_shape():
    """C5: 250 000 peaks, batch 4096, hidden 900 (precision 'bf16').  Forward against the dense product, weight
    gradient against the dense product and linear in dY (size-independent property)."""
    import scpair_amd as S
    from scpair_amd import ops, synth
    S.set_precision("bf16")
    try:
        B, F, H = 4096, 250_000, 900
        atac = synth.make_atac_device(8192, F, DEV, seed=9, p=0.02)
        rows = np.random.default_rng(3).permutation(8192)[:B]
        t = S.BatchTiles(atac, rows, want_lib=False)
        g_ = torch.Generator(device=DEV).manual_seed(2)
        W = (torch.randn(H, F, device=DEV, generator=g_) * 0.02).requires_grad_(True)
        X = t.to_dense()
        out = ops.fcnn_layer(t, None, W, None, None, None, 1.0)
        y_ref = X @ W.detach().bfloat16().float().t()
        assert _scale_err(out, y_ref) < 2e-5, _scale_err(out, y_ref)
        dY1 = torch.randn(B, H, device=DEV, generator=g_)
        dY2 = torch.randn(B, H, device=DEV, generator=g_)
        (gW1,) = torch.autograd.grad(out, W, dY1, retain_graph=True)
        ref1 = dY1.bfloat16().float().t() @ X
        assert _scale_err(gW1, ref1) < 2e-5, _scale_err(gW1, ref1)
        del ref1, y_ref
        # linearity: dW(2 dY1 + dY2) = 2 dW(dY1) + dW(dY2) up to the bf16 rounding of the combined operand
        (gW2,) = torch.autograd.grad(out, W, dY2, retain_graph=True)
        comb = (2 * dY1.bfloat16().float() + dY2.bfloat16().float())
        (gW3,) = torch.autograd.grad(out, W, comb)
        lin = 2 * gW1 + gW2
        # comb is re-rounded to bf16 inside the kernel: relative 2^-9 per element, random sign, summed over B terms
        assert _scale_err(gW3, lin) < 2e-3, _scale_err(gW3, lin)
    finally:
        S.set_precision("fp32")


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_ber_decoder_c5_shape(prec):
    """Fused Bernoulli decoder at 250 000 peaks x 4096 cells, K = 40, with the inferred-library factor: loss and all
    gradients against dense torch math on the device (logits [4096 x 250 000] in fp32, reductions in fp64)."""
    import scpair_amd as S
    from scpair_amd import ops, synth
    S.set_precision(prec)
    try:
        B, P, K = 4096, 250_000, 40
        atac = synth.make_atac_device(8192, P, DEV, seed=4, p=0.02)
        rows = np.random.default_rng(5).permutation(8192)[:B]
        t = S.BatchTiles(atac, rows, want_lib=False)
        g_ = torch.Generator(device=DEV).manual_seed(3)
        h = torch.randn(B, K, device=DEV, generator=g_).requires_grad_(True)
        W = (torch.randn(P, K, device=DEV, generator=g_) * 0.1).requires_grad_(True)
        b = (torch.randn(P, device=DEV, generator=g_) * 0.1 - 3.0).requires_grad_(True)
        lat = torch.randn(B, 1, device=DEV, generator=g_).requires_grad_(True)
        loss = ops.ber_decoder_loss(h, W, b, None, lat, t)
        gh, gW, gb, gl = torch.autograd.grad(loss, (h, W, b, lat))
        # dense restatement (model.py:164-190, loss.py:36-41); bf16 mode rounds the MFMA operands h and W once
        rnd = (lambda x: x.bfloat16().float()) if prec == "bf16" else (lambda x: x)
        hr, Wr = h.detach().clone().requires_grad_(True), W.detach().clone().requires_grad_(True)
        br, lr_ = b.detach().clone().requires_grad_(True), lat.detach().clone().requires_grad_(True)
        hq = hr + (rnd(hr.detach()) - hr.detach())
        Wq = Wr + (rnd(Wr.detach()) - Wr.detach())
        q = torch.sigmoid(hq @ Wq.t() + br) * torch.sigmoid(lr_)
        Y = t.to_dense()
        ll = torch.where(Y > 0, torch.log(q).clamp_min(-100.0), torch.log1p(-q).clamp_min(-100.0))
        ref = -(ll.double().sum() / (B * P)).float()
        rh, rW, rb, rl = torch.autograd.grad(ref, (hr, Wr, br, lr_))
        lt, gt_ = (2e-5, 5e-4) if prec == "fp32" else (2e-3, 2e-2)       # hardware exp / log / rcp in bf16 mode
        assert abs(float(loss) - float(ref)) < lt * abs(float(ref)), (float(loss), float(ref))
        for nm, a, r in (("dh", gh, rh), ("dW", gW, rW), ("db", gb, rb), ("dlat", gl, rl)):
            assert _scale_err(a, r) < gt_, (nm, _scale_err(a, r))
        # eval pass (no gradients) returns the same loss
        with torch.no_grad():
            le = ops.ber_decoder_loss(h, W, b, None, lat, t)
        assert abs(float(le) - float(loss)) <= 1e-6 * abs(float(loss))
    finally:
        S.set_precision("fp32")


@pytest.mark.parametrize("lik", ["nb", "zinb"])
def test_vae_step_c4_shape(lik):
    """One unimodal VAE training step at the C4 shape (20 000 genes, hidden [900], latent 40, batch 1024; precision
    'fp32') against the CPU oracle on the same weights, noise and batch: total / reconstruction / KL losses and the
    gradients of the wide first encoder layer and of the decoder head."""
    import scpair_amd as S
    from scpair_amd import synth
    from oracle import ref_cpu as O
    S.set_precision("fp32")
    B, G, Hd, LAT = 1024, 20_000, 900, 40
    torch.manual_seed(0)
    enc = S.VAE_ENCODER(G, 0, [Hd], LAT, layernorm=True, dropout_rate=0, infer_library_size=False)
    dec = S.VAE_DECODER(G, 0, [Hd], LAT, layernorm=True, dropout_rate=0, infer_library_size=False, sample_factor=True,
                        feature_factor=False, distribution=lik, dispersion="feature-cell")
    sd_e = {k: v.detach().clone().requires_grad_(True) for k, v in enc.state_dict().items()}
    sd_d = {k: v.detach().clone().requires_grad_(True) for k, v in dec.state_dict().items()}
    rna = synth.make_rna(2048, G, seed=3)
    rows = np.random.default_rng(1).permutation(2048)[:B]
    eps = torch.randn(B, LAT)
    enc.eps_source = lambda shape, device: eps.to(device).reshape(shape)
    enc.to(DEV).train(); dec.to(DEV).train()
    t = S.BatchTiles(S.DeviceCSR.from_scipy(rna, DEV), rows)
    from scpair_amd.utils import _vae_losses
    tot, rec, kl = _vae_losses(t, t.lib.reshape(-1, 1), enc, dec, 1.0, lik, False, None)
    tot.backward()
    x = torch.tensor(np.asarray(rna[rows].todense(), dtype=np.float32))
    mu_, sigma_, z_, _ = O.vae_encoder(sd_e, x, eps)
    _, theta_, rate_, pi_ = O.vae_decoder(sd_d, z_, None, x.sum(1, keepdim=True), None, distribution=lik, sample_factor=True)
    rrec = O.raw_loss(x, rate_, theta_, pi_ if lik == "zinb" else None, lik)
    rkl = O.kl_normal(mu_, sigma_)
    rtot = rkl + rrec
    rtot.backward()
    for nm, a, r in (("total", tot, rtot), ("recon", rec, rrec), ("kl", kl, rkl)):
        assert abs(float(a) - float(r)) < 1e-4 * abs(float(r)), (nm, float(a), float(r))
    first = [k for k in sd_e if k.endswith("weight") and sd_e[k].ndim == 2 and sd_e[k].shape[1] == G][0]
    ge = dict(enc.named_parameters())[first].grad
    assert _scale_err(ge, sd_e[first].grad) < 1e-3, _scale_err(ge, sd_e[first].grad)
    for k, p in dec.named_parameters():
        r = sd_d[k].grad
        if r is None or p.grad is None:
            continue
        assert _scale_err(p.grad, r) < 1e-3, (k, _scale_err(p.grad, r))


def test_wide_linear_c3_shape_against_dense_matmuls():
    """The ZINB zero-inflation layer at its C3 size (20 000 -> 20 000, batch 1024; model.py:228,255-256) on the
    hand-written wide GEMM: forward with bias, input gradient with the relu mask, weight gradient, each against torch's
    fp32 matmul of the same bf16-rounded operands on the device.  Both sides add 20 000 (1 024 for the weight
    gradient) exact products in fp32 in different orders: 2e-5 of the output scale is stated.  Then the same layer with
    the AdamW step in the weight-gradient epilogue (scp_gemm_bf16_adamw) against the formula applied to that gradient."""
    import scpair_amd as S
    from scpair_amd import ops
    S.set_precision("bf16")
    try:
        G, B = 20000, 1024
        g_ = torch.Generator(device=DEV).manual_seed(3)
        x = torch.randn(B, G, device=DEV, generator=g_)
        W = torch.nn.Parameter(torch.randn(G, G, device=DEV, generator=g_) * 0.01)
        b = torch.nn.Parameter(torch.randn(G, device=DEV, generator=g_))
        dy = torch.randn(B, G, device=DEV, generator=g_)
        xr = x.clone().requires_grad_(True)
        y = ops.linear(xr, W, b, relu_input=True)
        y.backward(dy)
        xb = torch.relu(x).bfloat16().float()
        Wb = W.detach().bfloat16().float()
        dyb = dy.bfloat16().float()
        assert _scale_err(y, xb @ Wb.t() + b.detach()) < 2e-5
        assert _scale_err(xr.grad, (dyb @ Wb) * (xb > 0)) < 2e-5
        dW_ref = dyb.t() @ xb
        assert _scale_err(W.grad, dW_ref) < 2e-5
        del y, xr, Wb
        # AdamW in the epilogue: one step from zero moments
        lr, wd, b1, b2, eps = 1e-3, 1e-2, 0.9, 0.999, 1e-8
        W0 = W.detach().clone()
        W.grad = None
        opt = ops.FusedAdamW([{"params": [W], "weight_decay": wd, "lr": lr}], betas=(b1, b2), eps=eps)
        opt.fuse_weight_update(W)
        xr = x.clone().requires_grad_(True)
        ops.linear(xr, W, None, relu_input=True).backward(dy)
        assert W.grad is None
        opt.step()
        m = (1 - b1) * dW_ref
        v = (1 - b2) * dW_ref * dW_ref
        Wn = W0 * (1 - lr * wd) - (lr / (1 - b1)) * m / (v.sqrt() / (1 - b2) ** 0.5 + eps)
        st = opt.state[W]
        assert _scale_err(st["exp_avg"], m) < 5e-5 and _scale_err(st["exp_avg_sq"], v) < 5e-5
        # the first step from zero moments is lr * g / (|g| + eps'): where g is within rounding of zero its sign, and
        # with it the whole update, is decided by the summation order -- compare the applied change elsewhere
        ok = dW_ref.abs() > 1e-3 * float(dW_ref.abs().max())
        assert float(ok.float().mean()) > 0.99
        err = ((W.detach() - W0) - (Wn - W0)).abs()[ok].max()
        assert float(err) < 2e-3 * lr, float(err)
        assert torch.equal(ops.shadow_of(W, create=False)[:, :G], W.detach().bfloat16())
    finally:
        S.set_precision("fp32")
