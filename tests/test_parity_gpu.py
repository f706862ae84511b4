"""GPU parity of the stand-alone likelihood entry points and of every module variant against the fixtures generated
from the reference itself (tools/make_goldens.py): kat_likelihood / kat_bce_mse / mod_input / mod_gau_map / mod_vae.

Everything goes through libscpair_hip.so (C ABI).  Tolerances are written per test; fp32 mode unless stated.
"""
import json

import numpy as np
import pytest
import torch
from scipy import special as sps
from torch import nn

from tests import util as U

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
EPS32 = 2.0 ** -23


@pytest.fixture(scope="module")
def S():
    import scpair_amd as pkg
    import scpair_amd.ops as ops
    import scpair_amd.sparse as sparse
    ops.set_precision("fp32")
    return type("S", (), dict(pkg=pkg, ops=ops, sparse=sparse))


def close(a, b, rtol, atol, what=""):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).double().cpu().reshape(a.shape)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), f"{what}: max err {float(err.max())} (scale {float(b.abs().max())}), " \
                                     f"worst err/tol {float((err / tol).max())}"


def _cond_close(got, exact, ref32, cond, what, k=8.0, rtol=2e-6):
    """fp32 kernel output against the exact (fp64 reference) value.  The formulas cancel catastrophically in places
    (theta = 1e6: lgamma(x+theta) - lgamma(theta) is a difference of numbers near 1.3e7), so the bound is stated in
    units of the fp32 rounding of the largest intermediate term, `cond`: |got - exact| <= k * 2^-23 * cond +
    rtol * |exact|.  The reference's own fp32 result obeys the same bound (checked here too), which is what makes it
    the right yardstick.  inf / NaN must agree with the reference's fp32 result exactly in kind."""
    got = torch.as_tensor(got).detach().double().cpu().numpy().reshape(-1)
    exact = np.asarray(exact, dtype=np.float64).reshape(-1)
    ref32 = np.asarray(ref32, dtype=np.float64).reshape(-1)
    fin = np.isfinite(ref32) & np.isfinite(exact)
    assert np.array_equal(np.isnan(got[~fin]), np.isnan(ref32[~fin])), f"{what}: NaN pattern differs from the reference"
    inf_ref = np.isinf(ref32) & ~fin
    assert np.array_equal(got[inf_ref], ref32[inf_ref]), f"{what}: inf pattern differs from the reference"
    tol = k * EPS32 * cond[fin] + rtol * np.abs(exact[fin]) + 1e-30
    e_ref = np.abs(ref32[fin] - exact[fin]) / tol
    e_got = np.abs(got[fin] - exact[fin]) / tol
    assert np.isfinite(got[fin]).all(), f"{what}: kernel non-finite where the reference is finite"
    assert e_ref.max() <= 1.0, f"{what}: yardstick too tight for the reference itself ({e_ref.max():.2f})"
    assert e_got.max() <= 1.0, f"{what}: worst err/tol {e_got.max():.2f} (reference fp32: {e_ref.max():.2f})"
    return float(e_got.max()), float(e_ref.max())


def _nb_conds(x, mu, th):
    """Magnitudes of the intermediate terms of log_nb_positive and of its gradients (scVI_distribution.py:97-105)."""
    e = 1e-8
    s = th + mu + e
    lt, ls, lm = np.abs(np.log(th + e)), np.abs(np.log(s)), np.abs(np.log(mu + e))
    lg = np.abs(sps.gammaln(x + th)) + np.abs(sps.gammaln(th)) + np.abs(sps.gammaln(x + 1))
    c_val = th * (lt + ls) + x * (lm + ls) + lg + 1.0
    c_dmu = th / s + x / (mu + e) + x / s
    c_dth = lt + ls + th / (th + e) + th / s + x / s + np.abs(sps.digamma(x + th)) + np.abs(sps.digamma(th))
    return c_val, c_dmu, c_dth


def test_log_nb_positive_grid(S):
    """scp_nbll_fwd / scp_nbll_bwd (pi = NULL) over the reference's KAT grid x in {0..1000}, mu in {1e-8..1e4},
    theta in {1e-4..1e6} (scVI_distribution.py:74-105), values and autograd gradients."""
    from scpair_amd.scVI_distribution import log_nb_positive
    d = U.load("kat_likelihood")
    x, mu, th = [torch.tensor(d[k], dtype=torch.float32).to(DEV) for k in ("x", "mu", "theta")]
    mu.requires_grad_(True); th.requires_grad_(True)
    out = log_nb_positive(x, mu, th)
    gm, gt = torch.autograd.grad(out.sum(), [mu, th])
    c_val, c_dmu, c_dth = _nb_conds(d["x"], d["mu"], d["theta"])
    _cond_close(out.cpu(), d["nb_f64"], d["nb_f32"], c_val, "log_nb value")
    _cond_close(gm.cpu(), d["nb_dmu_f64"], d["nb_dmu_f32"], c_dmu, "log_nb dmu")
    _cond_close(gt.cpu(), d["nb_dtheta_f64"], d["nb_dtheta_f32"], c_dth, "log_nb dtheta")
    # SURVEY KATs
    kx, km, kt = [torch.tensor(v, device=DEV) for v in ([0., 1., 5.], [.5, 2., 3.], [1., 2., .5])]
    close(log_nb_positive(kx, km, kt), d["kat_nb"], rtol=2e-6, atol=1e-7, what="kat nb")
    # 1-D theta broadcast over the columns of a 2-D x (the dispersion='feature' call form, model.py:249-250)
    x2, mu2 = x[:2680].reshape(-1, 8), mu.detach()[:2680].reshape(-1, 8)
    thv = th.detach()[:8].clone()
    full = log_nb_positive(x2, mu2, thv.expand_as(x2).contiguous())
    vec = log_nb_positive(x2, mu2, thv)
    assert torch.equal(full, vec)


def test_log_zinb_positive_grid(S):
    """scp_nbll_fwd / scp_nbll_bwd with pi over the same grid x pi in {-20..20} (scVI_distribution.py:29-71): both
    case terms are evaluated and 0/1-weighted like the reference, so its inf / NaN pattern is reproduced."""
    from scpair_amd.scVI_distribution import log_zinb_positive
    d = U.load("kat_likelihood")
    x, mu, th, pi = [torch.tensor(d[k], dtype=torch.float32).to(DEV) for k in ("x", "mu", "theta", "pi")]
    mu.requires_grad_(True); th.requires_grad_(True); pi.requires_grad_(True)
    out = log_zinb_positive(x, mu, th, pi)
    gm, gt, gp = torch.autograd.grad(out.sum(), [mu, th, pi])
    c_val, c_dmu, c_dth = _nb_conds(d["x"], d["mu"], d["theta"])
    c_val = c_val + 2 * np.abs(d["pi"]) + 2.0
    # zero counts: the gradients carry sigmoid(t), t = -pi + theta (log(theta+eps) - log(theta+mu+eps)); the rounding of
    # t (theta * ulp(log)) moves sigmoid(t) by up to a quarter of it
    e = 1e-8
    s = d["theta"] + d["mu"] + e
    t_err = (d["x"] == 0) * d["theta"] * (np.abs(np.log(d["theta"] + e)) + np.abs(np.log(s))) * 0.25
    c_dmu = c_dmu + t_err * d["theta"] / s
    c_dth = c_dth + t_err * (np.abs(np.log(d["theta"] + e)) + np.abs(np.log(s)) + 2.0)
    _cond_close(out.cpu(), d["zinb_f64"], d["zinb_f32"], c_val, "log_zinb value")
    _cond_close(gm.cpu(), d["zinb_dmu_f64"], d["zinb_dmu_f32"], c_dmu, "log_zinb dmu")
    _cond_close(gt.cpu(), d["zinb_dtheta_f64"], d["zinb_dtheta_f32"], c_dth, "log_zinb dtheta")
    _cond_close(gp.cpu(), d["zinb_dpi_f64"], d["zinb_dpi_f32"], 2.0 + t_err, "log_zinb dpi")
    kx, km, kt, kp = [torch.tensor(v, device=DEV) for v in ([0., 1., 5.], [.5, 2., 3.], [1., 2., .5], [0., -1., 2.])]
    close(log_zinb_positive(kx, km, kt, kp), d["kat_zinb"], rtol=2e-6, atol=1e-7, what="kat zinb")


def test_nb_zinb_distribution_classes(S):
    """NegativeBinomial / ZeroInflatedNegativeBinomial .log_prob as raw_loss calls them (loss.py:46-51)."""
    from scpair_amd.loss import raw_loss
    d = U.load("mod_raw")
    n = 0
    for tag in json.loads(str(d["cases"])):
        c = U.cfg(d, f"{tag}/cfg")
        y = torch.tensor(d[f"{tag}/y"]).to(DEV)
        rate, theta = torch.tensor(d[f"{tag}/rate"]).to(DEV), torch.tensor(d[f"{tag}/theta"]).to(DEV)
        pi = U.opt(d, f"{tag}/pi")
        rate.requires_grad_(True)
        loss = raw_loss(x=y, px_mu_rate=rate, px_theta=theta, px_mu_scale=None,
                        px_pi=None if pi is None else pi.to(DEV), likelihood=c["lik"])
        close(loss, d[f"{tag}/loss"], rtol=2e-5, atol=0, what=f"{tag} raw_loss")
        loss.backward()
        assert rate.grad is not None and bool(torch.isfinite(rate.grad).all())
        n += 1
    assert n >= 12


def test_binarized_loss_standalone(S):
    """scp_bce_masked_fwd / _bwd against the reference's binarized_loss (loss.py:36-41): p = 0, 1, 1e-9, 1 - 1e-7 and
    the denormal 1e-45 hit BCELoss's -100 log clamp and the 1e-12 floor of its backward; NaN labels are masked."""
    from scpair_amd.loss import binarized_loss
    b = U.load("kat_bce_mse")
    x = torch.tensor(b["bce_x"]).to(DEV)
    p = torch.tensor(b["bce_p"]).to(DEV).requires_grad_(True)
    loss = binarized_loss(x, p)
    loss.backward()
    close(loss, b["bce_loss"], rtol=2e-6, atol=0, what="bce loss")
    close(p.grad, b["bce_dp"], rtol=2e-6, atol=1e-9, what="bce dp")
    assert bool((p.grad[torch.isnan(x)] == 0).all())
    kx, kp = torch.tensor([0., 1., 1.], device=DEV), torch.tensor([0.2, 0.9, 1e-9], device=DEV)
    close(binarized_loss(kx, kp), b["kat_bce"], rtol=2e-6, atol=0, what="kat bce")
    close(binarized_loss(kx, kp), 7.01725721, rtol=2e-6, atol=0, what="kat bce (SURVEY value)")
    # upstream gradient scaling
    p2 = torch.tensor(b["bce_p"]).to(DEV).requires_grad_(True)
    (binarized_loss(x, p2) * 3.0).backward()
    close(p2.grad, 3.0 * b["bce_dp"], rtol=3e-6, atol=1e-9, what="bce dp scaled")


# ---------------------------------------------------------------------------------------------------------
def _load(module, sd):
    module.load_state_dict({k: v for k, v in sd.items()})
    return module.to(DEV)


def _check_grads(module, grads, rtol, atol, what):
    n = 0
    for k, p in module.named_parameters():
        if k not in grads:
            continue
        g = grads[k]
        assert p.grad is not None, f"{what}: {k} received no gradient"
        close(p.grad, g, rtol=rtol, atol=atol + rtol * float(g.abs().max()), what=f"{what} grad {k}")
        n += 1
    return n


@pytest.mark.parametrize("sparse_in", [True, False])
def test_input_module_golden(S, sparse_in):
    """Input_Module over +-library branch, +-covariates, +-BatchNorm, +-extra linear (model.py:60-143): outputs,
    every parameter gradient, BatchNorm running statistics after the training-mode forward, eval-mode outputs;
    input as a sparse tile list and as a dense tensor."""
    d = U.load("mod_input")
    for tag in json.loads(str(d["cases"])):
        c = U.cfg(d, f"{tag}/cfg")
        sd = U.sub(d, f"{tag}/sd")
        enc = S.pkg.Input_Module(input_dim=32, input_batch_num=3 if c["cov"] else 0, hidden_layer=[16, 8],
                                 activation=nn.LeakyReLU(), layernorm=True, batchnorm=c["bn"], dropout_rate=0.0,
                                 add_linear_layer=c["extra"], infer_library_size=c["lib"])
        enc = _load(enc, sd).train()
        x = torch.tensor(d[f"{tag}/x"])
        cov = U.opt(d, f"{tag}/cov")
        covd = None if cov is None else cov.to(DEV)
        xin = S.sparse.BatchTiles(S.sparse.DeviceCSR.from_dense(x, DEV), np.arange(x.shape[0])) if sparse_in else x.to(DEV)
        rep, lat = enc(xin, covd)
        close(rep, d[f"{tag}/rep"], rtol=1e-4, atol=1e-5, what=f"{tag} rep")
        loss = (rep * torch.tensor(d[f"{tag}/w_rep"]).to(DEV)).sum()
        if c["lib"]:
            close(lat, d[f"{tag}/lat"], rtol=1e-4, atol=1e-5, what=f"{tag} lat")
            loss = loss + (lat * torch.tensor(d[f"{tag}/w_lat"]).to(DEV)).sum()
        else:
            assert lat is None
        loss.backward()
        n = _check_grads(enc, U.sub(d, f"{tag}/grad"), rtol=1e-3, atol=1e-5, what=tag)
        assert n == len(list(enc.parameters()))
        if c["bn"]:
            after = enc.state_dict()
            for k, v in U.sub(d, f"{tag}/sd_after").items():
                close(after[k], v, rtol=1e-5, atol=1e-6, what=f"{tag} {k}")
        enc.eval()
        with torch.no_grad():
            rep_e, lat_e = enc(xin, covd)
        close(rep_e, d[f"{tag}/rep_eval"], rtol=1e-4, atol=1e-5, what=f"{tag} rep_eval")
        if c["lib"]:
            close(lat_e, d[f"{tag}/lat_eval"], rtol=1e-4, atol=1e-5, what=f"{tag} lat_eval")


def test_output_gau_and_module_module_golden(S):
    """Output_Module_Gau (+-feature factor, +-covariates; model.py:268-300) with scaled_loss, and Module_Module
    (plain, non_neg, hidden + LayerNorm, two hidden + BatchNorm; model.py:305-336)."""
    from scpair_amd.loss import scaled_loss
    d = U.load("mod_gau_map")
    for i in range(3):
        tag = f"gau{i}"
        c = U.cfg(d, f"{tag}/cfg")
        dec = S.pkg.Output_Module_Gau(output_dim=32, output_batch_num=3 if c["cov"] else 0, hidden_layer=[16, 8],
                                      infer_library_size=False, feature_factor=c["ff"])
        dec = _load(dec, U.sub(d, f"{tag}/sd")).train()
        h = torch.tensor(d[f"{tag}/h"]).to(DEV).requires_grad_(True)
        cov = U.opt(d, f"{tag}/cov")
        out = dec(h, None if cov is None else cov.to(DEV), None, None)
        close(out, d[f"{tag}/out"], rtol=1e-5, atol=1e-6, what=f"{tag} out")
        loss = scaled_loss(torch.tensor(d[f"{tag}/y"]).to(DEV), out)
        close(loss, d[f"{tag}/loss"], rtol=1e-5, atol=0, what=f"{tag} loss")
        loss.backward()
        close(h.grad, d[f"{tag}/dh"], rtol=1e-4, atol=1e-7, what=f"{tag} dh")
        assert _check_grads(dec, U.sub(d, f"{tag}/grad"), rtol=2e-4, atol=1e-7, what=tag) >= 2
    for i in range(4):
        tag = f"map{i}"
        c = U.cfg(d, f"{tag}/cfg")
        net = S.pkg.Module_Module(origin_module_dim=8, target_module_dim=8, translational_hidden_nodes=c["hid"],
                                  non_neg=c["nonneg"], activation=nn.LeakyReLU(), layernorm=c["ln"], batchnorm=c["bn"],
                                  dropout_rate=0)
        net = _load(net, U.sub(d, f"{tag}/sd")).train()
        x = torch.tensor(d[f"{tag}/x"]).to(DEV).requires_grad_(True)
        pred = net(x)
        close(pred, d[f"{tag}/pred"], rtol=1e-4, atol=1e-6, what=f"{tag} pred")
        loss = scaled_loss(torch.tensor(d[f"{tag}/y"]).to(DEV), pred)
        close(loss, d[f"{tag}/loss"], rtol=1e-5, atol=0, what=f"{tag} loss")
        loss.backward()
        close(x.grad, d[f"{tag}/dx"], rtol=1e-3, atol=1e-6, what=f"{tag} dx")
        assert _check_grads(net, U.sub(d, f"{tag}/grad"), rtol=1e-3, atol=1e-6, what=tag) >= 2


@pytest.mark.parametrize("sparse_in", [True, False])
def test_vae_modules_golden(S, sparse_in):
    """VAE_ENCODER (+-enc_lib softplus branch) and VAE_DECODER (nb / zinb x feature-cell / feature dispersion x observed /
    inferred library size, and the Gaussian head; pairedVAE.py:24-187) through train_VAE's loss (utils.py:179-199):
    mu, sigma, z with the recorded noise, KL, reconstruction loss, total, and every parameter gradient."""
    from scpair_amd.utils import _vae_losses
    d = U.load("mod_vae")
    for i in range(int(d["ncases"])):
        tag = f"c{i}"
        c = U.cfg(d, f"{tag}/cfg")
        kw = dict(input_dim=32, input_batch_num=0, hidden_dim=[16], latent_dim=6, layernorm=True,
                  activation=nn.LeakyReLU(), batchnorm=False, dropout_rate=0, infer_library_size=c["infer"])
        enc = _load(S.pkg.VAE_ENCODER(**kw), U.sub(d, f"{tag}/enc_sd")).train()
        dec = _load(S.pkg.VAE_DECODER(**kw, sample_factor=not c["infer"], feature_factor=False, distribution=c["dist"],
                                      dispersion=c["disp"]), U.sub(d, f"{tag}/dec_sd")).train()
        eps = torch.tensor(d[f"{tag}/eps"]).to(DEV)
        enc.eps_source = lambda shape, device: eps
        x = torch.tensor(d[f"{tag}/x"])
        if sparse_in and c["dist"] != "gau":
            xin = S.sparse.BatchTiles(S.sparse.DeviceCSR.from_dense(x, DEV), np.arange(x.shape[0]))
        else:
            xin = x.to(DEV)
        lib = U.opt(d, f"{tag}/lib")
        total, recon, kl = _vae_losses(xin, None if lib is None else lib.to(DEV), enc, dec, c["kl_weight"], c["dist"],
                                       False, None)
        close(kl, d[f"{tag}/kl"], rtol=2e-5, atol=0, what=f"{tag} kl")
        close(recon, d[f"{tag}/recon"], rtol=2e-5, atol=0, what=f"{tag} recon")
        close(total, d[f"{tag}/total"], rtol=2e-5, atol=0, what=f"{tag} total")
        total.backward()
        _check_grads(enc, U.sub(d, f"{tag}/enc_grad"), rtol=2e-3, atol=1e-6, what=f"{tag} enc")
        _check_grads(dec, U.sub(d, f"{tag}/dec_grad"), rtol=2e-3, atol=1e-6, what=f"{tag} dec")
        with torch.no_grad():
            mu, sigma, z, lat = enc(xin, None)
            close(mu, d[f"{tag}/mu"], rtol=1e-4, atol=1e-6, what=f"{tag} mu")
            close(sigma, d[f"{tag}/sigma"], rtol=1e-4, atol=1e-6, what=f"{tag} sigma")
            close(z, d[f"{tag}/z"], rtol=1e-4, atol=1e-6, what=f"{tag} z")
            if c["infer"]:
                close(lat, d[f"{tag}/latlib"], rtol=1e-4, atol=1e-6, what=f"{tag} latent_lib")
            outs = dec(z, None, None if lib is None else lib.to(DEV), lat)
            if c["dist"] == "gau":
                close(outs, d[f"{tag}/px"], rtol=1e-4, atol=1e-5, what=f"{tag} px")
            else:
                close(outs[0], d[f"{tag}/scale"], rtol=1e-4, atol=1e-8, what=f"{tag} scale")
                close(outs[1], d[f"{tag}/theta"], rtol=1e-4, atol=1e-7, what=f"{tag} theta")
                close(outs[2], d[f"{tag}/rate"], rtol=1e-4, atol=1e-6, what=f"{tag} rate")
                if c["dist"] == "zinb":
                    close(outs[3], d[f"{tag}/pi"], rtol=1e-4, atol=1e-5, what=f"{tag} pi")


def test_inference_heads_golden(S):
    """The decoder heads predict() runs (scp_ber_prob, scp_relu_softmax_rows) through the modules' no-grad forward and
    predict_head, over every Bernoulli / raw-count decoder variant of the reference fixtures: all returned tensors in
    fp32, the kept output [0] also as bf16 (one bf16 rounding of the fp32 result)."""
    d = U.load("mod_ber")
    for tag in json.loads(str(d["cases"])):
        c = U.cfg(d, f"{tag}/cfg")
        cov = U.opt(d, f"{tag}/cov")
        dec = S.pkg.Output_Module_Ber(output_dim=64, output_batch_num=0 if cov is None else 3, hidden_layer=[16, 8],
                                      infer_library_size=c["latlib"], sample_factor=c["sf"], feature_factor=c["ff"])
        dec = _load(dec, U.sub(d, f"{tag}/sd")).eval()
        h = torch.tensor(d[f"{tag}/h"]).to(DEV)
        lib, lat = U.opt(d, f"{tag}/lib"), U.opt(d, f"{tag}/latlib")
        args = (h, None if cov is None else cov.to(DEV), None if lib is None else lib.to(DEV), None if lat is None else lat.to(DEV))
        with torch.no_grad():
            res = dec(*args)
        close(res[0], d[f"{tag}/out0"], rtol=2e-5, atol=1e-7, what=f"{tag} out0")
        close(res[1], d[f"{tag}/out1"], rtol=2e-5, atol=1e-7, what=f"{tag} out1")
        p32 = dec.predict_head(*args)
        assert torch.equal(p32, res[0])
        p16 = dec.predict_head(*args, out_dtype=torch.bfloat16)
        assert p16.dtype == torch.bfloat16 and torch.equal(p16, p32.bfloat16())
    d = U.load("mod_raw")
    for tag in json.loads(str(d["cases"])):
        c = U.cfg(d, f"{tag}/cfg")
        cov = U.opt(d, f"{tag}/cov")
        dec = S.pkg.Output_Module_Raw(output_dim=32, output_batch_num=0 if cov is None else 3, hidden_layer=[16, 8],
                                      infer_library_size=not c["sf"], sample_factor=c["sf"], feature_factor=c["ff"],
                                      zero_inflated=(c["lik"] == "zinb"), dispersion=c["disp"])
        dec = _load(dec, U.sub(d, f"{tag}/sd")).eval()
        h = torch.tensor(d[f"{tag}/h"]).to(DEV)
        lib, lat = U.opt(d, f"{tag}/lib"), U.opt(d, f"{tag}/latlib")
        args = (h, None if cov is None else cov.to(DEV), None if lib is None else lib.to(DEV), None if lat is None else lat.to(DEV))
        with torch.no_grad():
            scale, theta, rate, pi = dec(*args)
        close(scale, d[f"{tag}/scale"], rtol=2e-5, atol=1e-8, what=f"{tag} scale")
        close(theta, d[f"{tag}/theta"], rtol=2e-5, atol=1e-7, what=f"{tag} theta")
        close(rate, d[f"{tag}/rate"], rtol=2e-5, atol=1e-7, what=f"{tag} rate")
        if c["lik"] == "zinb":
            close(pi, d[f"{tag}/pi"], rtol=1e-4, atol=1e-5, what=f"{tag} pi")
        p32 = dec.predict_head(*args)
        assert torch.equal(p32, scale)
        assert torch.equal(dec.predict_head(*args, out_dtype=torch.bfloat16), p32.bfloat16())
