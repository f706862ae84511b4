#!/usr/bin/env python3
"""Generate golden fixtures by IMPORTING the reference (this container only).

Run:  python tools/make_goldens.py            (writes tests/golden/*.npz, *.json)

The reference package lives read-only under /root/reference and cannot travel to the GPU box;
only the DATA written here (inputs + expected outputs, fp32/fp64 arrays) is committed.  jax,
numpyro, anndata, scanpy and scvi are absent in the image and are needed by the reference only at
import time (scVI_distribution.py:11-17, utils.py:21-23), never on the hot path, so empty stub
modules are registered for them.

Every fixture records the torch version that produced it.
"""
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Dummy:
    def __init__(self, *a, **k):
        pass


def import_reference():
    jax = _stub("jax")
    jax.numpy = _stub("jax.numpy", ndarray=_Dummy, log=None)
    jax.scipy = _stub("jax.scipy")
    jax.scipy.special = _stub("jax.scipy.special", gammaln=None)
    npr = _stub("numpyro")
    npr.distributions = _stub(
        "numpyro.distributions", NegativeBinomial2=_Dummy,
        constraints=types.SimpleNamespace(positive=None, nonnegative_integer=None))
    _stub("numpyro.distributions.util", promote_shapes=lambda *a: a, validate_sample=lambda f: f)
    _stub("anndata")
    _stub("scanpy")
    _stub("scvi")
    sys.path.insert(0, "/root/reference")
    import scpair  # noqa
    return scpair


ref = import_reference()
import torch  # noqa: E402
from torch import nn  # noqa: E402

from scpair_amd import synth  # noqa: E402

torch.set_num_threads(8)
META = {"torch": torch.__version__, "numpy": np.__version__}


def save(name, **arrays):
    arrays = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
              for k, v in arrays.items() if v is not None}
    arrays["__torch_version__"] = np.array(torch.__version__)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    print("wrote", name, {k: v.shape for k, v in list(arrays.items())[:6]}, "...")


# --------------------------------------------------------------------------------------
# 1. likelihood KAT grids (values + autograd gradients), fp32 and fp64
# --------------------------------------------------------------------------------------
def gold_likelihood():
    xs = np.array([0, 1, 2, 5, 50, 1000], dtype=np.float64)
    mus = np.array([1e-8, 1e-4, 1e-2, 0.5, 3.0, 40.0, 1e3, 1e4], dtype=np.float64)
    ths = np.array([1e-4, 1e-2, 0.3, 1.0, 7.5, 1e2, 1e4, 1e6], dtype=np.float64)
    pis = np.array([-20, -3, -0.5, 0, 0.5, 3, 20], dtype=np.float64)
    X, M, T, P = np.meshgrid(xs, mus, ths, pis, indexing="ij")
    X, M, T, P = [a.reshape(-1) for a in (X, M, T, P)]
    out = {"x": X, "mu": M, "theta": T, "pi": P}
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        x = torch.tensor(X, dtype=dt)
        mu = torch.tensor(M, dtype=dt, requires_grad=True)
        th = torch.tensor(T, dtype=dt, requires_grad=True)
        pi = torch.tensor(P, dtype=dt, requires_grad=True)
        nb = ref.log_nb_positive(x, mu, th)
        gm, gt = torch.autograd.grad(nb.sum(), [mu, th])
        out.update({f"nb_{tag}": nb, f"nb_dmu_{tag}": gm, f"nb_dtheta_{tag}": gt})
        z = ref.log_zinb_positive(x, mu, th, pi)
        gm, gt, gp = torch.autograd.grad(z.sum(), [mu, th, pi])
        out.update({f"zinb_{tag}": z, f"zinb_dmu_{tag}": gm, f"zinb_dtheta_{tag}": gt,
                    f"zinb_dpi_{tag}": gp})
    # SURVEY KATs
    x = torch.tensor([0., 1., 5.]); mu = torch.tensor([.5, 2., 3.]); th = torch.tensor([1., 2., .5])
    out["kat_nb"] = ref.log_nb_positive(x, mu, th)
    out["kat_zinb"] = ref.log_zinb_positive(x, mu, th, torch.tensor([0., -1., 2.]))
    save("kat_likelihood", **out)

    # BCE (binarized_loss) incl. clamp at -100 and NaN mask; MSE (scaled_loss) with NaNs
    rng = np.random.default_rng(3)
    p = rng.uniform(0, 1, size=(16, 24)).astype(np.float32)
    p[0, :6] = [0.0, 1.0, 1e-9, 1 - 1e-7, 1e-45, 0.5]
    xb = (rng.uniform(size=p.shape) < 0.3).astype(np.float32)
    xb[0, :6] = [1, 0, 1, 0, 1, 1]
    xb[rng.uniform(size=p.shape) < 0.1] = np.nan
    xb[0, :6] = [1, 0, 1, 0, 1, 1]
    pt = torch.tensor(p, requires_grad=True)
    loss = ref.binarized_loss(torch.tensor(xb), pt)
    (gp,) = torch.autograd.grad(loss, pt)
    mu = rng.normal(size=(16, 24)).astype(np.float32)
    xm = rng.normal(size=(16, 24)).astype(np.float32)
    xm[rng.uniform(size=xm.shape) < 0.2] = np.nan
    mut = torch.tensor(mu, requires_grad=True)
    lm = ref.scaled_loss(torch.tensor(xm), mut)
    (gmu,) = torch.autograd.grad(lm, mut)
    save("kat_bce_mse", bce_p=p, bce_x=xb, bce_loss=loss, bce_dp=gp,
         mse_mu=mu, mse_x=xm, mse_loss=lm, mse_dmu=gmu,
         kat_bce=ref.binarized_loss(torch.tensor([0., 1., 1.]), torch.tensor([0.2, 0.9, 1e-9])),
         kat_mse=ref.scaled_loss(torch.tensor([1., float("nan"), 3.]), torch.tensor([0., 5., 1.])))


# --------------------------------------------------------------------------------------
# 2. per-module forward/backward
# --------------------------------------------------------------------------------------
def sd_arrays(module, prefix):
    return {f"{prefix}/{k}": v.clone() for k, v in module.state_dict().items()}


def grad_arrays(module, prefix):
    return {f"{prefix}/{k}": (p.grad.clone() if p.grad is not None else None)
            for k, p in module.named_parameters()}


B, G, P, H = 8, 32, 64, [16, 8]


def toy_counts(seed, n=B, g=G, lam=1.5):
    rng = np.random.default_rng(seed)
    x = rng.poisson(lam * rng.gamma(0.5, 1.0, size=g)[None, :] * rng.gamma(5, .2, size=n)[:, None])
    return torch.tensor(x.astype(np.float32))


def toy_binary(seed, n=B, p=P, d=0.15):
    rng = np.random.default_rng(seed)
    return torch.tensor((rng.uniform(size=(n, p)) < d).astype(np.float32))


def toy_cov(seed, n=B, c=3):
    rng = np.random.default_rng(seed)
    lab = rng.integers(0, c, size=n)
    return torch.tensor(np.eye(c, dtype=np.float32)[lab])


def gold_input_module():
    cases = []
    for lib in (False, True):
        for cov in (False, True):
            for bn in (False, True):
                for extra in (False, True):
                    cases.append(dict(lib=lib, cov=cov, bn=bn, extra=extra))
    out = {}
    names = []
    for ci, c in enumerate(cases):
        ref.set_seed(ci)
        enc = ref.Input_Module(input_dim=G, input_batch_num=3 if c["cov"] else 0, hidden_layer=H,
                               activation=nn.LeakyReLU(), layernorm=True, batchnorm=c["bn"],
                               dropout_rate=0.0, add_linear_layer=c["extra"], infer_library_size=c["lib"])
        enc.apply(ref.init_weights)
        # make LN / bias params non-trivial so their gradients are exercised
        with torch.no_grad():
            for n_, p_ in enc.named_parameters():
                if p_.ndim == 1:
                    p_.add_(0.1 * torch.randn_like(p_))
        enc.train()
        tag = f"c{ci}"
        out.update(sd_arrays(enc, f"{tag}/sd"))     # BEFORE the training-mode forward (BN running stats)
        x = toy_counts(100 + ci)
        cov = toy_cov(200 + ci) if c["cov"] else None
        rep, lat = enc(x, cov)
        w_rep = torch.randn_like(rep)
        loss = (rep * w_rep).sum()
        w_lat = None
        if lat is not None:
            w_lat = torch.randn_like(lat)
            loss = loss + (lat * w_lat).sum()
        loss.backward()
        names.append(tag)
        out.update(grad_arrays(enc, f"{tag}/grad"))
        out.update({f"{tag}/x": x, f"{tag}/cov": cov, f"{tag}/rep": rep, f"{tag}/lat": lat,
                    f"{tag}/w_rep": w_rep, f"{tag}/w_lat": w_lat,
                    f"{tag}/cfg": np.array(json.dumps(c))})
        if c["bn"]:
            # running stats after the training-mode forward
            out.update({f"{tag}/sd_after/{k}": v.clone() for k, v in enc.state_dict().items()
                        if "running" in k or "num_batches" in k})
        # eval-mode forward too
        enc.eval()
        with torch.no_grad():
            rep_e, lat_e = enc(x, cov)
        out.update({f"{tag}/rep_eval": rep_e, f"{tag}/lat_eval": lat_e})
    out["cases"] = np.array(json.dumps(names))
    save("mod_input", **out)


def gold_ber():
    out, names = {}, []
    ci = 0
    for sf in (False, True):
        for ff in (False, True):
            for use_latlib in (False, True):
                for cov in (False, True):
                    for nan in (False, True):
                        if nan and (cov or ff):
                            continue
                        ref.set_seed(50 + ci)
                        dec = ref.Output_Module_Ber(output_dim=P, output_batch_num=3 if cov else 0,
                                                    hidden_layer=H, infer_library_size=use_latlib,
                                                    sample_factor=sf, feature_factor=ff)
                        dec.apply(ref.init_weights)
                        with torch.no_grad():
                            dec.output_layer.bias.add_(0.3 * torch.randn(P))
                            if ff:
                                dec.feature_scalar.add_(0.5 * torch.randn(P))
                        h = torch.randn(B, H[-1], requires_grad=True)
                        cv = toy_cov(300 + ci) if cov else None
                        y = toy_binary(400 + ci)
                        if nan:
                            y[1, 3] = float("nan"); y[5, 60] = float("nan"); y[7, 0] = float("nan")
                        lib = torch.tensor(np.random.default_rng(ci).uniform(0.1, 3, size=(B, 1)).astype(np.float32)) if sf else None
                        latlib = torch.randn(B, 1, requires_grad=True) if use_latlib else None
                        res = dec(h, cv, lib, None if sf else latlib)
                        loss = ref.binarized_loss(y, res[0])
                        loss.backward()
                        tag = f"c{ci}"
                        names.append(tag)
                        out.update(sd_arrays(dec, f"{tag}/sd"))
                        out.update(grad_arrays(dec, f"{tag}/grad"))
                        out.update({f"{tag}/h": h.detach(), f"{tag}/cov": cv, f"{tag}/y": y, f"{tag}/lib": lib,
                                    f"{tag}/latlib": latlib.detach() if (latlib is not None and not sf) else None,
                                    f"{tag}/out0": res[0], f"{tag}/out1": res[1], f"{tag}/loss": loss,
                                    f"{tag}/dh": h.grad,
                                    f"{tag}/dlatlib": latlib.grad if (latlib is not None and latlib.grad is not None) else None,
                                    f"{tag}/cfg": np.array(json.dumps(dict(sf=sf, ff=ff, latlib=bool(use_latlib and not sf), cov=cov, nan=nan)))})
                        ci += 1
    out["cases"] = np.array(json.dumps(names))
    save("mod_ber", **out)


def gold_raw():
    out, names = {}, []
    ci = 0
    for lik in ("nb", "zinb"):
        for disp in ("feature-cell", "feature"):
            for sf in (True, False):
                for ff in (False, True):
                    for cov in (False, True):
                        if cov and (ff or disp == "feature"):
                            continue
                        ref.set_seed(70 + ci)
                        dec = ref.Output_Module_Raw(output_dim=G, output_batch_num=3 if cov else 0,
                                                    hidden_layer=H, infer_library_size=not sf,
                                                    sample_factor=sf, feature_factor=ff,
                                                    zero_inflated=(lik == "zinb"), dispersion=disp)
                        dec.apply(ref.init_weights)
                        with torch.no_grad():
                            dec.output_mean[0].bias.add_(0.3 * torch.randn(G))
                            if ff:
                                dec.feature_scalar.add_(0.5 * torch.randn(G))
                            if lik == "zinb":
                                dec.output_dropout.bias.add_(0.2 * torch.randn(G))
                        h = torch.randn(B, H[-1], requires_grad=True)
                        cv = toy_cov(500 + ci) if cov else None
                        y = toy_counts(600 + ci)
                        lib = y.sum(1, keepdim=True) if sf else None
                        latlib = None if sf else torch.randn(B, 1, requires_grad=True)
                        scale, theta, rate, pi = dec(h, cv, lib, latlib)
                        loss = ref.raw_loss(y, rate, theta, scale, pi, likelihood=lik)
                        loss.backward()
                        tag = f"c{ci}"
                        names.append(tag)
                        out.update(sd_arrays(dec, f"{tag}/sd"))
                        out.update(grad_arrays(dec, f"{tag}/grad"))
                        out.update({f"{tag}/h": h.detach(), f"{tag}/cov": cv, f"{tag}/y": y, f"{tag}/lib": lib,
                                    f"{tag}/latlib": latlib.detach() if latlib is not None else None,
                                    f"{tag}/scale": scale, f"{tag}/theta": theta, f"{tag}/rate": rate, f"{tag}/pi": pi,
                                    f"{tag}/loss": loss, f"{tag}/dh": h.grad,
                                    f"{tag}/dlatlib": latlib.grad if latlib is not None else None,
                                    f"{tag}/cfg": np.array(json.dumps(dict(lik=lik, disp=disp, sf=sf, ff=ff, cov=cov)))})
                        ci += 1
    out["cases"] = np.array(json.dumps(names))
    save("mod_raw", **out)


def gold_gau_map():
    out = {}
    # Output_Module_Gau (trainer always builds it with infer_library_size=False, feature_factor=False)
    for ci, (ff, cov) in enumerate([(False, False), (True, False), (False, True)]):
        ref.set_seed(90 + ci)
        dec = ref.Output_Module_Gau(output_dim=G, output_batch_num=3 if cov else 0, hidden_layer=H,
                                    infer_library_size=False, feature_factor=ff)
        dec.apply(ref.init_weights)
        with torch.no_grad():
            if ff:
                dec.feature_scalar.add_(1 + 0.5 * torch.randn(G))
        h = torch.randn(B, H[-1], requires_grad=True)
        cv = toy_cov(700 + ci) if cov else None
        y = torch.randn(B, G)
        y[2, 5] = float("nan")
        res = dec(h, cv, None, None)
        loss = ref.scaled_loss(y, res)
        loss.backward()
        tag = f"gau{ci}"
        out.update(sd_arrays(dec, f"{tag}/sd")); out.update(grad_arrays(dec, f"{tag}/grad"))
        out.update({f"{tag}/h": h.detach(), f"{tag}/cov": cv, f"{tag}/y": y, f"{tag}/out": res,
                    f"{tag}/loss": loss, f"{tag}/dh": h.grad,
                    f"{tag}/cfg": np.array(json.dumps(dict(ff=ff, cov=cov)))})
    # Module_Module
    for ci, (hid, nonneg, ln, bn) in enumerate([(None, False, False, False), (None, True, False, False),
                                                ([12], False, True, False), ([12, 10], False, False, True)]):
        ref.set_seed(95 + ci)
        net = ref.Module_Module(origin_module_dim=H[-1], target_module_dim=H[-1],
                                translational_hidden_nodes=hid, non_neg=nonneg, activation=nn.LeakyReLU(),
                                layernorm=ln, batchnorm=bn, dropout_rate=0)
        net.apply(ref.init_weights)
        net.train()
        x = torch.randn(B, H[-1], requires_grad=True)
        y = torch.randn(B, H[-1])
        pred = net(x)
        loss = ref.scaled_loss(y, pred)
        loss.backward()
        tag = f"map{ci}"
        out.update(sd_arrays(net, f"{tag}/sd")); out.update(grad_arrays(net, f"{tag}/grad"))
        out.update({f"{tag}/x": x.detach(), f"{tag}/y": y, f"{tag}/pred": pred, f"{tag}/loss": loss,
                    f"{tag}/dx": x.grad,
                    f"{tag}/cfg": np.array(json.dumps(dict(hid=hid, nonneg=nonneg, ln=ln, bn=bn)))})
    save("mod_gau_map", **out)


def gold_vae():
    out = {}
    L = 6
    hid = [16]
    ci = 0
    for dist in ("nb", "zinb", "gau"):
        for disp in ("feature-cell", "feature"):
            for lib_mode in ("sample", "infer"):
                if dist == "gau" and (disp == "feature" or lib_mode == "infer"):
                    continue
                infer = lib_mode == "infer"
                ref.set_seed(110 + ci)
                enc = ref.VAE_ENCODER(input_dim=G, input_batch_num=0, hidden_dim=hid, latent_dim=L,
                                      layernorm=True, activation=nn.LeakyReLU(), batchnorm=False,
                                      dropout_rate=0, infer_library_size=infer)
                dec = ref.VAE_DECODER(input_dim=G, input_batch_num=0, hidden_dim=hid, latent_dim=L,
                                      layernorm=True, activation=nn.LeakyReLU(), batchnorm=False,
                                      dropout_rate=0, infer_library_size=infer, sample_factor=not infer,
                                      feature_factor=False, distribution=dist, dispersion=disp)
                enc.apply(ref.init_weights); dec.apply(ref.init_weights)
                enc.train(); dec.train()
                x = toy_counts(800 + ci) if dist != "gau" else torch.randn(B, G)
                lib = x.sum(1, keepdim=True) if (dist != "gau" and not infer) else None
                eps = torch.randn(B, L)
                orig = torch.randn_like
                torch.randn_like = lambda t, *a, **k: eps.clone()
                try:
                    mu, sigma, z, latlib = enc(x, None)
                    outputs = dec(z, None, lib, latlib)
                finally:
                    torch.randn_like = orig
                kl = torch.distributions.kl_divergence(torch.distributions.Normal(mu, sigma),
                                                       torch.distributions.Normal(0, 1)).mean()
                if dist == "gau":
                    recon = ref.scaled_loss(x, outputs)
                    outs = {"px": outputs}
                else:
                    scale, theta, rate, pi = outputs
                    recon = ref.raw_loss(x, rate, theta, scale, pi, likelihood=dist)
                    outs = {"scale": scale, "theta": theta, "rate": rate, "pi": pi}
                total = 0.7 * kl + recon
                total.backward()
                tag = f"c{ci}"
                out.update(sd_arrays(enc, f"{tag}/enc_sd")); out.update(grad_arrays(enc, f"{tag}/enc_grad"))
                out.update(sd_arrays(dec, f"{tag}/dec_sd")); out.update(grad_arrays(dec, f"{tag}/dec_grad"))
                out.update({f"{tag}/x": x, f"{tag}/lib": lib, f"{tag}/eps": eps, f"{tag}/mu": mu, f"{tag}/sigma": sigma,
                            f"{tag}/z": z, f"{tag}/latlib": latlib, f"{tag}/kl": kl, f"{tag}/recon": recon,
                            f"{tag}/total": total,
                            f"{tag}/cfg": np.array(json.dumps(dict(dist=dist, disp=disp, infer=infer, kl_weight=0.7)))})
                out.update({f"{tag}/{k}": v for k, v in outs.items()})
                ci += 1
    out["ncases"] = np.array(ci)
    save("mod_vae", **out)


# --------------------------------------------------------------------------------------
# 3. weight-decay groups + AdamW trajectories
# --------------------------------------------------------------------------------------
def gold_optimizer():
    groups = {}
    ref.set_seed(0)
    enc = ref.Input_Module(input_dim=G, input_batch_num=0, hidden_layer=H, activation=nn.LeakyReLU(),
                           layernorm=True, batchnorm=True, dropout_rate=0.1, add_linear_layer=True,
                           infer_library_size=True)
    _, _, d, nd = ref.add_weight_decay(enc, output_layer=["output"])
    groups["Input_Module"] = {"decay": d, "nodecay": nd, "ndim": {k: v.ndim for k, v in enc.named_parameters()}}
    raw = ref.Output_Module_Raw(output_dim=G, output_batch_num=0, hidden_layer=H, infer_library_size=False,
                                sample_factor=True, feature_factor=True, zero_inflated=True, dispersion="feature-cell")
    _, _, d, nd = ref.add_weight_decay(raw, output_layer=["output"])
    groups["Output_Module_Raw"] = {"decay": d, "nodecay": nd, "ndim": {k: v.ndim for k, v in raw.named_parameters()}}
    ber = ref.Output_Module_Ber(output_dim=P, output_batch_num=0, hidden_layer=H, infer_library_size=True,
                                sample_factor=False, feature_factor=True)
    _, _, d, nd = ref.add_weight_decay(ber, output_layer=["output"])
    groups["Output_Module_Ber"] = {"decay": d, "nodecay": nd, "ndim": {k: v.ndim for k, v in ber.named_parameters()}}
    gau = ref.Output_Module_Gau(output_dim=G, output_batch_num=0, hidden_layer=H, infer_library_size=False, feature_factor=True)
    _, _, d, nd = ref.add_weight_decay(gau, output_layer=["output"])
    groups["Output_Module_Gau"] = {"decay": d, "nodecay": nd, "ndim": {k: v.ndim for k, v in gau.named_parameters()}}
    mp = ref.Module_Module(origin_module_dim=8, target_module_dim=8, translational_hidden_nodes=[12], non_neg=False,
                           layernorm=True, activation=nn.LeakyReLU(), batchnorm=False, dropout_rate=0)
    _, _, d, nd = ref.add_weight_decay(mp, output_layer="bias")
    groups["Module_Module[bias]"] = {"decay": d, "nodecay": nd, "ndim": {k: v.ndim for k, v in mp.named_parameters()}}
    groups["__torch_version__"] = torch.__version__
    with open(os.path.join(OUT, "weight_decay_groups.json"), "w") as f:
        json.dump(groups, f, indent=1)
    print("wrote weight_decay_groups.json")

    # AdamW trajectory: 5 steps, two groups (decay 1e-2 to make decay visible, and 0)
    rng = np.random.default_rng(5)
    p0 = [rng.normal(size=(7, 5)).astype(np.float32), rng.normal(size=(11,)).astype(np.float32)]
    params = [nn.Parameter(torch.tensor(a)) for a in p0]
    opt = torch.optim.AdamW([{"params": [params[0]], "weight_decay": 1e-2, "lr": 1e-3},
                             {"params": [params[1]], "weight_decay": 0, "lr": 1e-3}])
    out = {"init_0": p0[0], "init_1": p0[1]}
    for step in range(5):
        gs = [(rng.normal(size=a.shape) * (10.0 ** rng.integers(-4, 2))).astype(np.float32) for a in p0]
        for p_, g_ in zip(params, gs):
            p_.grad = torch.tensor(g_)
        opt.step()
        for i, (p_, g_) in enumerate(zip(params, gs)):
            out[f"g{step}_{i}"] = g_
            out[f"p{step}_{i}"] = p_.detach().clone()
    st = opt.state[params[0]]
    out["m_final_0"] = st["exp_avg"]; out["v_final_0"] = st["exp_avg_sq"]
    save("adamw_traj", **out)


# --------------------------------------------------------------------------------------
# 4. training trajectories through the reference's own loops
# --------------------------------------------------------------------------------------
class Recorder:
    """Patches the reference namespaces to record what its loops do (no reference code changes)."""

    def __init__(self):
        self.train_losses = []      # per direction list of per-step losses
        self.val_losses = []
        self.batches = []           # per direction list of per-step row-index arrays
        self.init_sd = []           # snapshots at add_weight_decay time
        self.cur_sig = None

    def install(self, train_dict):
        rec = self
        u, m = ref.utils, ref.model
        self._orig = dict(raw=u.raw_loss, ber=u.binarized_loss, sc=u.scaled_loss, msc=m.scaled_loss,
                          awd=m.add_weight_decay, tn=m.train_net)

        def wrap(fn):
            def inner(*a, **k):
                v = fn(*a, **k)
                (rec.train_losses if torch.is_grad_enabled() else rec.val_losses)[-1].append(float(v.item()))
                return v
            return inner
        u.raw_loss = wrap(self._orig["raw"]); u.binarized_loss = wrap(self._orig["ber"])
        u.scaled_loss = wrap(self._orig["sc"]); m.scaled_loss = wrap(self._orig["msc"])

        def awd(model, output_layer=()):
            rec.init_sd.append({k: v.detach().clone().numpy() for k, v in model.state_dict().items()})
            return rec._orig["awd"](model, output_layer=output_layer)
        m.add_weight_decay = awd

        sig_vec = {}
        for key, t in train_dict.items():
            if key.endswith("_mtx"):
                g = torch.Generator().manual_seed(99)
                v = torch.randn(t.shape[1], generator=g, dtype=torch.float64)
                s = (t.double() @ v).numpy()
                sig_vec[t.shape[1]] = (v, {round(float(a), 6): i for i, a in enumerate(s)})
                assert len(sig_vec[t.shape[1]][1]) == t.shape[0], "row signatures collide"

        def tn(input, *a, **k):
            v, table = sig_vec[input.shape[1]]
            s = (input.double() @ v).numpy()
            rec.batches[-1].append(np.array([table[round(float(q), 6)] for q in s], dtype=np.int64))
            return rec._orig["tn"](input, *a, **k)
        m.train_net = tn

    def new_direction(self):
        self.train_losses.append([]); self.val_losses.append([]); self.batches.append([])

    def uninstall(self):
        u, m = ref.utils, ref.model
        u.raw_loss, u.binarized_loss, u.scaled_loss = self._orig["raw"], self._orig["ber"], self._orig["sc"]
        m.scaled_loss, m.add_weight_decay, m.train_net = self._orig["msc"], self._orig["awd"], self._orig["tn"]


class _Meta:
    def __init__(self, n):
        self.index = np.arange(n)


def dense_dicts(data, with_cov=False):
    rna, atac = data["rna"], data["atac"]
    dicts = []
    for split in ("train", "val", "test"):
        idx = data[split]
        d = {}
        for name, mat in (("Gene Expression", rna), ("Peaks", atac)):
            mtx = torch.tensor(np.asarray(mat[idx].todense(), dtype=np.float32))
            d[f"{name}_{split}_mtx"] = mtx
            d[f"{name}_{split}_cov"] = (torch.tensor(data["cov"][idx]) if with_cov else None)
            d[f"{name}_{split}_lib"] = mtx.sum(1).reshape(-1, 1)
        dicts.append(d)
    return dicts


def gold_trajectory(name, modalities, batchnorm=False, zero_inflated=True, with_cov=False, epochs=6,
                    n=512, g=64, p=128, hidden=(16, 8), batch=64, seed=0, patience=25, scheduler=None, lr=1e-3,
                    save_files=False):
    """patience / scheduler / lr / save_files: the epoch-control branches (model.py:592-602, 783-807): early stopping that
    fires, the 'StepLR' scheduler, and the files torch.save() leaves behind (names and sizes recorded in cfg)."""
    data = synth.make_paired(n, g, p, seed=seed, p_atac=0.1)
    if with_cov:
        lab = np.random.default_rng(7).integers(0, 4, size=n)
        data["cov"] = np.eye(4, dtype=np.float32)[lab]
    tr, va, te = dense_dicts(data, with_cov)
    rec = Recorder()
    rec.install(tr)
    # the reference loops over modalities; start a new recording slot each time the DataLoader is built
    orig_dl = ref.model.DataLoader
    import torch.utils.data as tud
    orig_tud_dl = tud.DataLoader

    class DL(orig_tud_dl):
        def __init__(self, *a, **k):
            rec.new_direction()
            super().__init__(*a, **k)
    tud.DataLoader = DL      # model.py re-imports DataLoader from torch.utils.data inside the loop
    ref.model.DataLoader = DL
    import tempfile
    save_dir = tempfile.mkdtemp(prefix="scpair_gold_") if save_files else None
    try:
        obj = ref.scPair_object(scobj=None, cov=(["labels"] if with_cov else None), modalities=modalities, SEED=0,
                                hidden_layer=list(hidden), dropout_rate=0.0, batchnorm=batchnorm,
                                layernorm=True, batch_size=batch, max_epochs=epochs,
                                device=torch.device("cpu"), save_model=save_files, save_path=save_dir,
                                zero_inflated=zero_inflated, early_stopping_patience=patience, weight_decay=scheduler,
                                learning_rate_prediction=lr, mapping_learning_rate=lr)
        enc_d, dec_d = obj.train_predicting_networks(tr, va, te)
        obj.encoder_dict, obj.decoder_dict = enc_d, dec_d
        obj.data_loader_dict_train, obj.data_loader_dict_val, obj.data_loader_dict_test = tr, va, te
        obj.train_metadata, obj.val_metadata, obj.test_metadata = _Meta(len(data["train"])), _Meta(len(data["val"])), _Meta(len(data["test"]))
        emb, _ = obj.reference_embeddings()
        n_dir = len(rec.train_losses)
        obj.mapping_dict = obj.train_bidirectional_mapping()
        memb, _ = obj.mapped_embeddings()
        preds = obj.predict()
    finally:
        tud.DataLoader = orig_tud_dl
        ref.model.DataLoader = orig_dl
        rec.uninstall()
    saved = {}
    if save_files:
        import shutil
        saved = {f: os.path.getsize(os.path.join(save_dir, f)) for f in sorted(os.listdir(save_dir))}
        shutil.rmtree(save_dir)
    out = {"cfg": np.array(json.dumps(dict(modalities=modalities, batchnorm=batchnorm, zero_inflated=zero_inflated,
                                           with_cov=with_cov, epochs=epochs, n=n, g=g, p=p, hidden=list(hidden),
                                           batch=batch, seed=seed, p_atac=0.1, patience=patience, scheduler=scheduler,
                                           lr=lr, saved_files=saved)))}
    if with_cov:
        out["cov"] = data["cov"]
    names = list(modalities.keys())
    # predicting networks: 2 directions; init_sd order: enc0, dec0, enc1, dec1, map0, map1
    for d in range(2):
        tag = f"dir{d}"
        out[f"{tag}/train_loss"] = np.array(rec.train_losses[d], dtype=np.float64)
        out[f"{tag}/val_loss"] = np.array(rec.val_losses[d], dtype=np.float64)
        bl = rec.batches[d]
        out[f"{tag}/batch_sizes"] = np.array([len(b) for b in bl])
        out[f"{tag}/batch_rows"] = np.concatenate(bl)
        for k, v in rec.init_sd[2 * d].items():
            out[f"{tag}/enc_init/{k}"] = v
        for k, v in rec.init_sd[2 * d + 1].items():
            out[f"{tag}/dec_init/{k}"] = v
        key = names[d] + " to " + names[1 - d]
        for k, v in enc_d[key].state_dict().items():
            out[f"{tag}/enc_final/{k}"] = v
        for k, v in dec_d[key].state_dict().items():
            out[f"{tag}/dec_final/{k}"] = v
    for d in range(2):
        tag = f"map{d}"
        out[f"{tag}/train_loss"] = np.array(rec.train_losses[n_dir + d], dtype=np.float64)
        out[f"{tag}/val_loss"] = np.array(rec.val_losses[n_dir + d], dtype=np.float64)
        for k, v in rec.init_sd[4 + d].items():
            out[f"{tag}/init/{k}"] = v
        key = names[d] + "_to_" + names[1 - d]
        for k, v in obj.mapping_dict[key].state_dict().items():
            out[f"{tag}/final/{k}"] = v
    for k, v in emb.items():
        out[f"emb/{k}"] = v
    for k, v in memb.items():
        out[f"memb/{k}"] = v
    for k, v in preds.items():
        out[f"pred/{k}"] = v
    save(name, **out)


def gold_vae_trajectory():
    """The package ships no VAE epoch loop; drive train_VAE/eval_VAE with a fixed
    batch order and a recorded epsilon stream so the GPU path can replay it."""
    n, g, L, hid, batch, epochs = 256, 48, 6, [16], 64, 3
    rna = synth.make_rna(n, g, seed=4)
    x_all = torch.tensor(np.asarray(rna.todense(), dtype=np.float32))
    lib_all = x_all.sum(1, keepdim=True)
    ref.set_seed(0)
    enc = ref.VAE_ENCODER(input_dim=g, input_batch_num=0, hidden_dim=hid, latent_dim=L, layernorm=True,
                          activation=nn.LeakyReLU(), batchnorm=False, dropout_rate=0, infer_library_size=False)
    ref.set_seed(0)
    dec = ref.VAE_DECODER(input_dim=g, input_batch_num=0, hidden_dim=hid, latent_dim=L, layernorm=True,
                          activation=nn.LeakyReLU(), batchnorm=False, dropout_rate=0, infer_library_size=False,
                          sample_factor=True, feature_factor=False, distribution="nb", dispersion="feature-cell")
    enc.apply(ref.init_weights); dec.apply(ref.init_weights)
    out = {f"enc_init/{k}": v.clone() for k, v in enc.state_dict().items()}
    out.update({f"dec_init/{k}": v.clone() for k, v in dec.state_dict().items()})
    opt = torch.optim.AdamW(list(enc.parameters()) + list(dec.parameters()), lr=1e-3, weight_decay=1e-9)
    rng = np.random.default_rng(11)
    eps_stream, rows, tot = [], [], []
    orig = torch.randn_like

    def fake_randn_like(t, *a, **k):
        e = torch.tensor(rng.normal(size=tuple(t.shape)).astype(np.float32))
        eps_stream.append(e.numpy().reshape(-1))
        return e
    torch.randn_like = fake_randn_like
    try:
        for ep in range(epochs):
            perm = rng.permutation(n)
            enc.train(); dec.train()
            for b0 in range(0, n, batch):
                idx = perm[b0:b0 + batch]
                rows.append(idx)
                ref.train_VAE(x_all[idx], lib_all[idx], enc, dec, opt, kl_weight=0.5, likelihood_type="nb")
            enc.eval(); dec.eval()
            t, r, k = ref.eval_VAE(x_all, lib_all, enc, dec, kl_weight=0.5, likelihood_type="nb")
            tot.append([float(t), float(r), float(k)])
    finally:
        torch.randn_like = orig
    out.update({f"enc_final/{k}": v.clone() for k, v in enc.state_dict().items()})
    out.update({f"dec_final/{k}": v.clone() for k, v in dec.state_dict().items()})
    out["rows"] = np.concatenate(rows); out["batch_sizes"] = np.array([len(r) for r in rows])
    out["eps"] = np.concatenate(eps_stream); out["eval"] = np.array(tot)
    out["cfg"] = np.array(json.dumps(dict(n=n, g=g, L=L, hid=hid, batch=batch, epochs=epochs, kl_weight=0.5, seed=4)))
    save("traj_vae_nb", **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["all"]
    run = lambda k: ("all" in which) or (k in which)
    if run("lik"):
        gold_likelihood()
    if run("mods"):
        gold_input_module(); gold_ber(); gold_raw(); gold_gau_map(); gold_vae()
    if run("opt"):
        gold_optimizer()
    if run("traj"):
        gold_trajectory("traj_zinb_ber", {"Gene Expression": "zinb", "Peaks": "ber"})
        gold_trajectory("traj_nb_ber", {"Gene Expression": "nb", "Peaks": "ber"}, zero_inflated=False)
        gold_trajectory("traj_nb_ber_bn", {"Gene Expression": "nb", "Peaks": "ber"}, zero_inflated=False,
                        batchnorm=True, epochs=4)
        gold_trajectory("traj_nb_ber_cov", {"Gene Expression": "nb", "Peaks": "ber"}, zero_inflated=False,
                        with_cov=True, epochs=3)
        gold_vae_trajectory()
    if run("epoch"):
        # epoch control: early stopping that fires (patience 2, lr 0.02: the toy overfits within a few epochs) with the
        # saved files; and the StepLR scheduler (step 25, gamma 0.1) across its first decay
        gold_trajectory("traj_earlystop", {"Gene Expression": "nb", "Peaks": "ber"}, zero_inflated=False, epochs=40,
                        patience=2, lr=2e-2, save_files=True)
        gold_trajectory("traj_steplr", {"Gene Expression": "nb", "Peaks": "ber"}, zero_inflated=False, epochs=29,
                        patience=40, scheduler="StepLR", lr=5e-3, n=256, batch=64)
    with open(os.path.join(OUT, "META.json"), "w") as f:
        json.dump(META, f)
