"""CPU oracle for the scPair training hot path.

*** TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT ***
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.  The
product package (scpair_amd/) never imports it and has no CPU fallback.

This is an independent restatement (plain torch CPU tensor ops, functional style, parameters
passed as a {name: tensor} dict with the reference's state_dict names) of the algorithm in the
reference; every function cites the reference file:line it follows.  It is PINNED by the golden
fixtures under tests/golden/, which were produced by importing the reference itself in the build
container (tools/make_goldens.py); tests/test_oracle_golden.py checks every function here
against them.  Arithmetic below the reference (nn.Linear, LayerNorm eps 1e-5 / biased variance,
BatchNorm1d momentum 0.1, BCELoss log clamp -100, softplus threshold 20, lgamma/digamma, AdamW)
is PyTorch's (reference pins torch 2.0.1; fixtures were generated with the torch recorded inside
each fixture).

Besides the autograd-differentiable forward restatement, the *_grad functions spell out the
closed-form backward formulas the HIP kernels implement, so they can be checked against autograd
on the CPU before they are checked on the GPU.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-8


# ----------------------------------------------------------------------------------------------
# likelihoods  (scVI_distribution.py:29-105, loss.py:22-51)
# ----------------------------------------------------------------------------------------------
def log_nb(x, mu, theta, eps=EPS):
    """scVI_distribution.py:74-105 log_nb_positive."""
    l_tm = torch.log(theta + mu + eps)
    return (theta * (torch.log(theta + eps) - l_tm) + x * (torch.log(mu + eps) - l_tm)
            + torch.lgamma(x + theta) - torch.lgamma(theta) - torch.lgamma(x + 1))


def log_zinb(x, mu, theta, pi, eps=EPS):
    """scVI_distribution.py:29-71 log_zinb_positive (both branches evaluated, then 0/1 weighted)."""
    sp_pi = F.softplus(-pi)
    l_t = torch.log(theta + eps)
    l_tm = torch.log(theta + mu + eps)
    t = -pi + theta * (l_t - l_tm)
    case0 = F.softplus(t) - sp_pi
    case1 = (-sp_pi + t + x * (torch.log(mu + eps) - l_tm)
             + torch.lgamma(x + theta) - torch.lgamma(theta) - torch.lgamma(x + 1))
    return (x < eps).to(x.dtype) * case0 + (x > eps).to(x.dtype) * case1


def nb_grad(x, mu, theta, eps=EPS):
    """d log_nb / d mu, d theta (closed form used by the NB decoder kernel)."""
    s = theta + mu + eps
    dmu = -theta / s + x * (1.0 / (mu + eps) - 1.0 / s)
    dth = (torch.log(theta + eps) - torch.log(s) + theta / (theta + eps) - theta / s - x / s
           + torch.digamma(x + theta) - torch.digamma(theta))
    return dmu, dth


def zinb_grad(x, mu, theta, pi, eps=EPS):
    """d log_zinb / d mu, d theta, d pi (closed form used by the ZINB kernel)."""
    s = theta + mu + eps
    l_t = torch.log(theta + eps)
    l_tm = torch.log(s)
    t = -pi + theta * (l_t - l_tm)
    dt_dmu = -theta / s
    dt_dth = l_t - l_tm + theta / (theta + eps) - theta / s
    sig_t = torch.sigmoid(t)
    sig_npi = torch.sigmoid(-pi)             # d softplus(-pi)/d pi = -sigmoid(-pi)
    zero = (x < eps).to(x.dtype)
    nz = (x > eps).to(x.dtype)
    # zero branch: softplus(t) - softplus(-pi)
    dmu0 = sig_t * dt_dmu
    dth0 = sig_t * dt_dth
    dpi0 = -sig_t + sig_npi
    # non-zero branch: -softplus(-pi) + t + x(log(mu+eps)-log s) + lgamma terms
    dmu1 = dt_dmu + x * (1.0 / (mu + eps) - 1.0 / s)
    dth1 = dt_dth - x / s + torch.digamma(x + theta) - torch.digamma(theta)
    dpi1 = sig_npi - 1.0
    return zero * dmu0 + nz * dmu1, zero * dth0 + nz * dth1, zero * dpi0 + nz * dpi1


def bce_masked(x, p):
    """loss.py:36-41 binarized_loss: mean over non-NaN labels of BCE with log clamp at -100."""
    m = ~torch.isnan(x)
    lp = torch.clamp(torch.log(p[m]), min=-100.0)
    l1p = torch.clamp(torch.log(1.0 - p[m]), min=-100.0)
    return -(x[m] * lp + (1.0 - x[m]) * l1p).mean()


def bce_grad(x, p):
    """d bce_masked / d p : (p-x)/max(p(1-p),1e-12)/count on unmasked entries (torch BCELoss backward)."""
    m = ~torch.isnan(x)
    n = m.sum()
    xz = torch.where(m, x, torch.zeros_like(x))
    g = (p - xz) / torch.clamp(p * (1.0 - p), min=1e-12) / n
    return torch.where(m, g, torch.zeros_like(g))


def mse_masked(x, mu):
    """loss.py:22-27 scaled_loss."""
    m = ~torch.isnan(x)
    return ((mu[m] - x[m]) ** 2).mean()


def raw_loss(x, rate, theta, pi, likelihood):
    """loss.py:46-51."""
    if likelihood == "nb":
        return -log_nb(x, rate, theta).mean()
    return -log_zinb(x, rate, theta, pi).mean()


# ----------------------------------------------------------------------------------------------
# layers  (utils.py:54-64 FCNN; torch semantics of Linear / LayerNorm / BatchNorm1d)
# ----------------------------------------------------------------------------------------------
def linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


def layer_norm(x, w, b, eps=1e-5):
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * w + b


def layer_norm_grad(x, w, dy, eps=1e-5):
    """Closed-form LayerNorm backward: returns dx, dw, db."""
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + eps)
    xh = (x - mean) * rstd
    dxh = dy * w
    dx = rstd * (dxh - dxh.mean(-1, keepdim=True) - xh * (dxh * xh).mean(-1, keepdim=True))
    return dx, (dy * xh).sum(0), dy.sum(0)


def batch_norm(x, sd, prefix, training, momentum=0.1, eps=1e-5):
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    if training:
        mean = x.mean(0)
        var = ((x - mean) ** 2).mean(0)
        n = x.shape[0]
        with torch.no_grad():
            sd[prefix + ".running_mean"] = (1 - momentum) * sd[prefix + ".running_mean"] + momentum * mean
            sd[prefix + ".running_var"] = (1 - momentum) * sd[prefix + ".running_var"] + momentum * var * n / max(n - 1, 1)
            sd[prefix + ".num_batches_tracked"] = sd[prefix + ".num_batches_tracked"] + 1
    else:
        mean, var = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    return (x - mean) / torch.sqrt(var + eps) * w + b


def leaky(x, slope):
    return torch.where(x > 0, x, x * slope)


def fcnn_layout(n_layers, layernorm, batchnorm):
    """Sequential indices of (Linear, LayerNorm|None, BatchNorm|None) per layer (utils.py:54-64)."""
    per = 3 + int(layernorm) + int(batchnorm)      # Linear [LN] act [BN] Dropout
    out = []
    for i in range(n_layers):
        base = i * per
        ln = base + 1 if layernorm else None
        bn = base + 1 + int(layernorm) + 1 if batchnorm else None
        out.append((base, ln, bn))
    return out


def count_fcnn_layers(sd, prefix):
    idx = sorted({int(k[len(prefix) + 1:].split(".")[0]) for k in sd
                  if k.startswith(prefix + ".") and k.endswith(".weight") and sd[k].ndim == 2
                  and k[len(prefix) + 1:].split(".")[0].isdigit() and k.count(".") == prefix.count(".") + 2})
    return len(idx)


def fcnn(sd, prefix, x, layernorm=True, batchnorm=False, slope=0.01, training=False):
    """utils.py:54-64: per layer Linear -> [LayerNorm] -> activation -> [BatchNorm1d] -> Dropout(p=0 here)."""
    n = count_fcnn_layers(sd, prefix)
    for lin, ln, bn in fcnn_layout(n, layernorm, batchnorm):
        x = linear(x, sd[f"{prefix}.{lin}.weight"], sd[f"{prefix}.{lin}.bias"])
        if ln is not None:
            x = layer_norm(x, sd[f"{prefix}.{ln}.weight"], sd[f"{prefix}.{ln}.bias"])
        x = leaky(x, slope)
        if bn is not None:
            x = batch_norm(x, sd, f"{prefix}.{bn}", training)
    return x


# ----------------------------------------------------------------------------------------------
# modules  (model.py:60-336, pairedVAE.py:24-187)
# ----------------------------------------------------------------------------------------------
def input_module(sd, x, cov=None, layernorm=True, batchnorm=False, slope=0.01, training=False):
    """model.py:124-143 Input_Module._forward_normal -> (latent_rep, latent_lib|None)."""
    xin = x if cov is None else torch.cat([x, cov], 1)
    rep = fcnn(sd, "hidden_rep", xin, layernorm, batchnorm, slope, training)
    if "extra_linear_rep.weight" in sd:
        rep = linear(rep, sd["extra_linear_rep.weight"], sd["extra_linear_rep.bias"])
    lat = None
    if any(k.startswith("hidden_lib.") for k in sd):
        hl = fcnn(sd, "hidden_lib.0", xin, layernorm, batchnorm, slope, training)
        lat = linear(hl, sd["hidden_lib.1.weight"], sd["hidden_lib.1.bias"])
    return rep, lat


def output_ber(sd, h, cov=None, lib=None, latent_lib=None, sample_factor=True):
    """model.py:164-190 Output_Module_Ber.forward -> (result, px, sample_factor, feature_factor)."""
    hin = h if cov is None else torch.cat([h, cov], 1)
    px = torch.sigmoid(linear(hin, sd["output_layer.weight"], sd["output_layer.bias"]))
    if sample_factor:
        sf = torch.sigmoid(lib)
        res = px * sf
    elif latent_lib is not None:
        sf = torch.sigmoid(latent_lib)
        res = px * sf
    else:
        sf, res = None, px
    ff = None
    if "feature_scalar" in sd:
        ff = torch.sigmoid(sd["feature_scalar"])
        res = res * ff
    return res, px, sf, ff


def output_raw(sd, h, cov=None, lib=None, latent_lib=None, sample_factor=True, dispersion="feature-cell"):
    """model.py:229-259 Output_Module_Raw.forward -> (scale, theta, rate, pi|None)."""
    hin = h if cov is None else torch.cat([h, cov], 1)
    a = torch.relu(linear(hin, sd["output_mean.0.weight"], sd["output_mean.0.bias"]))
    if "feature_scalar" in sd:
        scale = torch.softmax(a * torch.sigmoid(sd["feature_scalar"]), -1)
    else:
        scale = torch.softmax(a, -1)
    if sample_factor:
        rate = lib * scale
    elif latent_lib is not None:
        rate = torch.exp(leaky(latent_lib, 0.01)) * scale
    else:
        raise Exception("library size missing")
    if dispersion == "feature":
        theta = torch.exp(sd["output_inverse_dispersion"])
    else:
        theta = torch.exp(a)
    pi = None
    if "output_dropout.weight" in sd:
        pi = linear(a, sd["output_dropout.weight"], sd["output_dropout.bias"])
    return scale, theta, rate, pi


def output_gau(sd, h, cov=None, latent_lib=None):
    """model.py:282-300."""
    hin = h if cov is None else torch.cat([h, cov], 1)
    if latent_lib is not None:
        hin = torch.cat([hin, latent_lib], 1)
    out = linear(hin, sd["output_layer.weight"], sd["output_layer.bias"])
    if "feature_scalar" in sd:
        out = out * sd["feature_scalar"]
    return out


def module_module(sd, x, non_neg=False, layernorm=False, batchnorm=False, slope=0.01, training=False):
    """model.py:332-336."""
    if "translate.weight" in sd:
        y = linear(x, sd["translate.weight"], sd["translate.bias"])
    else:
        y = fcnn(sd, "translate.0", x, layernorm, batchnorm, slope, training)
        y = linear(y, sd["translate.1.weight"], sd["translate.1.bias"])
    return torch.clamp(y, min=0) if non_neg else y


def vae_encoder(sd, x, eps_noise, cov=None, slope=0.01):
    """pairedVAE.py:51-74 -> (mu, sigma, z, latent_lib|None); eps_noise replaces randn_like."""
    xin = x if cov is None else torch.cat([x, cov], 1)
    hdn = fcnn(sd, "enc_hidden", xin, True, False, slope)
    mu = linear(hdn, sd["latent_mu.weight"], sd["latent_mu.bias"])
    sigma = F.softplus(linear(hdn, sd["latent_sigma.weight"], sd["latent_sigma.bias"])) + 1e-7
    z = mu + eps_noise * sigma
    lat = None
    if any(k.startswith("enc_lib.") for k in sd):
        hl = fcnn(sd, "enc_lib.0", xin, True, False, slope)
        lat = F.softplus(linear(hl, sd["enc_lib.1.weight"], sd["enc_lib.1.bias"]))
    return mu, sigma, z, lat


def vae_decoder(sd, z, cov=None, lib=None, latent_lib=None, distribution="nb", dispersion="feature-cell",
                sample_factor=True, slope=0.01):
    """pairedVAE.py:124-187."""
    zin = z if cov is None else torch.cat([z, cov], 1)
    if latent_lib is not None:
        zin = torch.cat([zin, latent_lib], 1)
    d = fcnn(sd, "dec_hidden", zin, True, False, slope)
    out = linear(d, sd["output.weight"], sd["output.bias"])
    fs = sd.get("feature_scalar")
    if distribution == "gau":
        return out * fs if fs is not None else out
    if distribution == "ber":
        return torch.sigmoid(out * fs if fs is not None else out)
    scale = torch.softmax(out * torch.sigmoid(fs) if fs is not None else out, -1)
    if sample_factor:
        rate = lib * scale
    elif latent_lib is not None:
        rate = torch.exp(leaky(latent_lib, 0.01)) * scale
    else:
        rate = scale
    if dispersion == "feature":
        theta = torch.exp(sd["output_inverse_dispersion"])
    else:
        theta = F.softplus(linear(d, sd["output_inverse_dispersion.weight"], sd["output_inverse_dispersion.bias"])) + 1e-8
    pi = None
    if distribution == "zinb":
        pi = linear(d, sd["output_dropout.weight"], sd["output_dropout.bias"])
    return scale, theta, rate, pi


def kl_normal(mu, sigma):
    """utils.py:186 KL(N(mu,sigma)||N(0,1)).mean()."""
    return (-torch.log(sigma) + 0.5 * (sigma ** 2 + mu ** 2) - 0.5).mean()


# ----------------------------------------------------------------------------------------------
# optimiser  (utils.py:82-96 add_weight_decay; torch.optim.AdamW defaults, model.py:698-701)
# ----------------------------------------------------------------------------------------------
def weight_decay_split(names_shapes, output_layer):
    """names_shapes: iterable of (name, ndim).  utils.py:82-96 (note: `in` is list membership for the
    trainer's ['output'] and SUBSTRING test for the mapping nets' "bias")."""
    decay, nodecay = [], []
    for name, ndim in names_shapes:
        if ndim == 1 or name.split("_")[0] in output_layer:
            nodecay.append(name)
        else:
            decay.append(name)
    return decay, nodecay


class AdamW:
    """Decoupled-weight-decay Adam, torch defaults: betas (0.9,0.999), eps 1e-8, no amsgrad."""

    def __init__(self, params: dict, decay_names, lr=1e-3, wd=1e-9, betas=(0.9, 0.999), eps=1e-8):
        self.p = params
        self.decay = set(decay_names)
        self.lr, self.wd, self.b1, self.b2, self.eps = lr, wd, betas[0], betas[1], eps
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.t = {k: 0 for k in params}

    def step(self, grads: dict):
        for k, g in grads.items():
            if g is None:
                continue
            p = self.p[k]
            self.t[k] += 1
            t = self.t[k]
            wd = self.wd if k in self.decay else 0.0
            with torch.no_grad():
                p.mul_(1 - self.lr * wd)
                self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
                self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                bc1 = 1 - self.b1 ** t
                bc2 = 1 - self.b2 ** t
                denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
                p.addcdiv_(self.m[k], denom, value=-self.lr / bc1)


# ----------------------------------------------------------------------------------------------
# training step / loops  (utils.py:118-228, model.py:603-924)
# ----------------------------------------------------------------------------------------------
def _is_float_param(k, v):
    return v.dtype.is_floating_point and "running_" not in k


class Direction:
    """One encoder->decoder predictor (model.py:608-701) in functional form."""

    def __init__(self, enc_sd, dec_sd, likelihood, cfg):
        self.cfg = dict(layernorm=True, batchnorm=False, slope=0.01, sample_factor=True,
                        dispersion="feature-cell", lr=1e-3, wd=1e-9)
        self.cfg.update(cfg)
        self.lik = likelihood
        self.enc = {k: torch.as_tensor(v).clone() for k, v in enc_sd.items()}
        self.dec = {k: torch.as_tensor(v).clone() for k, v in dec_sd.items()}
        self.params = {}
        for tag, sd in (("enc", self.enc), ("dec", self.dec)):
            for k, v in sd.items():
                if _is_float_param(k, v):
                    v.requires_grad_(True)
                    self.params[f"{tag}:{k}"] = v
        e_dec, _ = weight_decay_split([(k, v.ndim) for k, v in self.enc.items() if _is_float_param(k, v)], ["output"])
        d_dec, _ = weight_decay_split([(k, v.ndim) for k, v in self.dec.items() if _is_float_param(k, v)], ["output"])
        decay = [f"enc:{k}" for k in e_dec] + [f"dec:{k}" for k in d_dec]
        self.opt = AdamW(self.params, decay, lr=self.cfg["lr"], wd=self.cfg["wd"])

    def forward_loss(self, x, y, lib, cov=None, training=False):
        c = self.cfg
        rep, lat = input_module(self.enc, x, cov, c["layernorm"], c["batchnorm"], c["slope"], training)
        if lib is not None:
            lat = None                                           # utils.py:121-124
        if self.lik in ("nb", "zinb"):
            scale, theta, rate, pi = output_raw(self.dec, rep, cov, lib, lat, c["sample_factor"], c["dispersion"])
            return raw_loss(y, rate, theta, pi, self.lik)
        if self.lik == "ber":
            return bce_masked(y, output_ber(self.dec, rep, cov, lib, lat, c["sample_factor"])[0])
        return mse_masked(y, output_gau(self.dec, rep, cov, lat))

    def train_step(self, x, y, lib, cov=None):
        loss = self.forward_loss(x, y, lib, cov, training=True)
        names = list(self.params)
        grads = torch.autograd.grad(loss, [self.params[k] for k in names], allow_unused=True)
        self.opt.step(dict(zip(names, grads)))
        return float(loss.detach())

    @torch.no_grad()
    def eval_loss(self, x, y, lib, cov=None):
        return float(self.forward_loss(x, y, lib, cov, training=False))

    @torch.no_grad()
    def embed(self, x, cov=None):
        c = self.cfg
        return input_module(self.enc, x, cov, c["layernorm"], c["batchnorm"], c["slope"], False)[0]

    def snapshot(self):
        return ({k: v.detach().clone() for k, v in self.enc.items()},
                {k: v.detach().clone() for k, v in self.dec.items()})


def run_direction(direction: Direction, xtr, ytr, libtr, xva, yva, libva, batches, n_epochs,
                  covtr=None, covva=None, patience=25, step_lr=None):
    """model.py:727-799 epoch loop with replayed batches (list of row-index arrays, in order).  step_lr = (step, gamma):
    the 'StepLR' scheduler of model.py:596-597, stepped once per epoch after validation while lr > 1e-7 (model.py:783-784)."""
    lr0 = direction.opt.lr
    per_epoch = len(batches) // n_epochs
    train_losses, val_losses = [], []
    best, best_sd, no_imp = float("inf"), None, 0
    it = iter(batches)
    for ep in range(n_epochs):
        for _ in range(per_epoch):
            rows = torch.as_tensor(next(it))
            train_losses.append(direction.train_step(
                xtr[rows], ytr[rows], None if libtr is None else libtr[rows],
                None if covtr is None else covtr[rows]))
        v = direction.eval_loss(xva, yva, libva, covva)
        val_losses.append(v)
        if step_lr is not None and direction.opt.lr > 1e-7:
            direction.opt.lr = lr0 * step_lr[1] ** ((ep + 1) // step_lr[0])
        if v < best:
            best, no_imp, best_sd = v, 0, direction.snapshot()
        else:
            no_imp += 1
        if ep > 1 and no_imp == patience:
            break
    return train_losses, val_losses, best_sd


# ---- evaluation metrics (evaluator.py:14-71) -- test infrastructure like the rest of this file -----------------------
def pairwise_sq_dists(x, y):
    """evaluator.py:14-21: D[i][j] = sum_k (x[i][k] - y[j][k])^2, direct form."""
    x = np.asarray(x); y = np.asarray(y)
    return np.sum(np.square(x[:, :, None] - y.T[None, :, :]), 1)


def foscttm(domain1, domain2):
    """evaluator.py:24-46 -> (average, per-cell 1->2 fractions, per-cell 2->1 fractions)."""
    n = np.asarray(domain1).shape[0]
    D = pairwise_sq_dists(domain1, domain2)
    f12 = np.array([np.sum(D[i, i] > D[i, :]) / (n - 1) for i in range(n)])
    f21 = np.array([np.sum(D[i, i] > D[:, i]) / (n - 1) for i in range(n)])
    return (f21.sum() / n + f12.sum() / n) / 2, f12, f21


def fosfttm(query, reference):
    """evaluator.py:49-71 -> (average, per-cell fractions)."""
    n = np.asarray(reference).shape[0]
    D = pairwise_sq_dists(query, reference)
    f = np.array([np.sum(D[i, i] < D[i, :]) / (n - 1) for i in range(n)])
    return f.sum() / n, f


def input_transform(x, mode, scale=1e4):
    """Opt-in encoder-input pre-transform of the build (no counterpart upstream, where model.py:492-510 feeds raw
    counts; parity unpinned, this is the defining formula): 'log1p' -> log1p(x); 'log1p_norm' ->
    log1p(x * scale / rowsum(x)), rows without counts left at zero."""
    x = torch.as_tensor(x, dtype=torch.float32)
    if mode is None:
        return x
    if mode == "log1p":
        return torch.log1p(x)
    if mode == "log1p_norm":
        lib = torch.nan_to_num(x, nan=0.0).sum(1, keepdim=True)
        return torch.log1p(torch.where(lib > 0, x * (scale / lib.clamp_min(1e-30)), x))
    raise ValueError(mode)
