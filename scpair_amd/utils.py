"""Builders and step functions with the reference's names and signatures (utils.py:40-228), running on
the HIP kernels.  Modules keep the reference's submodule layout (nn.Linear / nn.LayerNorm / ... inside
nn.Sequential) so parameter names, shapes, RNG consumption at construction and state_dicts are
interchangeable with the reference's; only forward() is replaced."""
from __future__ import annotations

import random

import numpy as np
import torch
from torch import nn
import torch.nn.functional as F

from . import ops
from .loss import binarized_loss, raw_loss, scaled_loss
from .sparse import BatchTiles, DeviceCSR


def set_seed(seed=None):
    """utils.py:40-50."""
    if seed is None:
        seed = np.random.choice(int(1e2))
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)


def _slope_of(act):
    if isinstance(act, nn.LeakyReLU):
        return float(act.negative_slope)
    if isinstance(act, nn.ReLU):
        return 0.0
    if isinstance(act, nn.Identity):
        return 1.0
    raise NotImplementedError(f"activation {type(act).__name__} is not supported by the HIP FCNN epilogue "
                              "(LeakyReLU, ReLU, Identity are)")


class HipFCNN(nn.Sequential):
    """nn.Sequential with the reference's layout (Linear -> [LayerNorm] -> act -> [BatchNorm1d] -> Dropout per layer,
    utils.py:54-64) whose forward runs one fused HIP layer per group.  The first layer accepts a BatchTiles
    (sparse minibatch) or a dense GPU tensor, plus an optional covariate block that the reference concatenates
    to the input (model.py:128)."""

    def _groups(self):
        mods = list(self)
        i, out = 0, []
        while i < len(mods):
            lin = mods[i]
            if not isinstance(lin, nn.Linear):
                raise RuntimeError("FCNN layout broken: expected nn.Linear")
            i += 1
            ln = bn = None
            if i < len(mods) and isinstance(mods[i], nn.LayerNorm):
                ln = mods[i]; i += 1
            act = mods[i]; i += 1
            if i < len(mods) and isinstance(mods[i], nn.BatchNorm1d):
                bn = mods[i]; i += 1
            drop = mods[i]; i += 1
            out.append((lin, ln, act, bn, drop))
        return out

    def forward(self, x, cov=None):
        for gi, (lin, ln, act, bn, drop) in enumerate(self._groups()):
            p = float(drop.p) if (self.training and drop.p > 0) else 0.0
            fused_p = 0.0 if bn is not None else p
            if gi == 0 and isinstance(x, ops.BatchTiles) and lin.weight.is_cuda and ops.get_precision() == "bf16":
                ops.ensure_homed(lin.weight)            # rows on 128-byte boundaries for the bf16 first-layer kernels
            x = ops.fcnn_layer(x, cov if gi == 0 else None, lin.weight, lin.bias,
                               None if ln is None else ln.weight, None if ln is None else ln.bias,
                               _slope_of(act), fused_p, ops.next_seed() if fused_p > 0 else 0,
                               1e-5 if ln is None else ln.eps)
            if bn is not None:
                # BatchNorm1d sits after the activation in the reference (utils.py:61-62): column statistics,
                # running-statistics update and the layer's dropout in one HIP kernel
                x = ops.batch_norm(x, bn, self.training, p, ops.next_seed() if p > 0 else 0)
        return x


def _branch_spec(fcnn, final, x, cov):
    """The tensors of an encoder branch the fused path (ops.enc_branches) takes, or None when the branch has another
    shape than the reference's default (sparse minibatch, optionally a covariate block, two FCNN layers with LayerNorm,
    both with or both without BatchNorm1d, a Linear with bias behind them or none).  BatchNorm encoders under data
    parallelism keep the layer-by-layer path: their global-batch statistics live in ops.SyncBatchNormFn."""
    groups = fcnn._groups() if isinstance(fcnn, HipFCNN) else None
    if groups is None or len(groups) != 2:
        return None
    (l1, n1, a1, bn1, d1), (l2, n2, a2, bn2, d2) = groups
    W3 = None if final is None else final.weight
    try:
        slope = _slope_of(a1)
        same_act = _slope_of(a2) == slope
    except NotImplementedError:
        return None
    if (bn1 is None) != (bn2 is None) or (bn1 is not None and (ops.bn_sync_active() or getattr(x, "shard", None) is not None)):
        return None
    if not (same_act and (final is None or final.bias is not None) and l2.bias is not None
            and ops.enc_branch_supported(x, cov, l1.weight, l2.weight, W3, n1 is not None, n2 is not None)
            and n1.elementwise_affine and n2.elementwise_affine):
        return None
    return dict(W1=l1.weight, b1=l1.bias, g1=n1.weight, be1=n1.bias, W2=l2.weight, b2=l2.bias, g2=n2.weight, be2=n2.bias,
                W3=W3, b3=None if final is None else final.bias, eps1=n1.eps, eps2=n2.eps, slope=slope,
                p1=float(d1.p) if (fcnn.training and d1.p > 0) else 0.0,
                p2=float(d2.p) if (fcnn.training and d2.p > 0) else 0.0,
                bn_mods=None if bn1 is None else (bn1, bn2, fcnn.training),
                bn1_w=None if bn1 is None else bn1.weight, bn1_b=None if bn1 is None else bn1.bias,
                bn2_w=None if bn2 is None else bn2.weight, bn2_b=None if bn2 is None else bn2.bias)


def encoder_branches(branches, x, cov=None):
    """[final(fcnn(x, cov)) for (fcnn, final) in branches] for the branches of one encoder (model.py:124-143: hidden_rep
    [+ extra_linear_rep], hidden_lib + its head).  Branches of the reference's default shape run together: their
    first-layer products as one launch over the packed weights (precision 'bf16') and their tails as ONE fused kernel
    (ops.enc_branches); everything else goes layer by layer."""
    if isinstance(x, ops.BatchTiles) and ops.get_precision() == "bf16":
        # the branches' first-layer weights become adjacent, 128-byte aligned row blocks of one tensor (idempotent)
        firsts = [f[0].weight for f, _ in branches if isinstance(f, HipFCNN) and len(f) and isinstance(f[0], nn.Linear)]
        if len(firsts) == len(branches) and all(w.is_cuda for w in firsts):
            ops.home_first_layers(*firsts[:ops.ENC_MAX_BRANCHES])
    specs = [_branch_spec(f, fin, x, cov) for f, fin in branches]
    outs = [None] * len(branches)
    fusable = [i for i, sp in enumerate(specs) if sp is not None]
    by_slope = {}
    for i in fusable:
        # one call = one slope and one BatchNorm mode (none / batch statistics / running statistics)
        bm = specs[i]["bn_mods"]
        mode = None if bm is None else bool(bm[2] or bm[0].running_mean is None or bm[1].running_mean is None)
        by_slope.setdefault((specs[i]["slope"], mode), []).append(i)
    for (slope, _mode), idx in by_slope.items():
        idx = idx[:ops.ENC_MAX_BRANCHES]
        for i in idx:
            bm = specs[i]["bn_mods"]
            specs[i]["bn"] = None if bm is None else ops.bn_spec(*bm)      # nn.BatchNorm1d's bookkeeping: once per forward
        res = ops.enc_branches(x, [specs[i] for i in idx], slope, cov=cov)
        for i, r in zip(idx, res):
            outs[i] = r
    if getattr(x, "shard", None) is not None and any(o is None for o in outs):
        # the collectives of the feature-sharded first layer live in the fused path only: fail loudly rather than run a
        # branch on this rank's column block as if it were the whole input
        raise RuntimeError("feature-sharded first layer: the encoder must have the fused shape (sparse minibatch, no "
                           "covariate block, two FCNN layers with LayerNorm and no BatchNorm, widths H1 % 4 == 0, H1 <= 1024, "
                           "H2 <= 64)")
    for i, (f, fin) in enumerate(branches):
        if outs[i] is None:
            h = f(x, cov)
            outs[i] = h if fin is None else fin(h)
    return outs


def encoder_branch(fcnn, final, x, cov=None):
    return encoder_branches([(fcnn, final)], x, cov)[0]


def FCNN(layers, layernorm=True, activation=nn.ReLU(), batchnorm=False, dropout_rate=0):
    """utils.py:54-64 (same module order, so same parameter names and the same RNG draws)."""
    fc_nn = []
    for i in range(1, len(layers)):
        fc_nn.append(nn.Linear(layers[i - 1], layers[i]))
        if layernorm:
            fc_nn.append(nn.LayerNorm(layers[i]))
        fc_nn.append(activation)
        if batchnorm:
            fc_nn.append(nn.BatchNorm1d(layers[i]))
        fc_nn.append(nn.Dropout(dropout_rate))
    return HipFCNN(*fc_nn)


class HipLinear(nn.Linear):
    """nn.Linear whose forward is the MFMA GEMM (same parameters / init as nn.Linear)."""

    def forward(self, x, relu_input=False):
        """relu_input: y = Linear(relu(x)) with the relu folded into the layer (its cast pass forward, a mask in the
        input-gradient epilogue backward)."""
        return ops.linear(x, self.weight, self.bias, relu_input)


def init_weights(nn_modules, method="xavier_normal"):
    """utils.py:68-78 (note the reference tests `type(m) == nn.Linear`; HipLinear must be initialised too)."""
    if isinstance(nn_modules, nn.Linear):
        if method == "xavier_normal":
            torch.nn.init.xavier_normal_(nn_modules.weight)
        elif method == "kaiming_normal":
            torch.nn.init.kaiming_normal_(nn_modules.weight, mode="fan_in", nonlinearity="relu")
        elif method == "xavier_uniform":
            torch.nn.init.xavier_uniform_(nn_modules.weight)
        elif method == "kaiming_uniform":
            torch.nn.init.kaiming_uniform_(nn_modules.weight, mode="fan_in", nonlinearity="relu")
        nn_modules.bias.data.fill_(0.0)


def add_weight_decay(model, output_layer=()):
    """utils.py:82-96."""
    decay_param, nodecay_param, decay_name, nodecay_name = [], [], [], []
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        if len(param.shape) == 1 or name.split("_")[0] in output_layer:
            nodecay_name.append(name)
            nodecay_param.append(param)
        else:
            decay_name.append(name)
            decay_param.append(param)
    return decay_param, nodecay_param, decay_name, nodecay_name


def training_mode(model1, model2=None):
    model1.train()
    if model2 is not None:
        model2.train()


def evaluating_mode(model1, model2=None):
    model1.eval()
    if model2 is not None:
        model2.eval()


# ------------------------------------------------------------------------------------------------------
def _as_labels(output, device):
    """Labels for the fused decoder kernels: BatchTiles as is; dense tensors are converted (host round trip)."""
    if isinstance(output, BatchTiles):
        return output
    csr = DeviceCSR.from_dense(output, device)
    return BatchTiles(csr, np.arange(csr.n_rows))


def _predictor_loss(input, output, library_size, encoder, decoder, likelihood_type, add_cov, input_batch, output_batch):
    """Shared body of train_net / eval_net (utils.py:119-139, 153-173)."""
    latent_rep, latent_lib = encoder(input, input_batch if add_cov else None)
    ob = output_batch if add_cov else None
    if library_size is not None:
        lib, latent_lib = library_size, None                      # utils.py:121-124
    else:
        lib = None
    if likelihood_type in ("zinb", "nb", "ber") and hasattr(decoder, "fused_loss"):
        fused = decoder.fused_loss(latent_rep, ob, lib, latent_lib, _as_labels(output, latent_rep.device),
                                   likelihood_type)
        if fused is not None:
            return fused
    outputs = decoder(latent_rep, ob, lib, latent_lib)
    y = output.to_dense() if isinstance(output, BatchTiles) else output
    if likelihood_type in ("zinb", "nb"):
        px_mu_scale, px_theta, px_mu_rate, px_pi = outputs
        return raw_loss(x=y, px_mu_rate=px_mu_rate, px_theta=px_theta, px_mu_scale=px_mu_scale, px_pi=px_pi,
                        likelihood=likelihood_type)
    if likelihood_type == "ber":
        return binarized_loss(y, outputs[0])
    if likelihood_type == "gau":
        return scaled_loss(y, outputs)
    raise ValueError("likelihood_type must be one of zinb, nb, ber, gau")


def train_net(input, output, library_size, encoder, decoder, optimizer, likelihood_type="zinb", add_cov=False,
              input_batch=None, output_batch=None, scaler=None):
    """utils.py:118-148.  `input` / `output` may be BatchTiles (sparse minibatches resident in HBM) or dense GPU
    tensors.  `scaler` (fp16 GradScaler in the reference) is accepted and ignored: bf16 MFMA inputs with fp32
    accumulation and fp32 master weights need no loss scaling."""
    training_loss = _predictor_loss(input, output, library_size, encoder, decoder, likelihood_type, add_cov,
                                    input_batch, output_batch)
    optimizer.zero_grad()
    ops.backward_root(training_loss)
    if isinstance(optimizer, ops.FusedAdamW):
        optimizer.step(release_grads=True)      # the next zero_grad() (40 parameters) becomes free
    else:
        optimizer.step()
    return training_loss.detach()


def eval_net(input, output, library_size, encoder, decoder, likelihood_type="zinb", add_cov=False, input_batch=None,
             output_batch=None):
    """utils.py:152-174."""
    return _predictor_loss(input, output, library_size, encoder, decoder, likelihood_type, add_cov, input_batch,
                           output_batch)


def _vae_losses(input, lib, encoder, decoder, kl_weight, likelihood_type, add_cov, input_batch):
    cov = input_batch if add_cov else None
    mu, sigma, z, latent_lib = encoder(input, cov)
    kl_divergence = encoder.last_kl
    if likelihood_type in ("zinb", "nb") and hasattr(decoder, "fused_loss"):
        recon = decoder.fused_loss(z, cov, lib, latent_lib, input, likelihood_type)
        if recon is not None:
            return kl_weight * kl_divergence + recon, recon, kl_divergence
    outputs = decoder(z, cov, lib, latent_lib)
    x = input.to_dense() if isinstance(input, BatchTiles) else input
    if likelihood_type in ("zinb", "nb"):
        px_mu_scale, px_theta, px_mu_rate, px_pi = outputs
        recon = raw_loss(x=x, px_mu_rate=px_mu_rate, px_theta=px_theta, px_mu_scale=px_mu_scale, px_pi=px_pi,
                         likelihood=likelihood_type)
    elif likelihood_type == "ber":
        # the reference indexes outputs[0] of a bare tensor here (utils.py:191) and fails; the evident intent is used
        recon = binarized_loss(x, outputs)
    elif likelihood_type == "gau":
        recon = scaled_loss(x, outputs)
    else:
        raise ValueError("likelihood_type must be one of zinb, nb, ber, gau")
    return kl_weight * kl_divergence + recon, recon, kl_divergence


def train_VAE(input, lib, encoder, decoder, optimizer, kl_weight=1, likelihood_type="zinb", add_cov=False,
              input_batch=None):
    """utils.py:179-202."""
    total_loss, _, _ = _vae_losses(input, lib, encoder, decoder, kl_weight, likelihood_type, add_cov, input_batch)
    optimizer.zero_grad()
    total_loss.backward()
    optimizer.step()
    return total_loss.detach()


def eval_VAE(input, lib, encoder, decoder, kl_weight=1, likelihood_type="zinb", add_cov=False, input_batch=None):
    """utils.py:206-228."""
    with torch.no_grad():
        return _vae_losses(input, lib, encoder, decoder, kl_weight, likelihood_type, add_cov, input_batch)
