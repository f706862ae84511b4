"""scPair modules and trainer with the reference's names, constructor arguments, parameter names and return
conventions (model.py:60-1036), running on the MI355X HIP kernels.

What changes underneath:
  * count matrices stay in HBM as CSR (sparse.DeviceCSR); a minibatch is a row-id vector re-bucketed on the GPU
    into a tile list (sparse.BatchTiles) -- no dense [cells x features] tensor is ever built for training;
  * the first encoder layer densifies those tiles inside LDS and runs on bf16 MFMA; LayerNorm/activation/dropout
    are one fused epilogue; decoders are fused with their likelihood and their backward;
  * AdamW is one fused multi-tensor launch.
"""
from __future__ import annotations

import copy
import itertools
import gc
import os

import numpy as np
import torch
from torch import nn
import torch.nn.functional as F

from . import ops
from .loss import binarized_loss, raw_loss, scaled_loss
from .sparse import BatchTiles, DeviceCSR, rows_to_device, scalar64
from .utils import (FCNN, HipLinear, add_weight_decay, encoder_branches, eval_net, evaluating_mode, init_weights,
                    set_seed, train_net, training_mode)


def clear_gpu_memory():
    """model.py:42-46 (kept for API compatibility; the HIP path allocates per step through torch's caching allocator)."""
    gc.collect()


def move_to_device_safe(tensor, device):
    """model.py:49-55."""
    if tensor is None:
        return None
    if isinstance(tensor, (BatchTiles, DeviceCSR)):
        return tensor
    if tensor.device == device:
        return tensor
    return tensor.to(device, non_blocking=True)


class LazyGPUDataset(torch.utils.data.Dataset):
    """model.py:26-38 (API compatibility only; training uses device-resident CSR instead)."""

    def __init__(self, *tensors, device="cuda"):
        self.tensors = tensors
        self.device = device
        self.length = len(tensors[0]) if tensors else 0

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        return tuple(tensor[idx] for tensor in self.tensors)


def _cat_cov(h, cov):
    return h if cov is None else torch.cat([h, cov], dim=1)


# ---------------------------------------------------------------------------------------------------------
class Input_Module(nn.Module):
    """Encoder (model.py:60-143): latent_rep = [extra_linear_rep](hidden_rep(cat[x, cov])),
    latent_lib = Linear(hidden_lib_FCNN(cat[x, cov])) if infer_library_size else None."""

    def __init__(self, input_dim: int, input_batch_num: int, hidden_layer, layernorm=True, activation=nn.ReLU(),
                 batchnorm=False, dropout_rate=0, add_linear_layer=False, infer_library_size=True,
                 clip_threshold=None):
        super().__init__()
        self.infer_library_size = infer_library_size
        self.add_linear_layer = add_linear_layer
        self.clip_threshold = clip_threshold
        self.hidden_rep = FCNN([input_dim + input_batch_num] + hidden_layer, layernorm=layernorm,
                               activation=activation, batchnorm=batchnorm, dropout_rate=dropout_rate)
        if self.add_linear_layer is True:
            self.extra_linear_rep = HipLinear(hidden_layer[-1], hidden_layer[-1], bias=True)
        if self.infer_library_size == True:  # noqa: E712  (reference spelling)
            self.hidden_lib = nn.Sequential(FCNN([input_dim + input_batch_num] + hidden_layer, activation=activation,
                                                 layernorm=layernorm, batchnorm=batchnorm,
                                                 dropout_rate=dropout_rate),
                                            HipLinear(hidden_layer[-1], 1, bias=True))

    def forward(self, input, input_batch):
        branches = [(self.hidden_rep, self.extra_linear_rep if self.add_linear_layer is True else None)]
        if self.infer_library_size == True:  # noqa: E712
            branches.append((self.hidden_lib[0], self.hidden_lib[1]))
        outs = encoder_branches(branches, input, input_batch)
        return outs[0], (outs[1] if len(outs) > 1 else None)


class Output_Module_Ber(nn.Module):
    """Bernoulli decoder (model.py:149-190)."""

    def __init__(self, output_dim: int, output_batch_num: int, hidden_layer, infer_library_size=True,
                 sample_factor=True, feature_factor=True):
        super().__init__()
        self.infer_library_size = infer_library_size
        self.sample_factor = sample_factor
        self.feature_factor = feature_factor
        if self.feature_factor == True:  # noqa: E712
            self.feature_scalar = torch.nn.Parameter(torch.zeros(output_dim))
        self.output_layer = HipLinear(hidden_layer[-1] + output_batch_num, output_dim, bias=True)

    def _lrow(self, lib, latent_lib):
        if self.sample_factor == True:  # noqa: E712
            return lib
        if latent_lib is not None:
            return latent_lib
        return None

    def forward(self, latent_rep, output_batch, lib, latent_lib):
        """Dense outputs (output_result, output_px, output_sample_factor, output_feature_factor) as in the
        reference; used by predict().  Training uses fused_loss(), which never materialises them."""
        latent_rep = _cat_cov(latent_rep, output_batch)
        self.output_temp = self.output_layer(latent_rep)
        lrow = self._lrow(lib, latent_lib)
        if not torch.is_grad_enabled() and self.output_temp.is_cuda:
            # inference: the fused head kernel (one pass per returned tensor, no intermediate products)
            fs = self.feature_scalar if self.feature_factor == True else None  # noqa: E712
            output_px = ops.ber_prob(self.output_temp)
            output_result = output_px if (lrow is None and fs is None) else ops.ber_prob(self.output_temp, lrow, fs)
            return (output_result, output_px, None if lrow is None else torch.sigmoid(lrow),
                    None if fs is None else torch.sigmoid(fs))
        output_px = torch.sigmoid(self.output_temp)
        if lrow is not None:
            output_sample_factor = torch.sigmoid(lrow)
            output_result = output_px * output_sample_factor
        else:
            output_sample_factor = None
            output_result = output_px
        if self.feature_factor == True:  # noqa: E712
            output_feature_factor = torch.sigmoid(self.feature_scalar)
            output_result = output_result * output_feature_factor
        else:
            output_feature_factor = None
        return output_result, output_px, output_sample_factor, output_feature_factor

    @torch.no_grad()
    def predict_head(self, latent_rep, output_batch, lib, latent_lib, out_dtype=torch.float32):
        """forward(...)[0] only (what predict() keeps, model.py:584-590): logits GEMM + one head kernel, optionally
        written as bf16."""
        z = self.output_layer(_cat_cov(latent_rep, output_batch))
        fs = self.feature_scalar if self.feature_factor == True else None  # noqa: E712
        return ops.ber_prob(z, self._lrow(lib, latent_lib), fs, out_dtype=out_dtype)

    def fused_loss(self, latent_rep, output_batch, lib, latent_lib, labels, likelihood="ber"):
        """binarized_loss(labels, forward(...)[0]) in one kernel pass with its backward (no [cells x peaks] tensor)."""
        h = _cat_cov(latent_rep, output_batch)
        if not ops.decoder_depth_supported(h.shape[1]):
            return None                                # many covariate levels: materialised forward + dense BCE kernel
        fs = self.feature_scalar if self.feature_factor == True else None  # noqa: E712
        return ops.ber_decoder_loss(h, self.output_layer.weight, self.output_layer.bias, fs,
                                    self._lrow(lib, latent_lib), labels)


class Output_Module_Raw(nn.Module):
    """(ZI)NB decoder (model.py:196-259).  Keeps the reference's quirks: for dispersion='feature-cell' an unused
    Linear(G, G) `output_inverse_dispersion` is allocated (state_dict compatibility), and `zero_inflated` is
    independent of the 'nb'/'zinb' likelihood string."""

    def __init__(self, output_dim: int, output_batch_num: int, hidden_layer, infer_library_size=True,
                 sample_factor=True, feature_factor=False, zero_inflated=True, dispersion="feature-cell", eps=1e-8):
        super().__init__()
        self.infer_library_size = infer_library_size
        self.sample_factor = sample_factor
        self.feature_factor = feature_factor
        self.zero_inflated = zero_inflated
        self.dispersion = dispersion
        self.epsilon = eps
        if self.feature_factor == True:  # noqa: E712
            self.feature_scalar = torch.nn.Parameter(torch.zeros(output_dim))
        self.output_mean = nn.Sequential(HipLinear(hidden_layer[-1] + output_batch_num, output_dim, bias=True),
                                         nn.ReLU())
        self.transformation_function = nn.LeakyReLU()
        self.scale_function = nn.Softmax(dim=-1)
        if self.dispersion == "feature":
            self.output_inverse_dispersion = torch.nn.Parameter(torch.randn(output_dim))
        elif self.dispersion == "feature-cell":
            self.output_inverse_dispersion = HipLinear(output_dim, output_dim, bias=True)
        if self.zero_inflated == True:  # noqa: E712
            self.output_dropout = HipLinear(output_dim, output_dim, bias=True)

    def forward(self, latent_rep, output_batch, lib, latent_lib):
        latent_rep = _cat_cov(latent_rep, output_batch)
        hip_head = (not torch.is_grad_enabled()) and latent_rep.is_cuda
        theta_k = None
        if hip_head:
            # inference: relu -> [feature factor] -> row softmax (and exp dispersion) in one row kernel
            z = self.output_mean[0](latent_rep)
            px_mu_scale, theta_k = ops.relu_softmax_rows(
                z, self.feature_scalar if self.feature_factor == True else None, True,  # noqa: E712
                want_theta=(self.dispersion == "feature-cell"))
            px_mu = torch.relu_(z) if self.zero_inflated == True else None  # noqa: E712   (operand of the pi layer)
        else:
            px_mu = self.output_mean(latent_rep)
            if self.feature_factor == True:  # noqa: E712
                px_mu_scale = self.scale_function(px_mu * torch.sigmoid(self.feature_scalar))
            else:
                px_mu_scale = self.scale_function(px_mu)
        if self.sample_factor == True:  # noqa: E712
            px_mu_rate = lib * px_mu_scale
        elif latent_lib is not None:
            px_mu_rate = torch.exp(self.transformation_function(latent_lib)) * px_mu_scale
        else:
            raise Exception("The library size is missing. Please ensure that either the 'sample_factor' or "
                            "'infer_library_size' parameter for the raw counts decoder is set to True.")
        if self.dispersion == "feature":
            px_theta = torch.exp(self.output_inverse_dispersion)
        elif self.dispersion == "feature-cell":
            px_theta = theta_k if hip_head else torch.exp(px_mu)
        if self.zero_inflated == True:  # noqa: E712
            px_pi = self.output_dropout(px_mu)
        else:
            px_pi = None
        return px_mu_scale, px_theta, px_mu_rate, px_pi

    @torch.no_grad()
    def predict_head(self, latent_rep, output_batch, lib, latent_lib, out_dtype=torch.float32):
        """forward(...)[0] only, the normalised mean `px_mu_scale` that predict() keeps (model.py:584-590): logits
        GEMM + one row kernel, optionally written as bf16."""
        z = self.output_mean[0](_cat_cov(latent_rep, output_batch))
        fs = self.feature_scalar if self.feature_factor == True else None  # noqa: E712
        return ops.relu_softmax_rows(z, fs, True, out_dtype=out_dtype)[0]

    def fused_loss(self, latent_rep, output_batch, lib, latent_lib, labels, likelihood="nb"):
        """raw_loss(labels, *forward(...)): likelihood 'nb' in three fused decoder passes (row softmax statistics,
        likelihood, gradients); 'zinb' through the materialised pre-activations and the fused row kernels (feature factor
        and dispersion 'feature' included).  Returns None for the combinations without a fused form."""
        if likelihood not in ("nb", "zinb"):
            return None
        h = _cat_cov(latent_rep, output_batch)
        fs = self.feature_scalar if self.feature_factor == True else None  # noqa: E712
        thp = self.output_inverse_dispersion if self.dispersion == "feature" else None
        lin = self.output_mean[0]
        if likelihood == "zinb":
            # The G x G zero-inflation layer needs px_mu as a GEMM operand, so the pre-activations are materialised:
            # z = Linear(h) (MFMA GEMM), px_mu = relu(z), pi = Linear_GxG(px_mu) (hand-written wide MFMA GEMM in bf16 mode), then
            # softmax / exp / ZINB likelihood and their backward in the fused row kernels.
            if self.zero_inflated != True:  # noqa: E712
                return None
            if self.sample_factor == True:  # noqa: E712
                L = lib
            elif latent_lib is not None:
                L = torch.exp(self.transformation_function(latent_lib))
            else:
                raise Exception("The library size is missing. Please ensure that either the 'sample_factor' or "
                                "'infer_library_size' parameter for the raw counts decoder is set to True.")
            z = ops.linear(h, lin.weight, lin.bias)
            pi = self.output_dropout(z, relu_input=True)          # Linear(relu(z)), model.py:240,255-256
            return ops.raw_rows_loss(z, pi, L, labels.to_dense(), fs=fs, thp=thp)
        if self.sample_factor == True:  # noqa: E712
            L, lat = lib, None
        elif latent_lib is not None:
            lat = latent_lib
            L = torch.exp(F.leaky_relu(latent_lib.detach(), 0.01))
        else:
            raise Exception("The library size is missing. Please ensure that either the 'sample_factor' or "
                            "'infer_library_size' parameter for the raw counts decoder is set to True.")
        if not ops.decoder_depth_supported(h.shape[1]):
            return None                                # many covariate levels: materialised forward + dense NB kernels
        return ops.nb_decoder_loss(h, lin.weight, lin.bias, fs, thp, L, lat, labels)


class Output_Module_Gau(nn.Module):
    """Gaussian decoder (model.py:268-300)."""

    def __init__(self, output_dim: int, output_batch_num: int, hidden_layer, infer_library_size=True,
                 feature_factor=False):
        super().__init__()
        self.infer_library_size = infer_library_size
        self.feature_factor = feature_factor
        if self.feature_factor == True:  # noqa: E712
            self.feature_scalar = torch.nn.Parameter(torch.zeros(output_dim))
        self.output_layer = HipLinear(hidden_layer[-1] + output_batch_num + self.infer_library_size, output_dim,
                                      bias=True)

    def forward(self, latent_rep, output_batch, lib, latent_lib):
        latent_rep = _cat_cov(latent_rep, output_batch)
        if latent_lib is not None:
            latent_rep = torch.cat([latent_rep, latent_lib], dim=1)
        self.output_temp = self.output_layer(latent_rep)
        if self.feature_factor == True:  # noqa: E712
            return self.output_temp * self.feature_scalar
        return self.output_temp


class Module_Module(nn.Module):
    """Latent translation network (model.py:305-336)."""

    def __init__(self, origin_module_dim: int, target_module_dim: int, translational_hidden_nodes, non_neg=True,
                 layernorm=False, activation=nn.ReLU(), batchnorm=False, dropout_rate=0):
        super().__init__()
        self.nonneg = non_neg
        self.translational_hidden_nodes = translational_hidden_nodes
        if self.translational_hidden_nodes is not None:
            self.translate = nn.Sequential(FCNN([origin_module_dim] + translational_hidden_nodes,
                                                activation=activation, batchnorm=batchnorm, layernorm=layernorm,
                                                dropout_rate=dropout_rate),
                                           HipLinear(translational_hidden_nodes[-1], target_module_dim, bias=True))
        else:
            self.translate = HipLinear(origin_module_dim, target_module_dim, bias=True)

    def forward(self, x):
        translation = self.translate(x)
        if self.nonneg == True:  # noqa: E712
            translation = torch.clamp(translation, min=0)
        return translation


# ---------------------------------------------------------------------------------------------------------
class _Index:
    def __init__(self, n):
        self.index = np.arange(n)


def _to_csr(v, device, binary=None):
    """dict-seam value -> DeviceCSR.  Accepts the reference's dense FloatTensor, scipy sparse, or DeviceCSR.  binary=None:
    a matrix whose stored entries are all 1 (binarised peaks) is held without a value array, which puts it on the kernels'
    implicit-ones path (same numbers: the value 1 is exact in every operand format)."""
    if isinstance(v, DeviceCSR):
        return v
    if isinstance(v, torch.Tensor) or isinstance(v, np.ndarray):
        return DeviceCSR.from_dense(v, device, binary=binary)
    return DeviceCSR.from_scipy(v, device, binary=binary)


def _snapshot(src, dst=None):
    """The best-so-far copy of a module that the epoch loops keep (the reference deep-copies the module every time the
    validation loss improves, model.py:790-792): the first call makes the deep copy, later calls overwrite its parameters
    and buffers in place -- the same values without rebuilding the module tree and re-allocating its tensors every epoch."""
    if dst is None or os.environ.get("SCPAIR_DEEPCOPY_CHECKPOINTS"):
        return copy.deepcopy(src)
    with torch.no_grad():
        to = list(dst.parameters()) + list(dst.buffers())
        frm = list(src.parameters()) + list(src.buffers())
        if len(to) != len(frm) or any(a.shape != b.shape or a.dtype != b.dtype for a, b in zip(to, frm)):
            return copy.deepcopy(src)
        if to:
            torch._foreach_copy_(to, frm)
    dst.train(src.training)
    return dst


class _FastIndexLoader:
    """What DataLoader(TensorDataset(arange(n)), batch_size, shuffle=True, num_workers=0) yields, without its per-row Python
    work: the same two draws from the global torch RNG per epoch -- the iterator's base seed when it is created
    (torch/utils/data/dataloader.py, _BaseDataLoaderIter.__init__) and the RandomSampler's private seed when the first batch
    is asked for (sampler.py, RandomSampler.__iter__) -- then slices of randperm(n) from that private generator.  It follows
    torch's internals, so _index_loader() checks it against the real DataLoader once per process and uses that instead if
    they ever disagree."""

    def __init__(self, n, batch_size, drop_last=False):
        self.n, self.batch_size, self.drop_last = int(n), int(batch_size), bool(drop_last)

    def __len__(self):
        return self.n // self.batch_size if self.drop_last else -(-self.n // self.batch_size)

    def __iter__(self):
        torch.empty((), dtype=torch.int64).random_()              # the iterator's base seed (unused without workers)
        return self._batches()

    def _batches(self):
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        g = torch.Generator()
        g.manual_seed(seed)
        perm = torch.randperm(self.n, generator=g)
        for i in range(0, self.n, self.batch_size):
            idx = perm[i:i + self.batch_size]
            if self.drop_last and idx.shape[0] < self.batch_size:
                break
            yield (idx,)


def _dataloader_over_indices(n, batch_size, drop_last=False):
    """The reference's loader (model.py:713-717: shuffle=True, num_workers=0 -> BatchSampler(RandomSampler(dataset), batch_size,
    drop_last)) over row ids, with the batch sampler handed over as `sampler` and automatic batching off: the dataset is
    indexed once per BATCH instead of once per row and nothing is collated (1024 `__getitem__` calls and a stack per batch
    cost 2.4 ms of host time, more than the training step itself).  Same draws from the global RNG, same batches."""
    ds = torch.utils.data.TensorDataset(torch.arange(n))
    batches = torch.utils.data.BatchSampler(torch.utils.data.RandomSampler(ds), batch_size, drop_last)
    return torch.utils.data.DataLoader(ds, batch_size=None, sampler=batches, num_workers=0)


_fast_loader_ok = None


def _fast_loader_matches_dataloader():
    """One-time check (forked RNG, a 23-row dataset, two epochs, both drop_last settings): _FastIndexLoader yields the
    DataLoader's batches and leaves the global RNG in the same state."""
    global _fast_loader_ok
    if _fast_loader_ok is None:
        ok = True
        try:
            with torch.random.fork_rng(devices=[]):
                for drop_last in (False, True):
                    out = []
                    for make in (_dataloader_over_indices, _FastIndexLoader):
                        torch.manual_seed(20240229)
                        ld = make(23, 5, drop_last)
                        got = [[b[0].clone() for b in ld] for _ in range(2)]
                        out.append((got, torch.get_rng_state()))
                    (a, sa), (b, sb) = out
                    ok = ok and bool((sa == sb).all()) and all(
                        len(x) == len(y) and all(p.dtype == q.dtype and torch.equal(p, q) for p, q in zip(x, y))
                        for x, y in zip(a, b))
        except Exception:
            ok = False
        _fast_loader_ok = ok
    return _fast_loader_ok


def _index_loader(n, batch_size, drop_last=False):
    """Shuffled minibatches of row ids, drawn from the global torch RNG the way the reference's DataLoader draws them
    (model.py:713-717), so batch composition and order are the reference's for the same seed."""
    if _fast_loader_matches_dataloader() and not os.environ.get("SCPAIR_TORCH_DATALOADER"):
        return _FastIndexLoader(n, batch_size, drop_last)
    return _dataloader_over_indices(n, batch_size, drop_last)


class _GraphStepper:
    """Full-size minibatch steps of one predictor (model.py:729-763) through a HIP graph: the first steps run eagerly
    (optimiser state, weight shadows and kernel attributes come into being), the next one is recorded, the rest are
    replays.  Ragged batches (the last one of an epoch, drop_last=False) stay eager."""
    WARM = 2

    def __init__(self, owner, xtr, ytr, ctr, use_lib, encoder, decoder, optimizer, dist):
        self.o, self.xtr, self.ytr, self.ctr, self.use_lib = owner, xtr, ytr, ctr, use_lib
        self.enc, self.dec, self.opt, self.dist = encoder, decoder, optimizer, dist
        self.B = int(owner.batch_size)
        self.seen = 0
        self.cs = None
        self.ok = all(m.momentum is not None for mod in (encoder, decoder) for m in mod.modules()
                      if isinstance(m, nn.BatchNorm1d))       # cumulative-average BN keeps a host-side counter

    def wants(self, rows):
        if not self.ok or len(rows) != self.B:
            return False
        if self.cs is not None:
            return True
        self.seen += 1
        return self.seen > self.WARM

    def _fn(self, rows_dev):
        tx = BatchTiles(self.xtr, rows_dev, want_lib=False, capacity=self.cap_x)
        ty = BatchTiles(self.ytr, rows_dev, want_lib=self.use_lib, capacity=self.cap_y)
        c = None if self.ctr is None else self.ctr[rows_dev]
        lib = ty.lib.reshape(-1, 1) if self.use_lib else None
        return train_net(tx, ty, lib, self.enc, self.dec, self.opt, likelihood_type=self.dist, add_cov=c is not None,
                         input_batch=c, output_batch=c)

    def step(self, rows, rows_dev):
        if rows.min() < 0 or rows.max() >= self.xtr.n_rows:
            raise IndexError("minibatch row id out of range")
        if len(np.unique(rows)) != len(rows):
            # the fixed tile-list capacity (max_batch_nnz) bounds minibatches of DISTINCT rows only
            raise ValueError("a captured step needs distinct row ids in a minibatch")
        if self.cs is None:
            from .graph import CapturedStep
            self.cap_x, self.cap_y = self.xtr.max_batch_nnz(self.B), self.ytr.max_batch_nnz(self.B)
            self.cs = CapturedStep(self._fn, [self.opt], self.B, self.o.device).capture(rows_dev)
        return self.cs(rows_dev).clone()

    def release(self):
        if self.cs is not None:
            self.cs.release()
            self.cs = None


class scPair_object():
    """model.py:341-1036.  Same constructor surface; `device` must be a GPU."""

    def __init__(self,
                 scobj,
                 modalities={'Gene Expression': 'zinb', 'Peaks': 'ber'},
                 cov=['labels'],
                 SEED=0,
                 hidden_layer=[900, 40],
                 dropout_rate=0.1,
                 batchnorm=False,
                 layernorm=True,
                 learning_rate_prediction=1e-3,
                 batch_size=130,
                 L2_lambda=1e-9,
                 max_epochs=1000,
                 activation=nn.LeakyReLU(),
                 early_stopping_activation=True,
                 early_stopping_patience=25,
                 weight_decay=None,
                 device=None,
                 save_model=True,
                 save_path=None,
                 input_module=Input_Module,
                 output_module={'nb': Output_Module_Raw, 'zinb': Output_Module_Raw, 'ber': Output_Module_Ber,
                                'gau': Output_Module_Gau},
                 add_linear_layer=True,
                 sample_factor_rna=True,
                 feature_factor_rna=False,
                 sample_factor_atac=False,
                 feature_factor_atac=False,
                 zero_inflated=True,
                 dispersion="feature-cell",
                 infer_library_size_rna=False,
                 infer_library_size_atac=True,
                 mapping_module=Module_Module,
                 mapping_hidden_nodes=None,
                 mapping_learning_rate=1e-3,
                 mapping_non_neg=False,
                 mapping_layernorm=False,
                 mapping_batchnorm=False,
                 mapping_activation=nn.LeakyReLU(),
                 mapping_dropout_rate=0,
                 use_mixed_precision=False,
                 gradient_checkpointing=True,
                 pin_memory=True,
                 memory_efficient_loading=True,
                 val_chunk=4096,
                 verbose=True,
                 grad_sync=None, fuse_optimizer=True, input_transform=None, capture_step=False,
                 shard_first_layer=False, concurrent_directions=True):
        self.scobj = scobj
        self.modalities = modalities
        self.modality_names = list(modalities.keys())
        self.modality_distributions = list(modalities.values())
        self.cov = cov
        self.SEED = SEED
        self.hidden_layer = hidden_layer
        self.dropout_rate = dropout_rate
        self.batchnorm = batchnorm
        self.layernorm = layernorm
        self.learning_rate_prediction = learning_rate_prediction
        self.mapping_learning_rate = mapping_learning_rate
        self.batch_size = batch_size
        self.L2_lambda = L2_lambda
        self.max_epochs = max_epochs
        self.activation = activation
        self.early_stopping_activation = early_stopping_activation
        self.early_stopping_patience = early_stopping_patience
        self.weight_decay = weight_decay
        if device is None:
            device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.device = torch.device(device)
        self.save_model = save_model
        self.save_path = save_path
        self.input_module = input_module
        self.output_module = output_module
        self.add_linear_layer = add_linear_layer
        self.sample_factor_rna = sample_factor_rna
        self.feature_factor_rna = feature_factor_rna
        self.sample_factor_atac = sample_factor_atac
        self.feature_factor_atac = feature_factor_atac
        self.zero_inflated = zero_inflated
        self.dispersion = dispersion
        self.infer_library_size_rna = infer_library_size_rna
        self.infer_library_size_atac = infer_library_size_atac
        self.mapping_module = mapping_module
        self.mapping_hidden_nodes = mapping_hidden_nodes
        self.mapping_non_neg = mapping_non_neg
        self.mapping_layernorm = mapping_layernorm
        self.mapping_batchnorm = mapping_batchnorm
        self.mapping_activation = mapping_activation
        self.mapping_dropout_rate = mapping_dropout_rate
        self.use_mixed_precision = use_mixed_precision     # reference: fp16 autocast; here: bf16 MFMA inputs
        self.gradient_checkpointing = gradient_checkpointing
        self.pin_memory = pin_memory
        self.memory_efficient_loading = memory_efficient_loading
        self.val_chunk = int(val_chunk)
        self.verbose = verbose
        self.grad_sync = grad_sync          # data-parallel hook (scpair_amd.distributed.GradSync) or None
        # data parallel only: every rank owns a COLUMN block of the wide first-layer weights and their AdamW state
        # (distributed.FirstLayerShard) instead of a replica whose dense gradient is all-reduced every step.  Needs the
        # fused encoder shape (no covariates, no BatchNorm) and global batches of at least `world` rows.
        self.shard_first_layer = bool(shard_first_layer) and grad_sync is not None
        # single process: the two predictors are trained side by side, one stream each (same results as one after the
        # other: each keeps its own stream of batch draws; only the numbering of dropout seeds differs)
        self.concurrent_directions = bool(concurrent_directions)
        self.fuse_optimizer = fuse_optimizer  # AdamW step of the first-layer weights inside their dW kernel (1 process)
        # opt-in, single process: full-size minibatch steps are recorded once per direction as a HIP graph and
        # replayed (scpair_amd.graph.CapturedStep; same kernels in the same order, so results do not change)
        if capture_step and not os.environ.get("SCPAIR_EXPERIMENTAL_GRAPH"):
            raise ValueError("capture_step=True is experimental (bitwise-tested at small shapes only; the full-size replay is "
                             "not trusted, DESIGN.md section 7): set SCPAIR_EXPERIMENTAL_GRAPH=1 to use it anyway")
        self.capture_step = capture_step
        # Opt-in input pre-transform per modality, e.g. {'Gene Expression': 'log1p_norm'} or {'Gene Expression':
        # ('log1p_norm', 1e4)}: applied to encoder inputs only, on the fly in the minibatch tile list.  The reference
        # has no normalisation anywhere (model.py:492-510 feeds raw counts), so the default None keeps parity.
        self.input_transform = dict(input_transform or {})
        for k_, v_ in self.input_transform.items():
            mode_ = v_[0] if isinstance(v_, (tuple, list)) else v_
            if mode_ not in (None, "log1p", "log1p_norm"):
                raise ValueError(f"input_transform[{k_!r}] must be None, 'log1p' or 'log1p_norm'")
        self.scaler = None
        self.history = {}
        if self.add_linear_layer is True and self.mapping_non_neg is True:
            raise ValueError('add_linear_layer and mapping_non_neg cannot be both True')
        if self.sample_factor_rna is False and self.infer_library_size_rna is False:
            raise ValueError('at leaset one of sample_factor_rna and infer_library_size_rna should be True')
        if self.sample_factor_rna is True and self.infer_library_size_rna is True:
            raise ValueError('sample_factor_rna and infer_library_size_rna cannot be both True')

    # -- helpers -----------------------------------------------------------------------------------------
    def _say(self, *a):
        if self.verbose:
            print(*a)

    def _precision(self):
        ops.set_precision("bf16" if self.use_mixed_precision else "fp32")

    def _other(self, modality):
        return [mn for mn in self.modality_names if mn != modality][0]

    def _cov_dev(self, d, key):
        v = d.get(key)
        return None if (v is None or self.cov is None) else v.to(self.device).float()

    def data_loader_builder(self):
        """model.py:450-516 for any AnnData-like `scobj` (attributes `.X` [cells x features: scipy sparse or array],
        `.obs` [DataFrame with a 'scPair_split' column and the covariate columns], `.var` [DataFrame with a
        'modality' column]; anndata itself is not needed).  Same covariate handling (one-hot for category/object
        columns, z-score with the pandas sample std for numeric ones, discrete block first), same split dicts and
        keys, same metadata frames -- but the count matrices stay sparse: each '{modality}_{split}_mtx' value is a
        scipy CSR matrix (the reference densifies the whole object here, model.py:492-507), which
        train_predicting_networks() puts into HBM as CSR."""
        import pandas as pd
        import scipy.sparse as sp
        data = self.scobj
        if data is None:
            raise ValueError("data_loader_builder needs scobj (or call train_predicting_networks with your own dicts)")
        obs = data.obs
        cov = self.cov
        discrete_cov, continuous_cov = [], []
        if cov is not None:
            for c in cov:
                if obs[c].dtype.name == 'category' or obs[c].dtype.name == 'object':
                    discrete_cov.append(pd.get_dummies(obs[c]).astype(int))
                else:
                    continuous_cov.append((obs[c] - obs[c].mean()) / obs[c].std())
        self.cov_discrete = pd.concat(discrete_cov, axis=1) if len(discrete_cov) > 0 else None
        self.cov_continuous = pd.concat(continuous_cov, axis=1) if len(continuous_cov) > 0 else None
        parts = [p_ for p_ in (self.cov_discrete, self.cov_continuous) if p_ is not None]
        self.cov_dummy = None if not parts else (parts[0] if len(parts) == 1 else pd.concat(parts, axis=1))
        if self.cov_dummy is not None:
            obs = pd.concat([obs['scPair_split'], self.cov_dummy], axis=1)       # model.py:477
        X = data.X
        X = sp.csr_matrix(X) if not sp.issparse(X) else X.tocsr()
        split_of = np.asarray(data.obs['scPair_split'])
        modality_of = np.asarray(data.var['modality'])
        out = {}
        for split in ['train', 'val', 'test']:
            ridx = np.flatnonzero(split_of == split)
            meta = obs.iloc[ridx].copy()
            setattr(self, f'{split}_metadata', meta)
            Xs = X[ridx]
            d = {}
            for modality in self.modality_names:
                cidx = np.flatnonzero(modality_of == modality)
                m = Xs[:, cidx].tocsr().astype(np.float32)
                m.sort_indices()
                d[f'{modality}_{split}_mtx'] = m
                d[f'{modality}_{split}_cov'] = (torch.FloatTensor(self.cov_dummy.loc[meta.index].values.astype(np.float32))
                                                if cov is not None else None)
                if modality in ['Gene Expression', 'Peaks']:
                    d[f'{modality}_{split}_lib'] = torch.from_numpy(np.asarray(m.sum(1), dtype=np.float32).reshape(-1, 1))
            out[split] = d
        self.data_loader_keys = [k for sp_ in ['train', 'val', 'test'] for k in out[sp_].keys()]
        self.data_loader_dict_train, self.data_loader_dict_val, self.data_loader_dict_test = \
            out['train'], out['val'], out['test']
        self._csr_cache = {}
        return self.data_loader_dict_train, self.data_loader_dict_val, self.data_loader_dict_test

    def set_data(self, train_data_dict, val_data_dict, test_data_dict=None):
        """Registers the split dicts (what data_loader_builder() stores at model.py:512-516)."""
        self.data_loader_dict_train, self.data_loader_dict_val = train_data_dict, val_data_dict
        if test_data_dict is not None:
            self.data_loader_dict_test = test_data_dict
        for split, d in (("train", train_data_dict), ("val", val_data_dict), ("test", test_data_dict)):
            if d is None:
                continue
            k = f"{self.modality_names[0]}_{split}_mtx"
            n = d[k].shape[0]
            if not hasattr(self, f"{split}_metadata") or getattr(self, f"{split}_metadata") is None:
                setattr(self, f"{split}_metadata", _Index(n))
        self._csr_cache = {}

    def _csr(self, d, key, as_input=False):
        """Device CSR of d[key] (cached).  as_input=True: the encoder-input view, which carries the modality's
        opt-in pre-transform; label roles always get the raw counts."""
        cache = self.__dict__.setdefault("_csr_cache", {})
        ck = (id(d), key)
        if ck not in cache:
            cache[ck] = _to_csr(d[key], self.device)
        csr = cache[ck]
        if as_input and self.input_transform:
            tf = next((v for m, v in self.input_transform.items() if key.startswith(m + '_')), None)
            if tf is not None:
                mode, scale = (tf[0], tf[1]) if isinstance(tf, (tuple, list)) else (tf, 1e4)
                csr = csr.with_explicit_values().with_transform(mode, scale)
        return csr

    def run(self):
        self.encoder_dict, self.decoder_dict = self.train_predicting_networks(*self.data_loader_builder())
        low_dim_embeddings, _ = self.reference_embeddings()
        self.mapping_dict = self.train_bidirectional_mapping()
        low_dim_embeddings_mapped, _ = self.mapped_embeddings()
        return self.encoder_dict, self.decoder_dict, self.mapping_dict, low_dim_embeddings, low_dim_embeddings_mapped

    def get_scheduler(self, optimizer):
        """model.py:592-602 ('ExponentialLR' and 'ReduceLROnPlateau' are broken upstream; 'StepLR' works)."""
        if self.weight_decay == 'StepLR':
            return torch.optim.lr_scheduler.StepLR(optimizer=optimizer, step_size=25, gamma=0.1, last_epoch=-1)
        if self.weight_decay == 'ExponentialLR':
            return torch.optim.lr_scheduler.ExponentialLR(optimizer=optimizer, gamma=0.99, last_epoch=-1)
        if self.weight_decay == 'ReduceLROnPlateau':
            return torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer=optimizer, mode='min', factor=0.1, patience=10)
        return None

    # -- inference in chunks (the reference runs whole splits through one forward, model.py:539, 777) -----
    @torch.no_grad()
    def _encode(self, encoder, csr, cov, want_lib=False):
        encoder.eval()
        reps, libs = [], []
        n = csr.n_rows
        for r0 in range(0, n, self.val_chunk):
            rows = np.arange(r0, min(n, r0 + self.val_chunk))
            t = BatchTiles(csr, rows, want_lib=False)
            rep, lat = encoder(t, None if cov is None else cov[r0:r0 + len(rows)])
            reps.append(rep)
            if lat is not None:
                libs.append(lat)
        rep = torch.cat(reps, 0)
        if want_lib:
            return rep, (torch.cat(libs, 0) if libs else None)
        return rep

    @torch.no_grad()
    def _eval_split(self, encoder, decoder, xcsr, ycsr, cov, use_lib, likelihood, shard=None):
        """Full-split validation loss = sum(chunk_loss * chunk_count) / sum(count): identical to the reference's
        single full-split mean (model.py:777).  shard (feature-sharded first layer): xcsr is this rank's column block of
        the split; every chunk is a global batch whose rows the ranks divide among themselves."""
        evaluating_mode(encoder, decoder)
        n = ycsr.n_rows
        num = torch.zeros((), dtype=torch.float64, device=self.device)
        den = torch.zeros((), dtype=torch.float64, device=self.device)
        starts = list(range(0, n, self.val_chunk))
        ends = starts[1:] + [n]
        if shard is not None and len(starts) > 1 and ends[-1] - starts[-1] < shard.world:
            starts.pop(); ends.pop(-2)            # a last chunk with fewer rows than ranks joins the one before it
        for ci, (r0, r1) in enumerate(zip(starts, ends)):
            if shard is None and self.grad_sync is not None and ci % self.grad_sync.world != self.grad_sync.rank:
                continue                          # data parallel: every rank takes every world-th chunk
            rows = np.arange(r0, r1)
            if shard is not None:
                tx, rows = shard.batch_tiles(xcsr, rows)
                ty = BatchTiles(ycsr, rows, want_lib=True)
                lib = ty.lib.reshape(-1, 1) if use_lib else None
                loss = eval_net(tx, ty, lib, encoder, decoder, likelihood, add_cov=False, input_batch=None, output_batch=None)
                cnt = float(len(rows) * ycsr.n_cols) - (ty.nnan.double()[0] if likelihood == "ber" else 0.0)
                cnt = scalar64(cnt, self.device)
                num += loss.double() * cnt
                den += cnt
                continue
            tx = BatchTiles(xcsr, rows, want_lib=False)
            ty = BatchTiles(ycsr, rows, want_lib=True)
            c = None if cov is None else cov[r0:r0 + len(rows)]
            lib = ty.lib.reshape(-1, 1) if use_lib else None
            if likelihood == "gau":
                loss = eval_net(tx, ty.to_dense(), lib, encoder, decoder, likelihood,
                                add_cov=c is not None, input_batch=c, output_batch=c)
                cnt = scalar64(float(len(rows) * ycsr.n_cols), self.device)
            else:
                loss = eval_net(tx, ty, lib, encoder, decoder, likelihood, add_cov=c is not None, input_batch=c,
                                output_batch=c)
                cnt = float(len(rows) * ycsr.n_cols) - (ty.nnan.double()[0] if likelihood == "ber" else 0.0)
                cnt = scalar64(cnt, self.device)
            num += loss.double() * cnt
            den += cnt
        if self.grad_sync is not None:
            num, den = self.grad_sync.reduce_scalars(num, den)
        return num / den

    def reference_embeddings(self):
        """model.py:527-542."""
        import pandas as pd
        embeddings, df_embeddings = {}, {}
        self._precision()
        for modality in self.modality_names:
            encoder_cal = self.encoder_dict[modality + ' to ' + self._other(modality)]
            encoder_cal.eval().to(self.device)
            for split in ['train', 'val', 'test']:
                if hasattr(self, f'data_loader_dict_{split}'):
                    d = getattr(self, f'data_loader_dict_{split}')
                    cov = self._cov_dev(d, f'{modality}_{split}_cov')
                    emb = self._encode(encoder_cal, self._csr(d, f'{modality}_{split}_mtx', as_input=True), cov)
                    embeddings[f'{modality}_{split}'] = emb.cpu().detach()
                    df_embeddings[f'{modality}_{split}'] = pd.DataFrame(
                        embeddings[f'{modality}_{split}'].numpy(), index=getattr(self, f'{split}_metadata').index)
        self.embeddings = embeddings
        return embeddings, df_embeddings

    def mapped_embeddings(self):
        """model.py:543-557."""
        import pandas as pd
        mapped_embeddings, df_mapped_embeddings = {}, {}
        self._precision()
        for input_modality in self.modality_names:
            output_modality = self._other(input_modality)
            mapping_network = self.mapping_dict[input_modality + '_to_' + output_modality]
            mapping_network.eval().to(self.device)
            for split in ['train', 'val', 'test']:
                if hasattr(self, f'data_loader_dict_{split}'):
                    key = f'{input_modality} to {output_modality}_{split}'
                    with torch.no_grad():
                        mapped_embeddings[key] = mapping_network(
                            self.embeddings[input_modality + f'_{split}'].to(self.device)).cpu().detach()
                    df_mapped_embeddings[key] = pd.DataFrame(mapped_embeddings[key].numpy(),
                                                             index=getattr(self, f'{split}_metadata').index)
        return mapped_embeddings, df_mapped_embeddings

    @torch.no_grad()
    def predict(self, out_dtype=None, topk=None):
        """model.py:558-591: decoder outputs [0] for every split (dense numpy arrays), lib = [1].

        Atlas-scale options (not in the reference, whose dense fp32 result is 4 B x cells x features on the host):
        `out_dtype=torch.bfloat16` makes the head kernels write bf16, halving the device-to-host copy (arrays come back
        as torch.bfloat16 tensors, numpy has no such dtype); `topk=k` keeps the k largest predictions per cell and
        returns {'values': [n, k], 'indices': [n, k]} per split."""
        predictions = {}
        od = torch.float32 if out_dtype is None else out_dtype
        if od not in (torch.float32, torch.bfloat16):
            raise ValueError("predict: out_dtype must be None, torch.float32 or torch.bfloat16")
        self._precision()
        for input_modality in self.modality_names:
            output_modality = self._other(input_modality)
            dist = self.modalities[output_modality]
            decoder_cal = self.decoder_dict[input_modality + ' to ' + output_modality]
            decoder_cal.eval().to(self.device)
            need_lat = dist in ['ber', 'nb', 'zinb'] and (
                (output_modality == 'Peaks' and self.infer_library_size_atac) or
                (output_modality == 'Gene Expression' and self.infer_library_size_rna))
            encoder_cal = None
            if need_lat:
                encoder_cal = self.encoder_dict[input_modality + ' to ' + output_modality]
                encoder_cal.eval().to(self.device)
            for split in ['train', 'val', 'test']:
                if not hasattr(self, f'data_loader_dict_{split}'):
                    continue
                d = getattr(self, f'data_loader_dict_{split}')
                emb = self.embeddings[input_modality + f'_{split}'].to(self.device)
                ocov = self._cov_dev(d, f'{output_modality}_{split}_cov')
                lat = None
                if need_lat:
                    icov = self._cov_dev(d, f'{input_modality}_{split}_cov')
                    _, lat = self._encode(encoder_cal, self._csr(d, f'{input_modality}_{split}_mtx', as_input=True), icov, want_lib=True)
                lib = torch.ones(1, device=self.device) if dist in ['ber', 'nb', 'zinb'] else None
                outs, idxs = [], []
                for r0 in range(0, emb.shape[0], self.val_chunk):
                    sl = slice(r0, r0 + self.val_chunk)
                    args = (emb[sl], None if ocov is None else ocov[sl], lib, None if lat is None else lat[sl])
                    if dist != 'gau' and hasattr(decoder_cal, 'predict_head'):
                        pred = decoder_cal.predict_head(*args, out_dtype=od)       # HIP head kernels, [0] only
                    else:
                        pred = decoder_cal(*args)
                        pred = (pred if dist == 'gau' else pred[0]).to(od)
                    if topk is not None:
                        v, ix = torch.topk(pred.float(), int(topk), dim=1)
                        outs.append(v.to(od).cpu()); idxs.append(ix.cpu())
                    else:
                        outs.append(pred.cpu())
                key = output_modality + f"_{split}"
                vals = torch.cat(outs, 0)
                vals = vals.numpy() if od == torch.float32 else vals
                predictions[key] = vals if topk is None else {"values": vals, "indices": torch.cat(idxs, 0).numpy()}
        return predictions

    # -- training ------------------------------------------------------------------------------------------
    def _build_predictor(self, input_modality, output_modality, in_dim, out_dim, n_cov):
        """model.py:618-694: same construction / seeding order, so initial weights equal the reference's."""
        dist = self.modalities[output_modality]
        if self.modalities[input_modality] not in ['zinb', 'nb', 'ber', 'gau'] or dist not in ['zinb', 'nb', 'ber', 'gau']:
            raise ValueError('scPair distribution should be one of the following: zinb, nb, ber, gau')
        set_seed(self.SEED)
        encoder = self.input_module(input_dim=in_dim, input_batch_num=n_cov, hidden_layer=self.hidden_layer,
                                    activation=self.activation, layernorm=self.layernorm, batchnorm=self.batchnorm,
                                    dropout_rate=self.dropout_rate, add_linear_layer=self.add_linear_layer,
                                    infer_library_size=(self.infer_library_size_rna if output_modality == 'Gene Expression'
                                                        else self.infer_library_size_atac))
        use_lib = False
        if dist == 'gau':
            set_seed(self.SEED)
            encoder = self.input_module(input_dim=in_dim, input_batch_num=n_cov, hidden_layer=self.hidden_layer,
                                        activation=self.activation, layernorm=self.layernorm, batchnorm=self.batchnorm,
                                        dropout_rate=self.dropout_rate, add_linear_layer=self.add_linear_layer,
                                        infer_library_size=False)
            set_seed(self.SEED)
            decoder = self.output_module[dist](output_dim=out_dim, output_batch_num=n_cov, hidden_layer=self.hidden_layer,
                                               infer_library_size=False, feature_factor=False)
        elif dist in ['zinb', 'nb']:
            set_seed(self.SEED)
            decoder = self.output_module[dist](output_dim=out_dim, output_batch_num=n_cov, hidden_layer=self.hidden_layer,
                                               infer_library_size=self.infer_library_size_rna,
                                               sample_factor=self.sample_factor_rna, feature_factor=self.feature_factor_rna,
                                               zero_inflated=self.zero_inflated, dispersion=self.dispersion)
            use_lib = self.sample_factor_rna is True
        else:
            set_seed(self.SEED)
            decoder = self.output_module[dist](output_dim=out_dim, output_batch_num=n_cov, hidden_layer=self.hidden_layer,
                                               infer_library_size=self.infer_library_size_atac,
                                               sample_factor=self.sample_factor_atac, feature_factor=self.feature_factor_atac)
            use_lib = self.sample_factor_atac is True
        encoder.apply(init_weights).to(self.device)
        decoder.apply(init_weights).to(self.device)
        return encoder, decoder, use_lib

    def _make_optimizer(self, encoder, decoder):
        """model.py:696-701: four AdamW groups."""
        decay_e, nodecay_e, _, _ = add_weight_decay(encoder, output_layer=['output'])
        decay_d, nodecay_d, _, _ = add_weight_decay(decoder, output_layer=['output'])
        lr = self.learning_rate_prediction
        groups = [{'params': decay_e, 'weight_decay': self.L2_lambda, 'lr': lr},
                  {'params': decay_d, 'weight_decay': self.L2_lambda, 'lr': lr},
                  {'params': nodecay_e, 'weight_decay': 0, 'lr': lr},
                  {'params': nodecay_d, 'weight_decay': 0, 'lr': lr}]
        opt = ops.FusedAdamW([g for g in groups if len(g['params']) > 0])
        self._wire_first_layers(opt, encoder, ('hidden_rep', 'hidden_lib'))
        if self.grad_sync is None and self.fuse_optimizer:
            # the ZINB head's G x G zero-inflation layer (model.py:228): AdamW step inside its weight-gradient GEMM
            # when the layer runs on the wide bf16 kernel (other precisions / small G keep the two-kernel path)
            wide = getattr(decoder, 'output_dropout', None)
            if isinstance(wide, nn.Linear) and wide.weight.requires_grad:
                opt.fuse_weight_update(wide.weight)
        return opt

    def _wire_first_layers(self, opt, encoder, names):
        """The wide sparse first-layer weights of `encoder`.  Single process: their AdamW step runs inside their
        weight-gradient kernel (precision 'bf16', no covariate block; other cases keep the two-kernel path).  Data
        parallel with a bf16 exchange (GradSync(compress=True), precision 'bf16'): the kernel writes dW as bf16
        straight into the exchange buffer that the all-reduce sums and the optimiser reads."""
        firsts = []
        for seq in (getattr(encoder, n_, None) for n_ in names):
            while isinstance(seq, nn.Sequential) and len(seq) > 0:
                seq = seq[0]
            if isinstance(seq, nn.Linear):
                firsts.append(seq.weight)
        opt._dp_sink = None
        if self.grad_sync is None or (firsts and all(getattr(p, "_scp_feature_sharded", False) for p in firsts)):
            # single process, or feature-sharded data parallel (the block's gradient is complete on its rank)
            if self.fuse_optimizer:
                opt.fuse_first_layer(*firsts)
        elif self.grad_sync.compress and ops.get_precision() == "bf16" and firsts and all(p.is_cuda for p in firsts):
            opt._dp_sink = self.grad_sync.make_sink(firsts)
        return firsts

    def _dp_step(self, loss, n_valid, optimizer, some_rank_empty=False, scalars=None):
        """Data-parallel tail of one optimisation step: backward on this rank's shard (loss None: empty shard), the
        gradient exchange, the identical AdamW step on every rank.  Returns the global-batch loss."""
        optimizer.zero_grad()
        if ops.bn_sync_active():        # BatchNorm's backward sums weigh the ranks like the loss does
            ops.set_bn_sync_weight(self.grad_sync.share_of(n_valid, self.device))
        if loss is not None:
            loss.backward()
        params = [p for g in optimizer.param_groups for p in g['params'] if not getattr(p, "_scp_feature_sharded", False)]
        gl = self.grad_sync.exchange(params, n_valid, loss, sink=getattr(optimizer, '_dp_sink', None),
                                     some_rank_empty=some_rank_empty, scalars=scalars)
        optimizer.step()
        return gl

    def _check_dp(self, *modules):
        """Data-parallel training of modules with BatchNorm1d: remember that they have such layers; the loops then switch
        the global-batch statistics on per step (_bn_step)."""
        self._dp_has_bn = self.grad_sync is not None and any(isinstance(m, nn.BatchNorm1d) for mod in modules
                                                             for m in mod.modules())

    def _bn_step(self, empty_somewhere):
        """Before a data-parallel step's forward: BatchNorm statistics over the global batch (ops.set_bn_sync).  Off for a
        step in which some rank's shard is empty (a last global batch smaller than the world): that rank runs no forward
        and could not join the layers' collectives; the other ranks use their own statistics for that one step."""
        if getattr(self, "_dp_has_bn", False):
            ops.set_bn_sync(None if empty_somewhere else self.grad_sync)

    def _concurrent_directions_ok(self, batch_replay):
        """Both predictors trained side by side, each on its own stream (the two share nothing but the data): single
        process, eager steps, sparse labels for both.  Off: `concurrent_directions=False` / SCPAIR_SEQUENTIAL_DIRECTIONS=1."""
        return bool(self.concurrent_directions and not os.environ.get("SCPAIR_SEQUENTIAL_DIRECTIONS")
                    and self.grad_sync is None and not self.capture_step and len(self.modality_names) == 2
                    and 'gau' not in self.modality_distributions)

    def _run_interleaved(self, tasks):
        """Round-robin over the directions' tasks, one optimisation step (or one epoch end) at a time, each resumed under its
        own stream: the kernels of one direction fill the gaps of the other (its HBM-bound weight-gradient + AdamW kernel
        next to the other's MFMA- and vector-bound ones), as `bench.py`'s paired step does.  A task that has stopped
        (early stopping, max_epochs) drops out; the other goes on alone."""
        cur = torch.cuda.current_stream(self.device)
        streams = [torch.cuda.Stream(self.device) for _ in tasks]
        for s in streams:
            s.wait_stream(cur)            # matrices, weights: created on the caller's stream
        live = list(range(len(tasks)))
        while live:
            for i in list(live):
                with torch.cuda.stream(streams[i]):
                    try:
                        next(tasks[i])
                    except StopIteration:
                        live.remove(i)
        for s in streams:
            cur.wait_stream(s)

    def _predictor_task(self, di, input_modality, train_data_dict, val_data_dict, batch_replay, encoder_dict, decoder_dict,
                        concurrent):
        """One direction of train_predicting_networks (model.py:612-811) as a generator: run to exhaustion it is the
        reference's loop; with `concurrent` it hands control back after set-up ("ready"), after every optimisation step and
        after every epoch, so that _run_interleaved can alternate between the two directions."""

        output_modality = self._other(input_modality)
        prediction_order = {'input': input_modality, 'output': output_modality}
        self._say('Set up predicting networks:', prediction_order)
        xtr = self._csr(train_data_dict, input_modality + '_train_mtx', as_input=True)
        ytr = self._csr(train_data_dict, output_modality + '_train_mtx')
        xva = self._csr(val_data_dict, input_modality + '_val_mtx', as_input=True)
        yva = self._csr(val_data_dict, output_modality + '_val_mtx')
        ctr = self._cov_dev(train_data_dict, input_modality + '_train_cov')
        cva = self._cov_dev(val_data_dict, input_modality + '_val_cov')
        n_cov = 0 if ctr is None else ctr.shape[1]
        dist = self.modalities[output_modality]
        encoder, decoder, use_lib = self._build_predictor(input_modality, output_modality, xtr.n_cols, ytr.n_cols, n_cov)
        shard = None
        if self.grad_sync is not None:
            self._check_dp(encoder, decoder)
            self.grad_sync.broadcast(list(encoder.parameters()) + list(decoder.parameters()) +
                                     list(encoder.buffers()) + list(decoder.buffers()))
            if self.shard_first_layer:
                if n_cov or self.batchnorm or dist == 'gau':
                    raise ValueError("shard_first_layer needs cov=None, batchnorm=False and a count / Bernoulli output")
                from .distributed import FirstLayerShard
                shard = FirstLayerShard(self.grad_sync, xtr.n_cols)
                shard.shard_module(encoder)
                xtr_cols, xva_cols = shard.col_csr(xtr), shard.col_csr(xva)
        optimizer = self._make_optimizer(encoder, decoder)
        scheduler = self.get_scheduler(optimizer)
        set_seed(self.SEED)
        rng_state = torch.get_rng_state() if concurrent else None     # this direction's own stream of loader draws
        loader = _index_loader(xtr.n_rows, self.batch_size) if batch_replay is None else None
        ytr_dense = ytr.to_dense().to(self.device) if dist == 'gau' else None
        if not self.early_stopping_activation:
            raise TypeError("early_stopping_activation=False is unusable upstream (model.py:719, 786: "
                            "`val < None`); keep it True")
        patience, no_improve, min_val = self.early_stopping_patience, 0, np.inf
        hist = {"train_loss": [], "val_loss": []}
        replay = None if batch_replay is None else list(batch_replay[di])
        per_epoch = None if replay is None else int(np.ceil(xtr.n_rows / self.batch_size))
        encoder_ckpt = decoder_ckpt = None
        stepper = _GraphStepper(self, xtr, ytr, ctr, use_lib, encoder, decoder, optimizer, dist) \
            if (self.capture_step and self.grad_sync is None and scheduler is None and dist != 'gau') else None
        if concurrent:
            yield "ready"
        for epoch in range(self.max_epochs):
            training_mode(encoder, decoder)
            if replay is not None:
                batches = replay[epoch * per_epoch:(epoch + 1) * per_epoch]
                if not batches:
                    break
            elif not concurrent:
                batches = (b[0].numpy() for b in loader)
            else:
                # the loader draws from the global torch RNG when an epoch's iterator comes into being (its base seed and
                # the sampler's seed); with the other direction's task interleaved, each direction keeps the RNG stream it
                # would have had on its own, so batches are those of the one-after-the-other order
                outer = torch.get_rng_state()
                torch.set_rng_state(rng_state)
                it = iter(loader)
                first = next(it, None)
                rng_state = torch.get_rng_state()
                torch.set_rng_state(outer)
                batches = (b[0].numpy() for b in itertools.chain(() if first is None else (first,), it))
            step_losses = []
            for rows in batches:
                empty_somewhere = False
                tx = None
                if shard is not None:
                    # feature-sharded first layer: the encoder input is the GLOBAL batch over this rank's columns
                    tx, rows = shard.batch_tiles(xtr_cols, rows)
                elif self.grad_sync is not None:
                    empty_somewhere = self.grad_sync.some_rank_empty(len(rows))
                    self._bn_step(empty_somewhere)
                    rows = self.grad_sync.shard(rows)
                    if len(rows) == 0:            # global batch smaller than the world: join the collectives only
                        step_losses.append(self._dp_step(None, 0.0, optimizer, True))
                        continue
                rows = np.asarray(rows, dtype=np.int64)
                rows_dev = rows_to_device(rows, self.device)          # one upload, shared by both modalities
                if stepper is not None and stepper.wants(rows):
                    step_losses.append(stepper.step(rows, rows_dev))
                    continue
                if tx is None:
                    tx = BatchTiles(xtr, rows_dev, want_lib=False, rows_host=rows)
                c = None if ctr is None else ctr[rows_dev]
                if dist == 'gau':
                    ty, lib = ytr_dense[rows_dev], None
                else:
                    ty = BatchTiles(ytr, rows_dev, want_lib=use_lib, rows_host=rows)
                    lib = ty.lib.reshape(-1, 1) if use_lib else None
                loss = self._train_step(tx, ty, lib, encoder, decoder, optimizer, dist, c, empty_somewhere)
                step_losses.append(loss)
                if concurrent:
                    yield "step"
            hist["train_loss"].extend(step_losses)
            val_total_loss = float(self._eval_split(encoder, decoder, xva if shard is None else xva_cols, yva, cva,
                                                    use_lib, dist, shard=shard).item())
            hist["val_loss"].append(val_total_loss)
            if scheduler is not None and optimizer.param_groups[0]['lr'] > 1e-7:
                scheduler.step()
            if val_total_loss < min_val:
                no_improve, min_val = 0, val_total_loss
                encoder_ckpt, decoder_ckpt = _snapshot(encoder, encoder_ckpt), _snapshot(decoder, decoder_ckpt)
            else:
                no_improve += 1
            self._say(f"[{input_modality} -> {output_modality}] epoch {epoch + 1} validation loss {val_total_loss:.6f}")
            if epoch > 1 and no_improve == patience:
                self._say("Early Stopped!")
                break
            if concurrent:
                yield "epoch"
        hist["train_loss"] = [float(v) for v in torch.stack(hist["train_loss"]).cpu()] if hist["train_loss"] else []
        if stepper is not None:
            stepper.release()
        key = input_modality + ' to ' + output_modality
        self.history[key] = hist
        encoder_dict[key] = copy.deepcopy(encoder_ckpt.eval())
        decoder_dict[key] = copy.deepcopy(decoder_ckpt.eval())
        if shard is not None:                 # the returned encoder holds the full first-layer weights on every rank
            shard.unshard_module(encoder_dict[key])
        if self.save_model is True:
            if self.save_path is None:
                self.save_path = os.getcwd()
            torch.save(encoder_dict[key], self.save_path + '/encoder_' + input_modality + '_to_' + output_modality + '.pt')
            torch.save(decoder_dict[key], self.save_path + '/decoder_' + input_modality + '_to_' + output_modality + '.pt')
        self._say(f"Completed training for {input_modality} to {output_modality}")
        if concurrent:
            self._rng_after[di] = rng_state

    def train_predicting_networks(self, train_data_dict, val_data_dict, test_data_dict, batch_replay=None):
        """model.py:603-812.  `batch_replay`: optional {direction_index: [row-id arrays]} to replay a recorded batch
        order (parity tests); otherwise batches come from a DataLoader over row ids driven by the global torch RNG
        exactly like the reference's."""
        if self.device.type != "cuda":
            raise RuntimeError("scpair_amd trains on a GPU only (device must be cuda); there is no CPU path")
        self._precision()
        self.set_data(train_data_dict, val_data_dict, test_data_dict)
        encoder_dict, decoder_dict = {}, {}
        concurrent = self._concurrent_directions_ok(batch_replay)
        tasks = []
        for di, input_modality in enumerate(self.modalities.keys()):
            task = self._predictor_task(di, input_modality, train_data_dict, val_data_dict, batch_replay, encoder_dict,
                                        decoder_dict, concurrent)
            if concurrent:
                next(task)            # construction, initialisation, optimiser: in the reference's order, one direction at a time
                tasks.append(task)
            else:
                for _ in task:        # the reference's order: this direction to completion before the next is set up
                    pass
        if concurrent:
            self._rng_after = {}
            self._run_interleaved(tasks)
            # the global torch RNG is left where the one-after-the-other order leaves it: behind the last direction's draws
            torch.set_rng_state(self._rng_after[len(tasks) - 1])
            order = [m + ' to ' + self._other(m) for m in self.modalities.keys()]
            encoder_dict = {k: encoder_dict[k] for k in order}
            decoder_dict = {k: decoder_dict[k] for k in order}
            self.history.update({k: self.history.pop(k) for k in order})
        self.encoder_dict, self.decoder_dict = encoder_dict, decoder_dict
        ops.set_bn_sync(None)
        return encoder_dict, decoder_dict

    def _train_step(self, tx, ty, lib, encoder, decoder, optimizer, dist, cov, some_rank_empty=False):
        if self.grad_sync is None:
            return train_net(tx, ty, lib, encoder, decoder, optimizer, likelihood_type=dist, add_cov=cov is not None,
                             input_batch=cov, output_batch=cov)
        from .utils import _predictor_loss
        loss = _predictor_loss(tx, ty, lib, encoder, decoder, dist, cov is not None, cov, cov)
        # weight of this rank's mean in the global-batch mean = its share of the valid label elements
        n_valid = float(ty.shape[0] * ty.shape[1]) if not isinstance(ty, BatchTiles) or dist != "ber" else \
            (float(ty.B * ty.F) - ty.nnan.double()[0])
        shd = getattr(tx, "shard", None)
        scalars = None
        if shd is not None:                       # read by the first layer's backward: weight of this rank's dY rows
            scalars = self.grad_sync.step_scalars(n_valid, loss, self.device)      # the step's one scalar collective, early
            shd.share = self.grad_sync.share_from(scalars)
        return self._dp_step(loss, n_valid, optimizer, some_rank_empty, scalars=scalars)

    def train_bidirectional_mapping(self):
        """model.py:813-924."""
        self._precision()
        embedding_dim = self.hidden_layer[-1]
        mapping_dict = {}
        for input_modality in self.modality_names:
            output_modality = self._other(input_modality)
            self._say('mapping direction:', input_modality, 'to', output_modality)
            xin = self.embeddings[input_modality + '_train'].to(self.device)
            yout = self.embeddings[output_modality + '_train'].to(self.device)
            xva = self.embeddings[input_modality + '_val'].to(self.device)
            yva = self.embeddings[output_modality + '_val'].to(self.device)
            set_seed(self.SEED)
            loader = _index_loader(xin.shape[0], self.batch_size)
            mapping_network = self.mapping_module(origin_module_dim=embedding_dim, target_module_dim=embedding_dim,
                                                  translational_hidden_nodes=self.mapping_hidden_nodes,
                                                  non_neg=self.mapping_non_neg, activation=self.mapping_activation,
                                                  layernorm=self.mapping_layernorm, batchnorm=self.mapping_batchnorm,
                                                  dropout_rate=self.mapping_dropout_rate)
            mapping_network.apply(init_weights).to(self.device)
            decay_param, nodecay_param, _, _ = add_weight_decay(mapping_network, output_layer="bias")
            groups = [{'params': decay_param, 'weight_decay': self.L2_lambda, 'lr': self.mapping_learning_rate},
                      {'params': nodecay_param, 'weight_decay': 0, 'lr': self.mapping_learning_rate}]
            optimizer = ops.FusedAdamW([g for g in groups if len(g['params']) > 0])
            scheduler = self.get_scheduler(optimizer)
            if self.grad_sync is not None:
                self._check_dp(mapping_network)
                self.grad_sync.broadcast(list(mapping_network.parameters()) + list(mapping_network.buffers()))
            patience, no_improve, min_val = self.early_stopping_patience, 0, np.inf
            hist = {"train_loss": [], "val_loss": []}
            ckpt = None
            for epoch in range(self.max_epochs):
                mapping_network.train()
                for (idx,) in loader:
                    if self.grad_sync is not None:
                        # data parallel: this rank's slice of the global batch, count-weighted gradient exchange
                        empty_somewhere = self.grad_sync.some_rank_empty(len(idx))
                        self._bn_step(empty_somewhere)
                        idx = torch.from_numpy(self.grad_sync.shard(idx.numpy()))
                        if len(idx) == 0:
                            hist["train_loss"].append(self._dp_step(None, 0.0, optimizer, True))
                            continue
                        idx = rows_to_device(idx.numpy(), self.device)
                        training_loss = scaled_loss(yout[idx], mapping_network(xin[idx]))
                        hist["train_loss"].append(self._dp_step(training_loss, float(len(idx) * embedding_dim), optimizer,
                                                                empty_somewhere))
                        continue
                    idx = rows_to_device(idx.numpy(), self.device)
                    pred_emb = mapping_network(xin[idx])
                    training_loss = scaled_loss(yout[idx], pred_emb)
                    optimizer.zero_grad()
                    training_loss.backward()
                    optimizer.step()
                    hist["train_loss"].append(training_loss.detach())
                mapping_network.eval()
                with torch.no_grad():
                    val_total_loss = float(scaled_loss(yva, mapping_network(xva)).item())
                hist["val_loss"].append(val_total_loss)
                if scheduler is not None and optimizer.param_groups[0]['lr'] > 1e-7:
                    scheduler.step()
                if val_total_loss < min_val:
                    no_improve, min_val = 0, val_total_loss
                    ckpt = _snapshot(mapping_network, ckpt)
                else:
                    no_improve += 1
                if epoch > 1 and no_improve == patience:
                    self._say("Early Stopped!")
                    break
            hist["train_loss"] = [float(v) for v in torch.stack(hist["train_loss"]).cpu()] if hist["train_loss"] else []
            key = input_modality + '_to_' + output_modality
            self.history[key] = hist
            mapping_dict[key] = copy.deepcopy(ckpt)
            if self.save_model is True:
                if self.save_path is None:
                    self.save_path = os.getcwd()
                torch.save(mapping_dict[key], self.save_path + '/mapping_' + input_modality + '_to_' + output_modality + '.pt')
            self._say(f"Completed mapping training for {input_modality} to {output_modality}")
        self.mapping_dict = mapping_dict
        ops.set_bn_sync(None)
        return mapping_dict

    def _augment_covariates(self, unimodal_scobj, unimodal_cov, n_uni, n_cov):
        """model.py:936-947.  Upstream, the only covariate combination that runs is "paired data with covariates,
        unimodal data without": the unimodal cells then get an all-zero covariate block.  A unimodal covariate list
        without paired covariates raises like upstream.  Equal non-empty lists fail upstream (`unimodal_cov_dummy` is
        never assigned, model.py:950); here they are encoded like data_loader_builder does (one-hot / z-score), aligned
        to the paired data's covariate columns (levels the paired data never saw are dropped, missing ones are zero)."""
        if unimodal_cov is not None and self.cov is None:
            raise ValueError('Unimodal data have covariate matrix while multimodal data do not')
        if self.cov is None:
            return None
        if unimodal_cov is None:
            self._say('Using multimodal cov for unimodal data, and set unimodal cov to all zeros during scPair_augment training')
            return torch.zeros((n_uni, n_cov), dtype=torch.float32, device=self.device)
        if list(unimodal_cov) != list(self.cov):
            raise ValueError('unimodal_cov must be None or the paired data\'s covariate list '
                             '(upstream drops the unimodal covariates here and then fails on the encoder input width)')
        if getattr(self, 'cov_dummy', None) is None or not hasattr(unimodal_scobj, 'obs'):
            raise ValueError('unimodal covariates need data_loader_builder() (covariate columns) and unimodal_scobj.obs')
        import pandas as pd
        obs = unimodal_scobj.obs
        parts = []
        for c in unimodal_cov:
            if obs[c].dtype.name in ('category', 'object'):
                parts.append(pd.get_dummies(obs[c]).astype(int))
            else:
                parts.append(((obs[c] - obs[c].mean()) / obs[c].std()).to_frame())
        dm = pd.concat(parts, axis=1).reindex(columns=self.cov_dummy.columns, fill_value=0)
        return torch.tensor(dm.values.astype(np.float32), device=self.device)

    def augment(self, unimodal_scobj, unimodal_modalities, unimodal_cov=None):
        """model.py:925-1036: fine-tune a copy of the trained encoder on unimodal cells to regress its own embeddings
        (dropout active), early-stopped on the bimodal train+val cells.  `unimodal_scobj` needs `.X` (scipy sparse,
        ndarray or DeviceCSR) and, for covariates, `.obs`.  With `grad_sync` every global batch is sliced by rank and
        the gradients are exchanged like in train_predicting_networks."""
        self._precision()
        self.unimodal_scobj, self.unimodal_modalities, self.unimodal_cov = unimodal_scobj, unimodal_modalities, unimodal_cov
        self.unimodal_modality_names = list(unimodal_modalities.keys())
        if not set(self.unimodal_modality_names).issubset(self.modalities.keys()):
            raise ValueError('Unimodal data should have the same modality as one of the multimodal data')
        name = self.unimodal_modality_names[0]
        other = [mn for mn in self.modalities.keys() if mn != name][0]
        X = unimodal_scobj.X if hasattr(unimodal_scobj, "X") else unimodal_scobj
        uni = _to_csr(X, self.device)
        tf = self.input_transform.get(name)
        if tf is not None:
            uni = uni.with_transform(*((tf[0], tf[1]) if isinstance(tf, (tuple, list)) else (tf, 1e4)))
        btr = self._csr(self.data_loader_dict_train, name + '_train_mtx', as_input=True)
        bva = self._csr(self.data_loader_dict_val, name + '_val_mtx', as_input=True)
        has_test = getattr(self, 'data_loader_dict_test', None) is not None
        bte = self._csr(self.data_loader_dict_test, name + '_test_mtx', as_input=True) if has_test else None
        ctr = self._cov_dev(self.data_loader_dict_train, name + '_train_cov')
        cva = self._cov_dev(self.data_loader_dict_val, name + '_val_cov')
        cte = self._cov_dev(self.data_loader_dict_test, name + '_test_cov') if has_test else None
        cuni = self._augment_covariates(unimodal_scobj, unimodal_cov, uni.n_rows, 0 if ctr is None else ctr.shape[1])
        enc = copy.deepcopy(self.encoder_dict[name + ' to ' + other]).to(self.device)
        enc.eval()
        uni_label = self._encode(enc, uni, cuni)
        btr_label, bva_label = self._encode(enc, btr, ctr), self._encode(enc, bva, cva)
        set_seed(self.SEED)
        loader = _index_loader(uni.n_rows, self.batch_size)
        optimizer = ops.FusedAdamW(enc.parameters(), lr=self.learning_rate_prediction, weight_decay=self.L2_lambda)
        if cuni is None:
            self._wire_first_layers(optimizer, enc, ('hidden_rep',))     # only latent_rep is trained (model.py:993)
        if self.grad_sync is not None:
            self._check_dp(enc)
            self.grad_sync.broadcast(list(enc.parameters()) + list(enc.buffers()))
        patience, no_improve, min_val = self.early_stopping_patience, 0, np.inf
        self.augment_encoder_ckpt = None
        emb_dim = self.hidden_layer[-1]
        hist = {"val_loss": []}
        for epoch in range(self.max_epochs):
            enc.train()
            for (idx,) in loader:
                rows = idx.numpy()
                empty_somewhere = False
                if self.grad_sync is not None:
                    empty_somewhere = self.grad_sync.some_rank_empty(len(rows))
                    self._bn_step(empty_somewhere)
                    rows = self.grad_sync.shard(rows)
                    if len(rows) == 0:
                        self._dp_step(None, 0.0, optimizer, True)
                        continue
                rows_dev = rows_to_device(np.asarray(rows, dtype=np.int64), self.device)
                pred = enc(BatchTiles(uni, rows_dev, want_lib=False, rows_host=rows),
                           None if cuni is None else cuni[rows_dev])[0]
                loss = scaled_loss(uni_label[rows_dev], pred)
                if self.grad_sync is not None:
                    self._dp_step(loss, float(len(rows) * emb_dim), optimizer, empty_somewhere)
                    continue
                optimizer.zero_grad()
                loss.backward()
                optimizer.step()
            with torch.no_grad():
                pv = torch.cat([self._encode(enc, btr, ctr), self._encode(enc, bva, cva)], 0)
                val_total_loss = float(scaled_loss(torch.cat([btr_label, bva_label], 0), pv).item())
            hist["val_loss"].append(val_total_loss)
            if val_total_loss < min_val:
                no_improve, min_val = 0, val_total_loss
                self.augment_encoder_ckpt = _snapshot(enc, self.augment_encoder_ckpt)
            else:
                no_improve += 1
            if epoch > 1 and no_improve == patience:
                self._say("Early Stopped!")
                break
        self.history['scPair_augment: ' + name + ' to ' + other] = hist
        self.augment_encoder_ckpt.eval()
        if self.save_model is True:
            if self.save_path is None:
                self.save_path = os.getcwd()
            torch.save(self.augment_encoder_ckpt, self.save_path + '/augment_encoder_' + name + '_to_' + other + '.pt')
        self.augment_encoder_dict = {'scPair_augment: ' + name + ' to ' + other: copy.deepcopy(self.augment_encoder_ckpt)}
        augment_emb = {}
        for split, csr, cv in (('unimodal_train', uni, cuni), ('bimodal_train', btr, ctr), ('bimodal_val', bva, cva),
                               ('bimodal_test', bte, cte)):
            if csr is None:
                continue
            augment_emb[name + f'_{split}'] = self._encode(self.augment_encoder_ckpt, csr, cv).cpu().numpy()
        ops.set_bn_sync(None)
        return self.augment_encoder_dict, augment_emb
