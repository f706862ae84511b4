"""Epoch loop for the unimodal VAE (BASELINE config 4: atlas pretraining on one modality).

The reference ships VAE_ENCODER / VAE_DECODER (pairedVAE.py:24-187) and the step functions train_VAE / eval_VAE
(utils.py:179-228) but no caller.  `scVAE_object` supplies the loop in the shape of the predictor loop
(scPair_object.train_predicting_networks, model.py:603-812): seeded construction and Xavier initialisation, AdamW
groups from add_weight_decay, a row-id DataLoader driven by the global torch RNG, per-epoch validation on the whole
validation split (here in chunks, reduced to the same means), early stopping on the validation total with a deep copy
of the best encoder / decoder.  The count matrix stays in HBM as CSR; single-process training runs the AdamW step of the
wide first-layer weights inside their weight-gradient kernel; with `grad_sync` every global batch is sliced by rank and
the gradients are exchanged (scpair_amd.distributed.GradSync).
"""
from __future__ import annotations

import copy
import os

import numpy as np
import torch
from torch import nn

from . import ops
from .pairedVAE import VAE_DECODER, VAE_ENCODER
from .sparse import BatchTiles, rows_to_device, scalar64
from .utils import _vae_losses, add_weight_decay, init_weights, set_seed


class scVAE_object:
    def __init__(self, distribution="nb", SEED=0, hidden_dim=[900], latent_dim=40, dropout_rate=0.1, batchnorm=False,
                 layernorm=True, learning_rate=1e-3, batch_size=130, L2_lambda=1e-9, max_epochs=1000,
                 activation=nn.LeakyReLU(), early_stopping_patience=25, kl_weight=1, device=None, save_model=False,
                 save_path=None, sample_factor=True, infer_library_size=False, feature_factor=False,
                 dispersion="feature-cell", use_mixed_precision=False, val_chunk=4096, verbose=True, grad_sync=None,
                 fuse_optimizer=True, weight_decay_groups=True):
        if distribution not in ("zinb", "nb", "gau", "ber"):
            raise ValueError("distribution should be one of the following: zinb, nb, gau, ber")
        self.distribution, self.SEED, self.hidden_dim, self.latent_dim = distribution, SEED, list(hidden_dim), latent_dim
        self.dropout_rate, self.batchnorm, self.layernorm, self.activation = dropout_rate, batchnorm, layernorm, activation
        self.learning_rate, self.batch_size, self.L2_lambda, self.max_epochs = learning_rate, batch_size, L2_lambda, max_epochs
        self.early_stopping_patience, self.kl_weight = early_stopping_patience, kl_weight
        if device is None:
            device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.device = torch.device(device)
        self.save_model, self.save_path = save_model, save_path
        self.sample_factor, self.infer_library_size, self.feature_factor = sample_factor, infer_library_size, feature_factor
        self.dispersion, self.use_mixed_precision, self.val_chunk, self.verbose = dispersion, use_mixed_precision, int(val_chunk), verbose
        self.grad_sync, self.fuse_optimizer, self.weight_decay_groups = grad_sync, fuse_optimizer, weight_decay_groups
        self.history = {"train_loss": [], "val_loss": [], "val_recon": [], "val_kl": []}
        self.encoder = self.decoder = None
        self.eps_source = None          # optional callable(shape, device) -> noise for the encoder (parity tests)

    def _say(self, *a):
        if self.verbose:
            print(*a)

    def build(self, input_dim, n_cov=0):
        """Same seeding / construction order as the predictor (model.py:620-694): seed, encoder, seed, decoder, Xavier."""
        kw = dict(input_dim=input_dim, input_batch_num=n_cov, hidden_dim=self.hidden_dim, latent_dim=self.latent_dim,
                  layernorm=self.layernorm, activation=self.activation, batchnorm=self.batchnorm,
                  dropout_rate=self.dropout_rate, infer_library_size=self.infer_library_size)
        set_seed(self.SEED)
        enc = VAE_ENCODER(**kw)
        set_seed(self.SEED)
        dec = VAE_DECODER(**kw, sample_factor=self.sample_factor, feature_factor=self.feature_factor,
                          distribution=self.distribution, dispersion=self.dispersion)
        enc.apply(init_weights)
        dec.apply(init_weights)
        return enc, dec

    def _optimizer(self, enc, dec):
        lr = self.learning_rate
        if self.weight_decay_groups:
            de, ne, _, _ = add_weight_decay(enc, output_layer=["output"])
            dd, nd, _, _ = add_weight_decay(dec, output_layer=["output"])
            groups = [{"params": de, "weight_decay": self.L2_lambda, "lr": lr}, {"params": dd, "weight_decay": self.L2_lambda, "lr": lr},
                      {"params": ne, "weight_decay": 0, "lr": lr}, {"params": nd, "weight_decay": 0, "lr": lr}]
            opt = ops.FusedAdamW([g for g in groups if g["params"]])
        else:
            opt = ops.FusedAdamW(list(enc.parameters()) + list(dec.parameters()), lr=lr, weight_decay=self.L2_lambda)
        firsts = [enc.enc_hidden[0].weight] + ([enc.enc_lib[0][0].weight] if self.infer_library_size else [])
        opt._dp_sink = None
        if self.grad_sync is None:
            if self.fuse_optimizer:
                opt.fuse_first_layer(*firsts)
                # (the decoder's 900 -> G heads keep the two-kernel update: fused into their weight-gradient GEMMs the
                # step measured 1.95 ms against 1.79 -- 900 columns leave a quarter of the tiles ragged and the short
                # K loop cannot hide the epilogue's HBM traffic)
        elif self.grad_sync.compress and ops.get_precision() == "bf16":
            opt._dp_sink = self.grad_sync.make_sink(firsts)
        return opt

    def _lib(self, tiles):
        return tiles.lib.reshape(-1, 1) if (self.sample_factor and self.distribution in ("nb", "zinb")) else None

    @torch.no_grad()
    def evaluate(self, csr, cov=None):
        """eval_VAE (utils.py:206-228) over a whole split in chunks: (total, reconstruction, KL), each the mean the
        reference's single full-split call returns (sum of chunk means x chunk counts / total count)."""
        enc, dec = self.encoder, self.decoder
        enc.eval(); dec.eval()
        acc = torch.zeros(4, dtype=torch.float64, device=self.device)      # recon*n, n_recon, kl*n, n_kl
        for ci, r0 in enumerate(range(0, csr.n_rows, self.val_chunk)):
            if self.grad_sync is not None and ci % self.grad_sync.world != self.grad_sync.rank:
                continue
            rows = np.arange(r0, min(csr.n_rows, r0 + self.val_chunk))
            t = BatchTiles(csr, rows, want_lib=True)
            c = None if cov is None else cov[r0:r0 + len(rows)]
            x = t if self.distribution in ("nb", "zinb") else t.to_dense()
            _, recon, kl = _vae_losses(x, self._lib(t), enc, dec, self.kl_weight, self.distribution, c is not None, c)
            n = float(len(rows))
            n_dev = scalar64(n, self.device)
            acc += torch.stack([recon.double() * n, n_dev, kl.double() * n, n_dev])
        if self.grad_sync is not None:
            self.grad_sync._sum(acc)
        recon, kl = acc[0] / acc[1], acc[2] / acc[3]
        return self.kl_weight * kl + recon, recon, kl

    def train(self, train_mtx, val_mtx, train_cov=None, val_cov=None, batch_replay=None, init_state=None):
        """train_mtx / val_mtx: dense tensor, scipy sparse or DeviceCSR [cells x features]; covariates: [cells x C]
        tensors or None.  `batch_replay`: optional list of row-id arrays (parity tests); `init_state`: optional
        (encoder_state_dict, decoder_state_dict) to start from.  Returns (encoder, decoder): the best-validation copies."""
        from .model import _index_loader, _snapshot, _to_csr
        if self.device.type != "cuda":
            raise RuntimeError("scpair_amd trains on a GPU only (device must be cuda); there is no CPU path")
        ops.set_precision("bf16" if self.use_mixed_precision else "fp32")
        xtr, xva = _to_csr(train_mtx, self.device), _to_csr(val_mtx, self.device)
        ctr = None if train_cov is None else train_cov.to(self.device).float()
        cva = None if val_cov is None else val_cov.to(self.device).float()
        enc, dec = self.build(xtr.n_cols, 0 if ctr is None else ctr.shape[1])
        if init_state is not None:
            enc.load_state_dict(init_state[0]); dec.load_state_dict(init_state[1])
        enc.to(self.device); dec.to(self.device)
        enc.eps_source = self.eps_source
        self.encoder, self.decoder = enc, dec
        opt = self._optimizer(enc, dec)
        if self.grad_sync is not None:
            if self.batchnorm:
                raise NotImplementedError("data-parallel training with batchnorm=True needs synchronised batch statistics")
            self.grad_sync.broadcast(list(enc.parameters()) + list(dec.parameters()) + list(enc.buffers()) + list(dec.buffers()))
        set_seed(self.SEED)
        loader = _index_loader(xtr.n_rows, self.batch_size) if batch_replay is None else None
        replay = None if batch_replay is None else list(batch_replay)
        per_epoch = None if replay is None else int(np.ceil(xtr.n_rows / self.batch_size))
        patience, no_improve, min_val = self.early_stopping_patience, 0, np.inf
        best = None
        dense_in = self.distribution not in ("nb", "zinb")
        for epoch in range(self.max_epochs):
            enc.train(); dec.train()
            if replay is not None:
                batches = replay[epoch * per_epoch:(epoch + 1) * per_epoch]
                if not batches:
                    break
            else:
                batches = (b[0].numpy() for b in loader)
            for rows in batches:
                empty_somewhere = False
                if self.grad_sync is not None:
                    empty_somewhere = self.grad_sync.some_rank_empty(len(rows))
                    rows = self.grad_sync.shard(rows)
                    if len(rows) == 0:
                        self.history["train_loss"].append(self._dp_step(None, 0.0, opt, True))
                        continue
                rows = np.asarray(rows, dtype=np.int64)
                rows_dev = rows_to_device(rows, self.device)
                t = BatchTiles(xtr, rows_dev, want_lib=True, rows_host=rows)
                c = None if ctr is None else ctr[rows_dev]
                x = t.to_dense() if dense_in else t
                total, _, _ = _vae_losses(x, self._lib(t), enc, dec, self.kl_weight, self.distribution, c is not None, c)
                if self.grad_sync is not None:
                    self.history["train_loss"].append(self._dp_step(total, float(len(rows)), opt, empty_somewhere))
                    continue
                opt.zero_grad()
                total.backward()
                opt.step()
                self.history["train_loss"].append(total.detach())
            vt, vr, vk = [float(v) for v in self.evaluate(xva, cva)]
            self.history["val_loss"].append(vt); self.history["val_recon"].append(vr); self.history["val_kl"].append(vk)
            self._say(f"[VAE] epoch {epoch + 1} validation total {vt:.6f} (reconstruction {vr:.6f}, KL {vk:.6f})")
            if vt < min_val:
                no_improve, min_val = 0, vt
                best = (_snapshot(enc, None if best is None else best[0]), _snapshot(dec, None if best is None else best[1]))
            else:
                no_improve += 1
            if epoch > 1 and no_improve == patience:
                self._say("Early Stopped!")
                break
        self.history["train_loss"] = [float(v) for v in torch.stack(self.history["train_loss"]).cpu()] \
            if self.history["train_loss"] else []
        self.last_encoder, self.last_decoder = enc, dec
        self.encoder, self.decoder = best[0].eval(), best[1].eval()
        if self.save_model:
            path = self.save_path or os.getcwd()
            torch.save(self.encoder, os.path.join(path, "vae_encoder.pt"))
            torch.save(self.decoder, os.path.join(path, "vae_decoder.pt"))
        return self.encoder, self.decoder

    def _dp_step(self, loss, n_valid, opt, some_rank_empty):
        opt.zero_grad()
        if loss is not None:
            loss.backward()
        params = [p for g in opt.param_groups for p in g["params"]]
        gl = self.grad_sync.exchange(params, n_valid, loss, sink=opt._dp_sink, some_rank_empty=some_rank_empty)
        opt.step()
        return gl

    @torch.no_grad()
    def embeddings(self, mtx, cov=None):
        """Posterior means mu of every cell (the VAE's embedding), in chunks."""
        from .model import _to_csr
        csr = _to_csr(mtx, self.device)
        cov = None if cov is None else cov.to(self.device).float()
        self.encoder.eval()
        out = []
        for r0 in range(0, csr.n_rows, self.val_chunk):
            rows = np.arange(r0, min(csr.n_rows, r0 + self.val_chunk))
            mu = self.encoder(BatchTiles(csr, rows, want_lib=False), None if cov is None else cov[r0:r0 + len(rows)])[0]
            out.append(mu)
        return torch.cat(out, 0)
