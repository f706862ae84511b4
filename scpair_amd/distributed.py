"""Cell-sharded data parallelism: one process per GPU, gradients summed with a bucketed all-reduce
(torch.distributed backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for the tests).

The reference has no distributed code; the semantics implemented here are those of its
single-process loop with the GLOBAL batch: every loss is a mean over label elements (loss.py:26, 39-40, 48, 50), so
the global-batch gradient is sum_r (n_r / n_total) * g_r where n_r is rank r's number of valid label elements.

Buckets: tensors >= `big` elements (the wide first-layer dW, produced last in backward, the only exposed message)
travel in one flat buffer (bf16 when `compress`), everything smaller in one fp32 buffer: two collectives per exchange,
issued async and waited together, so they overlap with whatever the caller runs in between.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .sparse import scalar64 as _scalar64


class GradSync:
    def __init__(self, group=None, big: int = 1 << 20, compress: bool = False):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.big = big
        # compress: tensors >= `big` elements (the wide first-layer dW, > 95 % of the bytes) travel as bf16.  Meant
        # for precision mode 'bf16', where those gradients are products of bf16-rounded operands anyway; it halves
        # the per-step message (524 MB -> 272 MB at C2), which is what the point-to-point xGMI links are short of.
        self.compress = compress
        self._pending = []
        self._backend = dist.get_backend(group)
        from . import ops
        ops.set_seed_rank(self.rank)          # dropout masks differ between the shards of a global batch

    # -- batch sharding --------------------------------------------------------------------------------------
    def shard(self, rows):
        """This rank's contiguous slice of a global batch (all ranks draw the same global batch order)."""
        rows = np.asarray(rows)
        return np.array_split(rows, self.world)[self.rank]

    # -- collectives -----------------------------------------------------------------------------------------
    def _sum(self, t):
        """Blocking SUM all-reduce of a small tensor, in place.  Backends other than nccl get device tensors through
        host memory (gloo's own device all-reduce stalls on this image; gloo + GPU is the one-GPU rehearsal only)."""
        if self._backend != "nccl" and t.is_cuda:
            h = t.detach().cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def gather_rows(self, t):
        """All-gather of one 1-D tensor per rank -> [world, n] in rank order (the same on every rank).  Same host
        detour as _sum for backends other than nccl."""
        src = t.detach()
        if self._backend == "nccl" and src.is_cuda:
            out = torch.empty((self.world, src.numel()), dtype=src.dtype, device=src.device)
            dist.all_gather_into_tensor(out, src.contiguous().view(-1), group=self.group)
            return out
        if self._backend != "nccl" and src.is_cuda:
            src = src.cpu()
        parts = [torch.empty_like(src) for _ in range(self.world)]
        dist.all_gather(parts, src.contiguous(), group=self.group)
        return torch.stack(parts).to(t.device)

    def reduce_scatter_rows(self, t):
        """t: [world * w, n] partial sums, rank-major row blocks.  Returns the [w, n] block of this rank summed over the
        ranks: a reduce-scatter on nccl (half the bytes of an all-reduce), an all-reduce + slice elsewhere (gloo has no
        reduce-scatter; it is the rehearsal backend only)."""
        w = t.shape[0] // self.world
        if self._backend == "nccl":
            out = torch.empty((w, t.shape[1]), dtype=t.dtype, device=t.device)
            dist.reduce_scatter_tensor(out, t.contiguous(), op=dist.ReduceOp.SUM, group=self.group)
            return out
        self._sum(t)
        return t[self.rank * w:(self.rank + 1) * w]

    def share_of(self, n_valid, device):
        """This rank's share n_r / n_total of a step's valid label elements (one small SUM all-reduce): the weight its
        loss carries in the global-batch loss.  A device scalar."""
        nv = _scalar64(n_valid, device).reshape(1).clone()
        tot = self._sum(nv.clone())
        return (nv / torch.clamp(tot, min=1.0)).to(torch.float32).reshape(())      # no valid label anywhere: shares 0

    def broadcast(self, tensors, src=0):
        for t in tensors:
            t = t.data if hasattr(t, "data") else t
            if not t.is_contiguous():                      # row-padded first-layer weights (ops.home_first_layers)
                c = t.contiguous()
                self.broadcast([c], src=src)
                t.copy_(c)
                continue
            if self._backend != "nccl" and t.is_cuda:      # staged through host like the other gloo + GPU collectives
                h = t.detach().cpu()
                dist.broadcast(h, src=src, group=self.group)
                t.copy_(h)
            else:
                dist.broadcast(t, src=src, group=self.group)

    def make_sink(self, params):
        """One flat bf16 exchange buffer for the wide first-layer weights `params` (precision 'bf16'): their
        weight-gradient kernels write dW into it as bf16 (ops.GradSink), start_all_reduce(..., sink=...) sums it in
        place with a single collective and FusedAdamW.step reads the gradients from it.  Replaces, per step, the
        fp32 dW write, the cast into a bf16 bucket and the copy back (≈1.9 GB of HBM traffic at C2)."""
        from . import ops
        sizes = [p.numel() for p in params]
        flat = torch.zeros(sum(sizes), dtype=torch.bfloat16, device=params[0].device)
        handles = [ops.set_grad_sink(p, v.view(p.shape)) for p, v in zip(params, flat.split(sizes))]
        return flat, handles

    def start_all_reduce(self, grads, sink=None):
        """Issue async SUM all-reduces for `grads` (list of tensors); returns nothing, call finish().

        Two buckets per call, each one collective: the wide tensors (>= `big` elements) packed into one flat buffer
        (bf16 when `compress`), everything smaller packed into one fp32 buffer.  Packing and unpacking are one
        multi-tensor copy each, so the host cost of an exchange does not grow with the number of tensors."""
        sink_flat = None
        if sink is not None and any(h.written for h in sink[1]):
            sink_flat = sink[0]
        if self._backend != "nccl" and (any(g.is_cuda for g in grads) or (sink_flat is not None and sink_flat.is_cuda)):
            # gloo with device tensors (one-GPU rehearsal of the N>1 path only): stage through host memory; gloo's
            # own device all-reduce stalls on this image
            for g in list(grads) + ([sink_flat] if sink_flat is not None else []):
                h = g.detach().cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                g.copy_(h)
            return
        if sink_flat is not None:
            work = dist.all_reduce(sink_flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._pending.append((work, None, None))
        wide = [g for g in grads if g.numel() >= self.big]
        small = [g for g in grads if g.numel() < self.big]
        for bucket, dtype in ((wide, torch.bfloat16 if self.compress else torch.float32), (small, torch.float32)):
            if not bucket:
                continue
            if len(bucket) == 1 and bucket[0].dtype == dtype and bucket[0].is_contiguous():
                work = dist.all_reduce(bucket[0], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self._pending.append((work, None, None))
                continue
            sizes = [g.numel() for g in bucket]
            flat = torch.empty(sum(sizes), dtype=dtype, device=bucket[0].device)
            torch._foreach_copy_(list(flat.split(sizes)), [g.reshape(-1) for g in bucket])
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._pending.append((work, flat, bucket))

    def finish(self):
        for work, flat, bucket in self._pending:
            if work is not None:
                work.wait()
            if flat is not None:
                parts = flat.split([g.numel() for g in bucket])
                torch._foreach_copy_(bucket, [p.view_as(g) for p, g in zip(parts, bucket)])
        self._pending = []

    def all_reduce_grads(self, params, n_valid):
        """Global-batch gradient: each rank scales its mean-gradient by n_r / n_total, then SUM all-reduce."""
        dev = params[0].device
        nv = _scalar64(n_valid, dev).reshape(1).clone()
        tot = self._sum(nv.clone())
        w = (nv / torch.clamp(tot, min=1.0)).to(torch.float32)
        grads = [p.grad for p in params if p.grad is not None]
        if grads:
            torch._foreach_mul_(grads, w.reshape(()))
        self.start_all_reduce(grads)
        self.finish()

    # -- one optimisation step's exchange (what the trainers call) ------------------------------------------------
    def some_rank_empty(self, n_global_rows: int) -> bool:
        """True when shard() leaves at least one rank without rows (global batch smaller than the world); every rank
        computes the same answer from the global batch size."""
        return int(n_global_rows) < self.world

    def _align_grads(self, params, sink):
        """Ranks with an empty shard ran no backward and hold no gradients: agree (MAX all-reduce of a presence mask) on
        which parameters carry one and supply zeros where this rank has none, so every rank enters the same
        collectives with the same buckets."""
        sink_ids = {} if sink is None else {id(h.param): h for h in sink[1]}
        have = [1.0 if (p.grad is not None or (id(p) in sink_ids and sink_ids[id(p)].written)) else 0.0 for p in params]
        dev = params[0].device if self._backend == "nccl" else "cpu"
        mask = torch.tensor(have, dtype=torch.float32, device=dev)
        dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=self.group)
        mask = mask.cpu().tolist()
        zero_sink = False
        for p, f, h in zip(params, mask, have):
            if f > 0 and h == 0:
                if id(p) in sink_ids:
                    zero_sink = True
                else:
                    p.grad = torch.zeros_like(p)
        if zero_sink:
            sink[0].zero_()
            for h_ in sink[1]:
                h_.written = True

    def step_scalars(self, n_valid, loss, device):
        """The step's one small collective: SUM over the ranks of (valid label elements, loss x valid label elements).
        Returns (this rank's count, the summed pair) as float64 device tensors.  `loss` None: this rank's shard was empty."""
        nv = _scalar64(n_valid, device)
        lv = torch.zeros((), dtype=torch.float64, device=device) if loss is None else loss.detach().double().reshape(())
        return nv, self._sum(torch.stack([nv, lv * nv]))

    @staticmethod
    def share_from(scalars):
        """n_r / n_total (float32 device scalar) from step_scalars()'s result; 0 when no rank has a valid label."""
        nv, v = scalars
        return (nv / torch.clamp(v[0], min=1.0)).to(torch.float32).reshape(())

    def exchange(self, params, n_valid, loss, sink=None, some_rank_empty=False, scalars=None):
        """After backward() on this rank's shard: one small collective for the step's scalars (this rank's number of
        valid label elements and its loss; `scalars`: the result of step_scalars() when the caller needed it before the
        backward pass already), gradients scaled by n_r / n_total in one multi-tensor launch, the bucketed
        SUM all-reduce.  `loss` None / n_valid 0: this rank's shard was empty.  Returns the global-batch loss."""
        dev = params[0].device
        nv, v = scalars if scalars is not None else self.step_scalars(n_valid, loss, dev)
        w = (nv / torch.clamp(v[0], min=1.0)).to(torch.float32)
        if some_rank_empty:
            self._align_grads(params, sink)
        grads = [p.grad for p in params if p.grad is not None]
        if grads:
            torch._foreach_mul_(grads, w)
        if sink is not None and any(h.written for h in sink[1]):
            sink[0].mul_(w.to(sink[0].dtype))      # exact for equal shards on 2^k ranks (w = 1 / world)
        self.start_all_reduce(grads, sink=sink)
        self.finish()
        return (v[1] / torch.clamp(v[0], min=1.0)).to(torch.float32)

    def weighted_mean(self, value, n_valid):
        dev = value.device
        nv = _scalar64(n_valid, dev)
        v = torch.stack([value.double().reshape(()) * nv, nv])
        self._sum(v)
        return (v[0] / v[1]).to(value.dtype)

    def reduce_scalars(self, num, den):
        v = self._sum(torch.stack([num.double(), den.double()]))
        return v[0], v[1]


# ---------------------------------------------------------------------------------------------------------
# Feature-sharded first layer (prototype of DESIGN.md section 7): every rank owns a COLUMN block W[:, f0:f1) of a
# wide first-layer weight (and its AdamW state), instead of a full replica whose gradient is all-reduced.
#   forward : every rank multiplies the GLOBAL batch's columns [f0, f1) with its block -> partial Y [B_global, H];
#             one SUM all-reduce of Y (B_global x H floats: 3.7 MB per 1024 cells at H = 900) gives the pre-activations;
#             each rank keeps the rows of its own cell shard.
#   backward: each rank's dY rows, weighted by its share n_r / n_total of the step's label elements, are all-gathered
#             into dY of the global batch; dW of the own column block = dY_global^T X[:, f0:f1) is then complete on the
#             rank: no gradient exchange for the wide weights (C2: 272-524 MB per direction and step become 2 x 3.7 MB
#             per 1024 cells), and the optimiser traffic and state of those weights divide by the number of ranks.
# The products themselves are callables supplied by the caller: ops.fcnn_layer / the bf16 tile-list kernels on the GPU
# (tests/test_dp_trainer_gpu.py), plain matmuls in the world-size-2 gloo test of the algebra on the CPU -- this module
# holds no compute path of its own.
# ---------------------------------------------------------------------------------------------------------
def shard_bounds(n: int, world: int):
    """[lo, hi) of every rank's slice under np.array_split(range(n), world) (GradSync.shard's rule)."""
    base, extra = divmod(n, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


class _FeatureShardedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, W_block, x_cols, sync, local_fwd, local_dw, share):
        y = local_fwd(x_cols, W_block)                      # [B_global, H]: this rank's columns only
        y = sync._sum(y.contiguous())
        B = y.shape[0]
        lo, hi = shard_bounds(B, sync.world)[sync.rank]
        ctx.x_cols, ctx.sync, ctx.local_dw, ctx.share, ctx.B = x_cols, sync, local_dw, share, B
        ctx.wshape = tuple(W_block.shape)
        return y[lo:hi].clone()

    @staticmethod
    def backward(ctx, dy):
        sync, B = ctx.sync, ctx.B
        bounds = shard_bounds(B, sync.world)
        width = max(hi - lo for lo, hi in bounds)
        mine = torch.zeros((width, dy.shape[1]), dtype=dy.dtype, device=dy.device)
        lo, hi = bounds[sync.rank]
        mine[:hi - lo] = dy * ctx.share                    # this rank's loss enters the global loss with weight n_r / n_total
        parts = sync.gather_rows(mine.reshape(-1)).reshape(sync.world, width, dy.shape[1])
        dy_g = torch.cat([parts[r, :h - l] for r, (l, h) in enumerate(bounds)], dim=0)
        dW = ctx.local_dw(ctx.x_cols, dy_g, ctx.wshape)     # complete gradient of the own column block
        return dW, None, None, None, None, None


def feature_sharded_first_layer(W_block, x_cols, sync: GradSync, local_fwd, local_dw, share):
    """Rows [lo, hi) (this rank's cell shard) of X @ W^T for the global batch, W split by columns over the ranks.

    W_block : this rank's [H, F_r] parameter; its .grad after backward() is the gradient of the GLOBAL-batch loss
              (do not hand it to GradSync.exchange: nothing is left to sum, and the share is already applied)
    x_cols  : the global batch restricted to this rank's columns, in whatever form the callables take
    local_fwd(x_cols, W_block) -> [B_global, H];  local_dw(x_cols, dY_global, W_block.shape) -> [H, F_r]
    share   : this rank's n_r / n_total of the step's valid label elements (GradSync.share_of)"""
    return _FeatureShardedFn.apply(W_block, x_cols, sync, local_fwd, local_dw, share)


# ---------------------------------------------------------------------------------------------------------
# The feature-sharded first layer wired to the fused encoder path (ops.EncBranchesFn) and the trainers
# ---------------------------------------------------------------------------------------------------------
class _StepShard:
    """What ops.EncBranchesFn reads off a minibatch's tile list (`tiles.shard`) in feature-sharded training."""

    def __init__(self, sync, width, n_local, share):
        self.sync, self.rank, self.world = sync, sync.rank, sync.world
        self.width, self.n_local, self.share = int(width), int(n_local), share


class FirstLayerShard:
    """Feature-sharded data parallelism for the wide first layer of an encoder.

    Every rank owns the column block W[:, lo:hi) of the encoder's first-layer weight(s) with its AdamW state, and holds
    the columns [lo, hi) of ALL cells of a split (DeviceCSR.col_block) instead of all columns of its own cells.  Per step:
      forward  : the rank multiplies the GLOBAL batch's columns with its block; ONE all-reduce of the pre-activations
                 (B_global x H floats) replaces the all-reduce of the dense weight gradient (H x F);
      backward : the ranks' dY rows (weighted by their share of the step's label elements) are all-gathered; the
                 block's gradient over the global batch is then complete on its rank, so its AdamW step can run inside the
                 weight-gradient kernel again (K2'), on 1 / world of the optimiser state and traffic.
    Everything behind the first layer stays cell-sharded data parallel (GradSync.exchange of the small tensors)."""

    def __init__(self, sync: GradSync, n_features: int):
        self.sync, self.rank, self.world = sync, sync.rank, sync.world
        self.F = int(n_features)
        align = 128 if self.F >= self.world * 1024 else 1      # whole feature tiles of the weight-gradient kernel
        per = -(-self.F // self.world)
        per = -(-per // align) * align
        self.bounds = [(min(self.F, r * per), min(self.F, (r + 1) * per)) for r in range(self.world)]
        if any(hi <= lo for lo, hi in self.bounds):
            raise ValueError(f"FirstLayerShard: {self.F} features cannot be split over {self.world} ranks")
        self.lo, self.hi = self.bounds[self.rank]
        self._sharded = []

    # -- modules ---------------------------------------------------------------------------------------------
    @staticmethod
    def _first_linears(encoder, names):
        out = []
        for seq in (getattr(encoder, n_, None) for n_ in names):
            while isinstance(seq, torch.nn.Sequential) and len(seq) > 0:
                seq = seq[0]
            if isinstance(seq, torch.nn.Linear):
                out.append(seq)
        return out

    def shard_module(self, encoder, names=("hidden_rep", "hidden_lib", "enc_hidden", "enc_lib")):
        """Replace the first-layer weights of `encoder` (already identical on every rank) by this rank's column block.
        Call before the optimiser is built.  Returns the new parameters."""
        new = []
        for lin in self._first_linears(encoder, names):
            if lin.weight.shape[1] != self.F:
                raise ValueError("FirstLayerShard: the first layer takes covariates or another feature count")
            with torch.no_grad():
                blk = lin.weight.detach()[:, self.lo:self.hi].clone()
            lin.weight = torch.nn.Parameter(blk)
            lin.in_features = self.hi - self.lo
            lin.weight._scp_feature_sharded = True
            new.append(lin.weight)
        return new

    def unshard_module(self, encoder, names=("hidden_rep", "hidden_lib", "enc_hidden", "enc_lib")):
        """The inverse: every rank gets the full first-layer weights back (all-gather of the blocks), e.g. on the
        checkpoint copy that the trainer returns.  Idempotent."""
        for lin in self._first_linears(encoder, names):
            if lin.weight.shape[1] == self.F and not getattr(lin.weight, "_scp_feature_sharded", False):
                continue
            w = lin.weight.detach()
            H = w.shape[0]
            per = max(hi - lo for lo, hi in self.bounds)
            mine = torch.zeros((H, per), dtype=w.dtype, device=w.device)
            mine[:, :self.hi - self.lo] = w
            parts = self.sync.gather_rows(mine.reshape(-1)).view(self.world, H, per)
            full = torch.cat([parts[r, :, :hi - lo] for r, (lo, hi) in enumerate(self.bounds)], dim=1).contiguous()
            lin.weight = torch.nn.Parameter(full, requires_grad=lin.weight.requires_grad)
            lin.in_features = self.F
        return encoder

    # -- data ------------------------------------------------------------------------------------------------
    def col_csr(self, csr):
        """This rank's column block of a split (all cells), with the padding row the batches' slices are padded with."""
        return csr.col_block(self.lo, self.hi, pad_row=True)

    def batch_tiles(self, col_csr, rows_global, share=None, rows_dev=None):
        """Tile list of a GLOBAL batch (row ids into the split; every rank passes the same ids) over this rank's columns.
        The ranks' slices (GradSync.shard's rule) are padded to a common multiple of 64 rows with the empty padding row, so
        cell blocks never straddle ranks.  share: this rank's n_r / n_total of the step's label elements (device scalar;
        None for forward-only use).  rows_dev: the same ids as an int64 device tensor (saves the upload when the slices
        need no padding).  Returns (tiles, this rank's row ids)."""
        from .sparse import BatchTiles
        rows_global = np.asarray(rows_global, dtype=np.int64)
        n = len(rows_global)
        if rows_dev is not None and n % (64 * self.world) == 0:
            w = n // self.world
            t = BatchTiles(col_csr, rows_dev, want_lib=False, rows_host=rows_global)
            t.shard = _StepShard(self.sync, w, w, share)
            return t, rows_global[self.rank * w:(self.rank + 1) * w]
        parts = np.array_split(rows_global, self.world)
        if min(len(p) for p in parts) == 0:
            raise ValueError("feature-sharded first layer: every rank needs at least one row of each global batch")
        width = (max(len(p) for p in parts) + 63) // 64 * 64
        pad_id = col_csr.n_rows - 1
        padded = np.full(self.world * width, pad_id, dtype=np.int64)
        for r, p in enumerate(parts):
            padded[r * width:r * width + len(p)] = p
        t = BatchTiles(col_csr, padded, want_lib=False)
        t.shard = _StepShard(self.sync, width, len(parts[self.rank]), share)
        return t, parts[self.rank]
