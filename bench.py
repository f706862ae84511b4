#!/usr/bin/env python3
"""Headline benchmark: training cells/sec, paired RNA+ATAC (BASELINE.json metric).

One "step" = one minibatch of paired cells pushed through BOTH directions (RNA->ATAC with the Bernoulli decoder,
ATAC->RNA with the NB decoder): tile-list build, encoder forward, fused decoder+likelihood+backward, encoder
backward, fused AdamW -- everything the reference's train_net does per batch (utils.py:118-148, model.py:727-763),
inputs already resident in HBM as CSR.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL gradient all-reduce, weak scaling:
   every rank keeps the configured per-GPU batch and its own shard of cells)

Rank 0 prints ONE JSON line with `roofline` (dominant kernel, live HIP-event timing) and `cpu_baseline`
(the CPU oracle restatement timed on the host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--cells", type=int, default=100_000)
    ap.add_argument("--genes", type=int, default=20_000)
    ap.add_argument("--peaks", type=int, default=100_000)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--hidden", type=int, nargs=2, default=[900, 40])
    ap.add_argument("--precision", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--p-atac", type=float, default=0.03)
    ap.add_argument("--rna-density", type=float, default=0.10)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--cpu-steps", type=int, default=3, help="paired steps of the CPU baseline (0 = skip)")
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--streams", type=int, default=2, choices=[1, 2],
                    help="2: the two directions (independent trainings, model.py:608) run on two HIP streams")
    ap.add_argument("--fp32-exchange", action="store_true",
                    help="N > 1: all-reduce the wide first-layer gradients in fp32 even in bf16 mode (default: bf16)")
    ap.add_argument("--no-fuse-opt", action="store_true",
                    help="keep the AdamW step of the first-layer weights in the optimiser kernel (always so for N > 1)")
    ap.add_argument("--no-grad-sink", action="store_true",
                    help="N > 1, bf16 exchange: keep fp32 first-layer gradients + cast/copy instead of the bf16 exchange buffer")
    ap.add_argument("--zinb", action="store_true", help="C3: zero-inflated NB for RNA (the G x G dropout layer is live)")
    ap.add_argument("--graph", action="store_true",
                    help="N = 1: replay the step as one HIP graph (scpair_amd.graph.CapturedStep) instead of issuing every "
                         "launch from the host.  Off by default: on ROCm 7.2 a graph with two forked streams is replayed "
                         "node by node at the eager host cost, and the linear (one stream) replay took a box down")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="build each minibatch's tile lists at the start of its own step instead of during the previous one")
    ap.add_argument("--roofline-steps", type=int, default=20,
                    help="untimed single-stream steps after the timed region whose per-launch HIP events give roofline.achieved")
    return ap.parse_args()


class Direction:
    """One encoder -> decoder predictor with its optimiser, built like scPair_object._build_predictor."""

    def __init__(self, S, in_dim, out_dim, hidden, out_dist, infer_lib, sample_factor, dropout, dev, seed=0):
        from torch import nn
        S.set_seed(seed)
        self.enc = S.Input_Module(input_dim=in_dim, input_batch_num=0, hidden_layer=hidden, activation=nn.LeakyReLU(),
                                  layernorm=True, batchnorm=False, dropout_rate=dropout, add_linear_layer=True,
                                  infer_library_size=infer_lib)
        S.set_seed(seed)
        if out_dist == "ber":
            self.dec = S.Output_Module_Ber(output_dim=out_dim, output_batch_num=0, hidden_layer=hidden,
                                           infer_library_size=infer_lib, sample_factor=sample_factor, feature_factor=False)
        else:
            self.dec = _raw_decoder(S, out_dim, hidden, sample_factor, zinb=(out_dist == "zinb"))
        self.enc.apply(S.init_weights).to(dev)
        self.dec.apply(S.init_weights).to(dev)
        de, ne, _, _ = S.add_weight_decay(self.enc, output_layer=["output"])
        dd, nd, _, _ = S.add_weight_decay(self.dec, output_layer=["output"])
        groups = [{"params": de, "weight_decay": 1e-9, "lr": 1e-3}, {"params": dd, "weight_decay": 1e-9, "lr": 1e-3},
                  {"params": ne, "weight_decay": 0, "lr": 1e-3}, {"params": nd, "weight_decay": 0, "lr": 1e-3}]
        self.opt = S.FusedAdamW([g for g in groups if g["params"]])
        self.first = [self.enc.hidden_rep[0].weight] + ([self.enc.hidden_lib[0][0].weight] if infer_lib else [])
        self.dist = out_dist
        self.fused = False
        self.sink = None
        self.use_lib = sample_factor
        self.enc.train(); self.dec.train()

    def n_trained_params(self):
        """Parameters that received an update in the last step (dead layers get no gradient and are skipped)."""
        extra = sum(p.numel() for p in self.first) if self.fused else 0
        return sum(self.opt.last_step_numel) + extra

    def adamw_kernel_bytes(self):
        """Algorithmic bytes of this direction's multi-tensor AdamW launch: 28 B/param, 26 where the
        gradient is read as bf16 from the exchange buffer; parameters updated inside their dW kernel are not in it."""
        n32, n16 = self.opt.last_step_numel
        return 28.0 * n32 + 26.0 * n16


def _raw_decoder(S, out_dim, hidden, sample_factor, zinb=False):
    # zero_inflated=False ('nb' headline config).  The reference also allocates an unused Linear(G,G)
    # `output_inverse_dispersion` (model.py:224-225, 1.6 GB at G=20k, never touched by forward/backward); it is kept
    # so the state_dict stays interchangeable.
    return S.Output_Module_Raw(output_dim=out_dim, output_batch_num=0, hidden_layer=hidden, infer_library_size=not sample_factor,
                               sample_factor=sample_factor, feature_factor=False, zero_inflated=zinb,
                               dispersion="feature-cell")


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus and world > 1:
        a.gpus = world
    # rehearsal knob: SCPAIR_BENCH_BACKEND=gloo runs the N>1 path with every rank on the visible GPU(s) of a
    # one-GPU box (RCCL refuses two ranks on one device); the driver's multi-GPU runs use nccl (= RCCL)
    backend = os.environ.get("SCPAIR_BENCH_BACKEND", "nccl")
    if os.environ.get("SCPAIR_BENCH_WATCHDOG"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["SCPAIR_BENCH_WATCHDOG"]), exit=True)
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    # SCPAIR_BENCH_FORCE_DIST=1 (under torchrun --nproc-per-node 1): run the exchange path with one rank, which is
    # all of the RCCL call sequence a one-GPU box can exercise
    use_dist = world > 1 or bool(os.environ.get("SCPAIR_BENCH_FORCE_DIST"))
    if use_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    import scpair_amd as S
    from scpair_amd import synth
    S.set_precision(a.precision)

    # ---- synthetic inputs, generated straight into HBM (per-rank shard of cells) ---------------------------
    cells_here = max(a.batch * 4, a.cells // world)
    rna = synth.make_rna_device(cells_here, a.genes, dev, seed=100 + rank, target_density=a.rna_density)
    atac = synth.make_atac_device(cells_here, a.peaks, dev, seed=200 + rank, p=a.p_atac)
    nnz_r = rna.nnz / cells_here
    nnz_a = atac.nnz / cells_here

    # ---- the two predictors (reference defaults: model.py:365-372) ---------------------------------------------
    A = Direction(S, a.genes, a.peaks, a.hidden, "ber", infer_lib=True, sample_factor=False, dropout=a.dropout, dev=dev)
    Bd = Direction(S, a.peaks, a.genes, a.hidden, "zinb" if a.zinb else "nb", infer_lib=False, sample_factor=True, dropout=a.dropout, dev=dev)
    sync = None
    if use_dist:
        from scpair_amd.distributed import GradSync
        sync = GradSync(compress=(a.precision == "bf16" and not a.fp32_exchange))
        for d_ in (A, Bd):
            sync.broadcast(list(d_.enc.parameters()) + list(d_.dec.parameters()))
            if sync.compress and not a.no_grad_sink:
                d_.sink = sync.make_sink(d_.first)     # dW of the wide first layers goes straight to the bf16 exchange buffer

    fuse = (sync is None) and (not a.no_fuse_opt) and a.precision == "bf16"
    if fuse:
        for d_ in (A, Bd):
            d_.opt.fuse_first_layer(*d_.first)
            d_.fused = True
            wide = getattr(d_.dec, "output_dropout", None)        # C3 (--zinb): the G x G zero-inflation layer
            if isinstance(wide, torch.nn.Linear):
                d_.opt.fuse_weight_update(wide.weight)

    rng = np.random.default_rng(1234 + rank)
    state = {"perm": rng.permutation(cells_here), "cur": 0}
    state["perm_dev"] = torch.from_numpy(state["perm"]).to(dev)
    adam_ev = []
    fused_ev = []

    host_t = {"tiles": 0.0, "fwd": 0.0, "bwd": 0.0, "opt": 0.0}      # host enqueue time by phase (diagnostic)
    pc = time.perf_counter

    def one_direction(D, tx, ty, time_adam):
        from scpair_amd.utils import _predictor_loss
        t0_ = pc()
        lib = ty.lib.reshape(-1, 1) if D.use_lib else None
        loss = _predictor_loss(tx, ty, lib, D.enc, D.dec, D.dist, False, None, None)
        t1_ = pc()
        D.opt.zero_grad()
        S.ops.backward_root(loss)
        host_t["fwd"] += t1_ - t0_
        host_t["bwd"] += pc() - t1_
        params = [p for g in D.opt.param_groups for p in g["params"] if p.grad is not None]
        if sync is not None:
            sync.start_all_reduce([p.grad for p in params], sink=D.sink)
        return loss

    def finish_direction(D, time_adam):
        t0_ = pc()
        _finish_direction(D, time_adam)
        host_t["opt"] += pc() - t0_

    def _finish_direction(D, time_adam):
        if sync is not None:
            sync.finish()
        if time_adam:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            D.opt.step(grad_scale=1.0 / world, release_grads=True)
            e1.record()
            adam_ev.append((e0, e1, D.adamw_kernel_bytes()))
        else:
            D.opt.step(grad_scale=1.0 / world, release_grads=True)

    def next_rows():
        if state["cur"] + a.batch > cells_here:
            state["perm"], state["cur"] = rng.permutation(cells_here), 0
            state["perm_dev"] = torch.from_numpy(state["perm"]).to(dev)        # one upload per epoch
        rows = state["perm"][state["cur"]:state["cur"] + a.batch]
        rows_dev = state["perm_dev"][state["cur"]:state["cur"] + a.batch]
        state["cur"] += a.batch
        return rows, rows_dev

    def both_directions(t_rna, t_atac, time_adam, side, prefetch=None):
        if side is None:
            la = one_direction(A, t_rna, t_atac, time_adam)
            finish_direction(A, time_adam)
            lb = one_direction(Bd, t_atac, t_rna, time_adam)
            finish_direction(Bd, time_adam)
            return la, lb
        # the two predictors share nothing but the minibatch: one stream each, both after the tile-list build
        cur = torch.cuda.current_stream()
        for t in (t_rna, t_atac):
            for x in (t.ptr, t.key, t.val, t.pk, t.lib, t.nnan, t.rows):
                if x is not None:
                    for st in side:
                        x.record_stream(st)
        out = []
        for st, (D, tx, ty) in zip(side, ((A, t_rna, t_atac), (Bd, t_atac, t_rna))):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                out.append(one_direction(D, tx, ty, time_adam))
                finish_direction(D, time_adam)
        if prefetch is not None:
            prefetch()      # the next minibatch's tile lists are built on this (otherwise idle) stream meanwhile
        for st in side:
            cur.wait_stream(st)
        return out[0], out[1]

    def build_tiles():
        rows, rows_dev = next_rows()
        t0_ = pc()
        t_rna = S.BatchTiles(rna, rows_dev, want_lib=True, rows_host=rows)
        t_atac = S.BatchTiles(atac, rows_dev, want_lib=False, rows_host=rows)
        host_t["tiles"] += pc() - t0_
        return t_rna, t_atac

    def step(time_adam=False, side_=None):
        # Two streams: the tile lists of the NEXT minibatch (the data-loader stage: row ids -> tile lists, no weights
        # involved) are built while this step's two directions run, as a prefetching loader would.
        nxt = state.pop("next_tiles", None)
        t_rna, t_atac = nxt if nxt is not None else build_tiles()
        if side_ is None or a.no_prefetch:
            return both_directions(t_rna, t_atac, time_adam, side_)

        def prefetch():
            state["next_tiles"] = build_tiles()
        return both_directions(t_rna, t_atac, time_adam, side_, prefetch)

    side = (torch.cuda.Stream(), torch.cuda.Stream()) if a.streams == 2 else None

    # ---- the whole paired step as one HIP graph (N = 1): fixed-capacity tile lists, device-resident per-step words ----
    # Under rocprofv3 the step is issued eagerly: the kernels and their order are the same, and a graph replay under
    # the profiler's queue interception took a box down once this round (the only difference from runs that passed).
    under_profiler = any(k_ in os.environ for k_ in ("ROCPROFILER_LIBRARY_CTOR", "ROCPROF_OUTPUT_PATH"))
    use_graph = (sync is None) and a.graph and not under_profiler
    captured = None
    if use_graph:
        from scpair_amd.graph import CapturedStep
        cap_r, cap_a = rna.max_batch_nnz(a.batch), atac.max_batch_nnz(a.batch)

        def graph_fn(rows_buf):
            t_rna = S.BatchTiles(rna, rows_buf, want_lib=True, capacity=cap_r)
            t_atac = S.BatchTiles(atac, rows_buf, want_lib=False, capacity=cap_a)
            return both_directions(t_rna, t_atac, False, side)

        captured = CapturedStep(graph_fn, [A.opt, Bd.opt], a.batch, dev)

    def graph_step():
        _, rows_dev = next_rows()
        return captured(rows_dev)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(a.warmup, 2) if use_graph else a.warmup):
        step(side_=side)
    if use_graph:
        barrier()
        captured.capture(next_rows()[1])
        for _ in range(3):
            graph_step()
    barrier()
    prof = None
    if os.environ.get("SCPAIR_BENCH_CPROFILE") and rank == 0:      # diagnostic: where the host enqueue time goes
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    for k_ in host_t:
        host_t[k_] = 0.0
    t0 = time.perf_counter()
    if use_graph:
        for _ in range(a.steps):
            la, lb = graph_step()
    else:
        for _ in range(a.steps):
            la, lb = step(side_=side)
    t_host = time.perf_counter() - t0             # host enqueue time of the timed steps (diagnostic)
    if prof is not None:
        import pstats
        prof.disable()
        with open(os.environ["SCPAIR_BENCH_CPROFILE"], "w") as fh:
            pstats.Stats(prof, stream=fh).sort_stats(os.environ.get("SCPAIR_BENCH_CPROFILE_SORT", "cumulative")).print_stats(70)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    losses = (float(la.detach()), float(lb.detach()))
    # host cost of issuing ONE step into an idle queue (the run-ahead figure above includes time blocked on a full queue)
    t_issue = 0.0
    for _ in range(10):
        torch.cuda.synchronize()
        t1_ = time.perf_counter()
        if use_graph:
            graph_step()
        else:
            step(side_=side)
        t_issue += time.perf_counter() - t1_
    barrier()
    host_by_phase = {k_: round(1e3 * v_ / a.steps, 3) for k_, v_ in host_t.items()}

    # ---- roofline pass (untimed): the same step issued eagerly on ONE stream, HIP events around every launch of the
    # dominant kernel on that stream, so the durations hold no time queued behind the other direction's kernels and
    # agree with rocprofv3's per-kernel durations of this command (profiles/) ----
    if fuse:
        S.ops.set_fused_timer(fused_ev)
    for _ in range(a.roofline_steps if world == 1 else 0):
        step(time_adam=True, side_=None)
    S.ops.set_fused_timer(None)
    barrier()
    n_roof = max(a.roofline_steps, 1)

    # ---- roofline of the dominant kernel ------------------------------------------------------------------------
    # Two-kernel path (--no-fuse-opt, N > 1): the multi-tensor AdamW kernel, 28 B/param.
    # Fused path (default at N = 1): the first-layer weight-gradient kernel with the AdamW step in its epilogue; it
    # moves 24 B/param of optimiser state (p, m, v read and written; the gradient stays on chip) + 2 B/param of bf16
    # shadow + its operands (tile-list entries, bf16 dY^T).  Timed with events around each launch on its stream.
    tot_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in adam_ev)
    tot_bytes = sum(nb for _, _, nb in adam_ev)
    adam_gbs = tot_bytes / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0
    n_params = A.n_trained_params() + Bd.n_trained_params()
    adam_roof = {"kernel": "adamw_kernel (fused multi-tensor AdamW)", "bound": "hbm", "achieved": adam_gbs,
                 "peak": 8000.0, "unit": "GB/s", "frac": adam_gbs / 8000.0, "traffic": None,
                 "algorithmic_bytes_per_launch_set": tot_bytes / n_roof,
                 "avg_ms_per_step_in_kernel": tot_ms / n_roof}

    def pmc_traffic(name, want_bytes):
        # HBM bytes of the kernel from PMC counters (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate rocprofv3 passes over
        # this command; profiles/*.json).  Only quoted when it was collected for this exact workload.
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                tr = json.load(fh)
            ref_bytes = tr.get("algorithmic_bytes_per_step", tr.get("optimizer_state_bytes_per_step"))
            # same weights and precision; the per-batch operand bytes (a few MB) are allowed to differ
            if abs(ref_bytes - want_bytes) < 0.02 * want_bytes and a.precision == "bf16":
                return tr["hbm_bytes_per_step"]
        except (OSError, KeyError, ValueError):
            pass
        return None

    if fused_ev:
        f_ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in fused_ev)
        f_bytes = sum(26.0 * n + ob for _, _, n, ob in fused_ev)
        f_gbs = f_bytes / (f_ms * 1e-3) / 1e9 if f_ms > 0 else 0.0
        roof = {"kernel": "spmm_dw128_bf16_kernel / spmm_dw_bf16_kernel<*, ADAM> (first-layer weight gradient with the AdamW step in its epilogue; one launch per encoder)",
                "bound": "hbm", "achieved": f_gbs, "peak": 8000.0, "unit": "GB/s", "frac": f_gbs / 8000.0,
                "traffic": pmc_traffic("r02_dw_adamw_traffic.json", f_bytes / n_roof),
                "traffic_source": "stored: profiles/r02_dw_adamw_traffic.json (separate rocprofv3 --pmc passes over this command, round 2 kernels)",
                "algorithmic_bytes_per_launch_set": f_bytes / n_roof,
                "avg_ms_per_step_in_kernel": f_ms / n_roof,
                "launches_per_step": len(fused_ev) / n_roof,
                "timing": f"HIP events around each launch, {n_roof} eager single-stream steps after the timed region",
                "other_params": adam_roof}
    else:
        if A.sink is None:      # the counters were collected for fp32 gradients
            adam_roof["traffic"] = pmc_traffic("r01_adamw_traffic.json", 28.0 * n_params)
        roof = adam_roof

    out = {
        "metric": "training cells/sec (whole node) paired RNA+ATAC",
        "value": a.batch * world * a.steps / dt,
        "unit": "paired cells/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16 MFMA inputs, f32 accumulate/weights" if a.precision == "bf16" else "f32 (3x bf16 split MFMA)",
        "data": "synthetic",
        "config": {"workload": f"{a.workload}: {a.cells} cells x {a.genes} genes (NB) x {a.peaks} peaks (Bernoulli), "
                               f"batch {a.batch}/GPU, hidden {a.hidden}, dropout {a.dropout}, both directions per step",
                   "cells_resident_per_gpu": cells_here, "nnz_per_cell_rna": round(nnz_r, 1),
                   "nnz_per_cell_atac": round(nnz_a, 1), "precision": a.precision, "parallelism": f"dp{world}", "streams": a.streams,
                   "grad_exchange": ("none" if sync is None else (("bf16 wide (written by the dW kernel) + fp32 rest" if A.sink is not None
                                                                    else "bf16 wide + fp32 rest") if sync.compress else "fp32")),
                   "optimizer": "AdamW; first-layer weights updated in their weight-gradient kernel" if fuse else "AdamW, one multi-tensor kernel",
                   "host_enqueue_ms_per_step": round(1e3 * t_issue / 10, 3),
                   "host_loop_ms_per_step": round(1e3 * t_host / a.steps, 3),
                   "host_enqueue_ms_by_phase": host_by_phase,
                   "step_issue": "HIP graph replay (rows copy + per-step words + hipGraphLaunch)" if use_graph else ("eager launches (profiler attached)" if under_profiler else "eager launches"),
                   "trained_params": n_params, "final_losses": losses},
        "roofline": roof,
    }

    if rank == 0 and world == 1 and a.cpu_steps > 0:
        out["cpu_baseline"] = cpu_baseline(a, S, A, Bd, rna, atac, rng)
    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


def cpu_baseline(a, S, A, Bd, rna, atac, rng):
    """The oracle (oracle/ref_cpu.py: dense fp32 torch CPU ops, same architecture/losses/AdamW grouping) timed on the
    host cores for `cpu_steps` paired steps of the same shapes, starting from the same weights."""
    from oracle import ref_cpu as O
    ncores = int(os.environ.get("OMP_NUM_THREADS", 0)) or min(16, os.cpu_count() or 1)
    torch.set_num_threads(ncores)
    cpu = lambda m: {k: v.detach().cpu() for k, v in m.state_dict().items()
                     if not k.startswith("output_inverse_dispersion")}     # dead G x G layer: no grads, skip the copy
    dA = O.Direction(cpu(A.enc), cpu(A.dec), "ber", dict(sample_factor=False))
    dB = O.Direction(cpu(Bd.enc), cpu(Bd.dec), "nb", dict(sample_factor=True))
    batches = []
    for _ in range(a.cpu_steps + 1):
        rows = np.sort(rng.choice(rna.n_rows, size=a.batch, replace=False))
        xr = S.BatchTiles(rna, rows).to_dense().cpu()
        xa = S.BatchTiles(atac, rows).to_dense().cpu()
        batches.append((xr, xa))
    xr, xa = batches[0]
    dA.train_step(xr, xa, None); dB.train_step(xa, xr, xr.sum(1, keepdim=True))          # warm-up
    t0 = time.perf_counter()
    for xr, xa in batches[1:]:
        dA.train_step(xr, xa, None)
        dB.train_step(xa, xr, xr.sum(1, keepdim=True))
    dt = time.perf_counter() - t0
    return {"value": a.batch * a.cpu_steps / dt, "unit": "paired cells/s", "cores": ncores, "kind": "port",
            "sample": f"{a.cpu_steps} paired steps of batch {a.batch} at the same shapes (after 1 warm-up step), "
                      f"dense fp32 torch CPU ops, dropout 0"}


if __name__ == "__main__":
    main()
