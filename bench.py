#!/usr/bin/env python3
"""Headline benchmark: training cells/sec (BASELINE.json metric), one line per run for any BASELINE config.

  python bench.py --gpus N --steps K --warmup W [--workload C2|C3|C4|C5]

Workloads (BASELINE.json `configs`; synthetic inputs of those shapes, generated straight into HBM as CSR):
  C2  100 k cells x 20 k genes (NB) x 100 k peaks (Bernoulli), batch 1024, bf16 MFMA          <- the default, `configs[1]`
  C3  C2 with the ZINB likelihood (the G x G zero-inflation layer is live) + the latent-mapping stage
  C4  unimodal VAE pretraining, 1 M cells x 20 k genes (NB), hidden [900], latent 40, batch 1024 per GPU
  C5  1 M paired cells, 250 k peaks, batch 4096 per GPU
One "step" = one minibatch pushed through everything the reference's train_net / train_VAE does per batch
(utils.py:118-148, 179-202; model.py:727-763): tile-list build, encoder forward, fused decoder + likelihood + backward,
encoder backward, AdamW; for the paired workloads BOTH directions (RNA->ATAC, ATAC->RNA) per step.

N > 1: one rank per GPU, cells sharded by rank, gradients all-reduced on RCCL, weak scaling (per-GPU batch fixed).  The
driver starts the ranks with torch.distributed.run; run by hand without it, `--gpus N` starts them itself (the parent
never touches the GPU; on a box with fewer than N GPUs it falls back to the gloo rehearsal with ranks sharing GPUs).

Rank 0 prints ONE JSON line with `roofline` (the dominant kernel of the workload, HIP events around each of its launches
in untimed single-stream steps after the timed region), `cpu_baseline` (the CPU oracle timed on the host cores on a
bounded sample of the same workload; N = 1 only) and, for the paired workloads at N = 1, `validation` and
`mapping_stage` cells/s (reported separately).
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

PRESETS = {
    # kind, cells (whole job as BASELINE quotes it), genes, peaks, batch per GPU, zinb, GPUs the config is quoted on
    "C2": dict(kind="paired", cells=100_000, genes=20_000, peaks=100_000, batch=1024, zinb=False, quoted_gpus=1),
    "C3": dict(kind="paired", cells=100_000, genes=20_000, peaks=100_000, batch=1024, zinb=True, quoted_gpus=1),
    "C4": dict(kind="vae", cells=1_000_000, genes=20_000, peaks=0, batch=1024, zinb=False, quoted_gpus=8),
    "C5": dict(kind="paired", cells=1_000_000, genes=20_000, peaks=250_000, batch=4096, zinb=False, quoted_gpus=8),
}
XGMI_PEAK_GBS = 7 * 153.0          # MI355X: 7 xGMI links x ~153 GB/s per GPU (MI355X_MICROARCH.md)


VALU_LANE_RATE = 256 * 4 * 16          # fp32 lanes per clock, whole chip: 256 CUs x 4 SIMDs x 16 lanes (a wave64 instruction = 4 cycles)
CLOCK_GHZ = 2.4                        # MI355X peak engine clock (MI355X_MICROARCH.md); the sustained clock under load is lower


def decoder_roofline(a, dec_ev, n_roof):
    """VALU-issue bound of the fused decoders (K5-K8: no HBM or MFMA roofline applies -- the logits never leave the CU and
    the K <= 64 products keep the matrix pipes under 10 % busy).  Bound = vector instructions the kernels retire
    (SQ_INSTS_VALU of a separate rocprofv3 --pmc pass over this command, stored in profiles/) x 4 cycles per wave64
    instruction, + 12 more cycles for every quarter-rate transcendental, over 1024 SIMDs at the peak clock.  `achieved`
    = elements per second of the decoder entry points (their passes + finishing launches, HIP events)."""
    if not dec_ev:
        return None
    stored = None
    try:
        with open(os.path.join(ROOT, "profiles", f"r03_{a.workload.lower()}_decoder_valu.json")) as fh:
            stored = json.load(fh)
        if (stored.get("batch"), stored.get("genes"), stored.get("peaks")) != (a.batch, a.genes, a.peaks) or a.precision != "bf16":
            stored = None
    except (OSError, ValueError):
        stored = None
    out = []
    for kind, passes, what in (("ber", 1, "Bernoulli decoder (scp_ber_dec_bce: dec_kernel<BER_TRAIN> + finishing launch)"),
                               ("nb", 3, "NB decoder (scp_nb_dec_ll: dec_kernel<NB_STATS / NB_LOSS / NB_GRAD> + 3 finishing launches)")):
        ev = [e for e in dec_ev if e[2] == kind and e[6]]
        if not ev:
            continue
        ms = sum(e0.elapsed_time(e1) for e0, e1, *_ in ev) / n_roof
        elems = sum(b_ * f_ for _, _, _, b_, f_, _, _ in ev) / n_roof          # label elements per step (each visited `passes` times)
        ent = {"kernel": what, "bound": "valu", "elements_per_step": elems, "passes": passes, "avg_ms_per_step": ms,
               "achieved": elems / (ms * 1e-3) / 1e9 if ms > 0 else 0.0, "unit": "G elements/s", "peak": None, "frac": None}
        st = (stored or {}).get("kernels", {}).get(kind)
        if st and ms > 0:
            cyc = (st["insts_valu"] * 4.0 + st["transcendentals_per_element"] * (elems / 64.0) * 12.0) / (256 * 4)
            floor_ms = cyc / (CLOCK_GHZ * 1e9) * 1e3
            ent.update({"valu_instructions_per_element": st["insts_valu"] / (elems / 64.0),
                        "transcendentals_per_element": st["transcendentals_per_element"],
                        "issue_floor_ms": floor_ms, "peak": elems / (floor_ms * 1e-3) / 1e9, "frac": floor_ms / ms,
                        "lds_bank_conflict_share": st.get("lds_bank_conflict_share"),
                        "counters_source": f"stored: profiles/r03_{a.workload.lower()}_decoder_valu.json ({stored.get('command')})"})
        out.append(ent)
    return out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 300 for C2, fewer for the heavier workloads)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="C2", choices=sorted(PRESETS))
    ap.add_argument("--cells", type=int, default=None)
    ap.add_argument("--genes", type=int, default=None)
    ap.add_argument("--peaks", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--hidden", type=int, nargs=2, default=[900, 40])
    ap.add_argument("--precision", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--p-atac", type=float, default=0.03)
    ap.add_argument("--rna-density", type=float, default=0.10)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--cov", type=int, default=0, help="levels of a one-hot covariate given to encoders and decoders (reference default cov=['labels'])")
    ap.add_argument("--batchnorm", action="store_true", help="BatchNorm1d after every FCNN activation (tutorial configuration)")
    ap.add_argument("--cpu-steps", type=int, default=3, help="steps of the CPU baseline (0 = skip)")
    ap.add_argument("--streams", type=int, default=None, choices=[1, 2],
                    help="paired workloads, 2: the two directions (independent trainings, model.py:608) run on two HIP streams "
                         "(the default for one process); N > 1 defaults to 1: that step is host-bound (one RCCL rank: 2.20 ms "
                         "against 2.41 with two streams) and both directions share one communicator")
    ap.add_argument("--fp32-exchange", action="store_true",
                    help="N > 1: all-reduce the wide first-layer gradients in fp32 even in bf16 mode (default: bf16)")
    ap.add_argument("--no-fuse-opt", action="store_true",
                    help="keep the AdamW step of the wide weights in the optimiser kernel (always so for N > 1)")
    ap.add_argument("--no-grad-sink", action="store_true",
                    help="N > 1, bf16 exchange: keep fp32 first-layer gradients + cast/copy instead of the bf16 exchange buffer")
    ap.add_argument("--zinb", action="store_true", help="same as --workload C3 on the C2 shapes")
    ap.add_argument("--shard-first-layer", action="store_true",
                    help="N > 1: feature-sharded first layers (every rank owns a column block of the wide weights and their "
                         "AdamW state; reduce-scatter of pre-activations + all-gather of dY instead of the dense gradient "
                         "all-reduce).  The default for N > 1 on the paired workloads without covariates / BatchNorm")
    ap.add_argument("--replicated", action="store_true",
                    help="N > 1: replicated first-layer weights and the dense gradient all-reduce (the round-2 scheme)")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="build each minibatch's tile lists at the start of its own step instead of during the previous one")
    ap.add_argument("--no-extras", action="store_true", help="skip the separately reported validation / mapping-stage rates")
    ap.add_argument("--roofline-steps", type=int, default=20,
                    help="untimed single-stream steps after the timed region whose per-launch HIP events give roofline.achieved")
    a = ap.parse_args()
    if a.zinb and a.workload == "C2":
        a.workload = "C3"
    pre = PRESETS[a.workload]
    a.kind = pre["kind"]
    a.zinb = bool(pre["zinb"])
    a.quoted_gpus = pre["quoted_gpus"]
    for k in ("cells", "genes", "peaks", "batch"):
        if getattr(a, k) is None:
            setattr(a, k, pre[k])
    if a.steps is None:
        a.steps = {"C2": 300, "C3": 100, "C4": 200, "C5": 60}[a.workload]
    return a


def spawn_ranks(a):
    """`--gpus N` run by hand (no WORLD_SIZE in the environment): start the N ranks as a torch.distributed.run child
    BEFORE this process makes any GPU call, wait, and exit with its status.  torch.cuda.device_count() does not
    initialise the GPU on this image."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    if torch.cuda.device_count() < a.gpus:
        env.setdefault("SCPAIR_BENCH_BACKEND", "gloo")       # rehearsal: ranks share the visible GPU(s)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _groups(S, enc, dec, lr=1e-3, wd=1e-9):
    de, ne, _, _ = S.add_weight_decay(enc, output_layer=["output"])
    dd, nd, _, _ = S.add_weight_decay(dec, output_layer=["output"])
    groups = [{"params": de, "weight_decay": wd, "lr": lr}, {"params": dd, "weight_decay": wd, "lr": lr},
              {"params": ne, "weight_decay": 0, "lr": lr}, {"params": nd, "weight_decay": 0, "lr": lr}]
    return [g for g in groups if g["params"]]


class Direction:
    """One encoder -> decoder predictor with its optimiser, built like scPair_object._build_predictor."""

    def __init__(self, S, a, in_dim, out_dim, out_dist, infer_lib, sample_factor, dev, seed=0):
        from torch import nn
        hidden = a.hidden
        S.set_seed(seed)
        self.enc = S.Input_Module(input_dim=in_dim, input_batch_num=a.cov, hidden_layer=hidden, activation=nn.LeakyReLU(),
                                  layernorm=True, batchnorm=a.batchnorm, dropout_rate=a.dropout, add_linear_layer=True,
                                  infer_library_size=infer_lib)
        S.set_seed(seed)
        if out_dist == "ber":
            self.dec = S.Output_Module_Ber(output_dim=out_dim, output_batch_num=a.cov, hidden_layer=hidden,
                                           infer_library_size=infer_lib, sample_factor=sample_factor, feature_factor=False)
        else:
            # zero_inflated=False for 'nb' (the headline config).  The reference also allocates an unused Linear(G,G)
            # `output_inverse_dispersion` (model.py:224-225, 1.6 GB at G=20k, never touched by forward/backward); it is
            # kept so the state_dict stays interchangeable.
            self.dec = S.Output_Module_Raw(output_dim=out_dim, output_batch_num=a.cov, hidden_layer=hidden,
                                           infer_library_size=not sample_factor, sample_factor=sample_factor,
                                           feature_factor=False, zero_inflated=(out_dist == "zinb"), dispersion="feature-cell")
        self.enc.apply(S.init_weights).to(dev)
        self.dec.apply(S.init_weights).to(dev)
        self.opt = S.FusedAdamW(_groups(S, self.enc, self.dec))
        self.first = [self.enc.hidden_rep[0].weight] + ([self.enc.hidden_lib[0][0].weight] if infer_lib else [])
        self.wide = []                      # wide plain layers whose AdamW step runs in their weight-gradient GEMM
        self.dist = out_dist
        self.fused = False
        self.sink = None
        self.shard = None
        self.use_lib = sample_factor
        self.enc.train(); self.dec.train()

    def apply_shard(self, S, shard, fuse):
        """Feature-sharded data parallelism: the first-layer weights become this rank's column block (the optimiser is
        rebuilt around the new parameters); their AdamW step runs in the weight-gradient kernel when `fuse`."""
        self.first = shard.shard_module(self.enc)
        self.opt = S.FusedAdamW(_groups(S, self.enc, self.dec))
        self.shard = shard
        if fuse:
            self.opt.fuse_first_layer(*self.first)
            self.fused = True

    def fuse(self):
        self.opt.fuse_first_layer(*self.first)
        self.fused = True
        w = getattr(self.dec, "output_dropout", None)          # C3: the G x G zero-inflation layer
        if isinstance(w, torch.nn.Linear) and self.dist == "zinb":
            self.opt.fuse_weight_update(w.weight)
            self.wide = [w.weight]

    def loss(self, S, tx, ty, cov):
        from scpair_amd.utils import _predictor_loss
        lib = ty.lib.reshape(-1, 1) if self.use_lib else None
        return _predictor_loss(tx, ty, lib, self.enc, self.dec, self.dist, cov is not None, cov, cov)

    def modules(self):
        return self.enc, self.dec

    def n_trained_params(self):
        """Parameters that received an update in the last step: those stepped by the multi-tensor kernel plus those
        updated inside their weight-gradient kernels (dead layers get no gradient and are not counted)."""
        extra = sum(p.numel() for p in self.first + self.wide) if self.fused else 0
        return sum(self.opt.last_step_numel) + extra

    def adamw_kernel_bytes(self):
        """Algorithmic bytes of this predictor's multi-tensor AdamW launch: 28 B/param, 26 where the
        gradient is read as bf16 from the exchange buffer; parameters updated inside their dW kernel are not in it."""
        n32, n16 = self.opt.last_step_numel
        return 28.0 * n32 + 26.0 * n16


class VaeModel:
    """VAE_ENCODER -> VAE_DECODER (pairedVAE.py:24-187) with the optimiser scVAE_object builds (C4)."""

    def __init__(self, S, a, in_dim, dev, seed=0):
        from torch import nn
        kw = dict(input_dim=in_dim, input_batch_num=a.cov, hidden_dim=[a.hidden[0]], latent_dim=a.hidden[1], layernorm=True,
                  activation=nn.LeakyReLU(), batchnorm=a.batchnorm, dropout_rate=a.dropout, infer_library_size=False)
        S.set_seed(seed)
        self.enc = S.VAE_ENCODER(**kw)
        S.set_seed(seed)
        self.dec = S.VAE_DECODER(**kw, sample_factor=True, feature_factor=False, distribution="nb", dispersion="feature-cell")
        self.enc.apply(S.init_weights).to(dev)
        self.dec.apply(S.init_weights).to(dev)
        self.opt = S.FusedAdamW(_groups(S, self.enc, self.dec))
        self.first = [self.enc.enc_hidden[0].weight]
        self.wide = []
        self.dist = "nb"
        self.fused = False
        self.sink = None
        self.enc.train(); self.dec.train()

    def fuse(self):
        # the decoder's 900 -> G heads keep the two-kernel update (fused into their GEMMs the step measured slower:
        # DESIGN.md section 4), so only the encoder's 20 k -> 900 layer is updated in its weight-gradient kernel
        self.opt.fuse_first_layer(*self.first)
        self.fused = True

    def loss(self, S, tx, ty, cov):
        from scpair_amd.utils import _vae_losses
        return _vae_losses(tx, tx.lib.reshape(-1, 1), self.enc, self.dec, 1.0, "nb", cov is not None, cov)[0]

    modules = Direction.modules
    n_trained_params = Direction.n_trained_params
    adamw_kernel_bytes = Direction.adamw_kernel_bytes


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)
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
    paired = a.kind == "paired"

    # ---- synthetic inputs, generated straight into HBM (per-rank shard of cells) ---------------------------
    # weak scaling: a rank holds the shard it would hold on the number of GPUs the config is quoted on
    cells_here = max(a.batch * 4, a.cells // max(world, a.quoted_gpus))
    rna = synth.make_rna_device(cells_here, a.genes, dev, seed=100 + rank, target_density=a.rna_density)
    atac = synth.make_atac_device(cells_here, a.peaks, dev, seed=200 + rank, p=a.p_atac) if paired else None
    nnz_r = rna.nnz / cells_here
    nnz_a = atac.nnz / cells_here if paired else 0.0
    cov_all = None
    if a.cov:
        g_ = torch.Generator(device=dev).manual_seed(300 + rank)
        cov_all = torch.nn.functional.one_hot(torch.randint(0, a.cov, (cells_here,), device=dev, generator=g_), a.cov).float()

    # ---- the models (reference defaults: model.py:365-372) ---------------------------------------------------
    if paired:
        A = Direction(S, a, a.genes, a.peaks, "ber", infer_lib=True, sample_factor=False, dev=dev)
        Bd = Direction(S, a, a.peaks, a.genes, "zinb" if a.zinb else "nb", infer_lib=False, sample_factor=True, dev=dev)
        models = [A, Bd]
    else:
        models = [VaeModel(S, a, a.genes, dev)]
    # feature-sharded first layers: the scheme for N > 1 wherever it applies (paired workload, fused encoder shape, a global
    # batch the weight-gradient kernel takes in one call); --replicated keeps the dense gradient all-reduce.  With one rank
    # it only runs when asked for (SCPAIR_BENCH_FORCE_DIST=1 --shard-first-layer: the RCCL call sequence on a one-GPU box)
    eligible = paired and not a.cov and not a.batchnorm and world * a.batch <= 32768 and a.precision == "bf16"
    sharded = bool(use_dist and eligible and not a.replicated and (world > 1 or a.shard_first_layer))
    if a.shard_first_layer and not sharded and use_dist:
        raise SystemExit("--shard-first-layer: paired bf16 workloads without covariates / BatchNorm, at most 32768 cells per "
                         "global batch")
    sync = None
    if use_dist:
        from scpair_amd.distributed import GradSync
        sync = GradSync(compress=(a.precision == "bf16" and not a.fp32_exchange))
        for d_ in models:
            sync.broadcast([p for m in d_.modules() for p in m.parameters()])
            if sync.compress and not a.no_grad_sink and not a.cov and not sharded:
                d_.sink = sync.make_sink(d_.first)     # dW of the wide first layers goes straight to the bf16 exchange buffer

    fuse = (sync is None) and (not a.no_fuse_opt) and a.precision == "bf16"
    if fuse:
        for d_ in models:
            d_.fuse()
    rna_cols = atac_cols = None
    if sharded:
        from scpair_amd.distributed import FirstLayerShard
        for d_, nfeat in ((A, a.genes), (Bd, a.peaks)):
            d_.sink = None
            d_.apply_shard(S, FirstLayerShard(sync, nfeat), fuse=(not a.no_fuse_opt) and a.precision == "bf16")
        # every rank holds ITS columns of ALL ranks' cells (the same entries in total as all columns of its own cells)
        def cols_of(make, own, shard, seed0, **kw):
            parts = []
            for r in range(world):
                full = own if r == rank else make(cells_here, shard.F, dev, seed=seed0 + r, **kw)
                parts.append(full.col_block(shard.lo, shard.hi))
                del full
            return S.DeviceCSR.vstack(parts).col_block(0, shard.hi - shard.lo, pad_row=True)
        rna_cols = cols_of(synth.make_rna_device, rna, A.shard, 100, target_density=a.rna_density)
        atac_cols = cols_of(synth.make_atac_device, atac, Bd.shard, 200, p=a.p_atac)
        torch.cuda.empty_cache()

    rng = np.random.default_rng(1234 + rank)
    state = {"perm": rng.permutation(cells_here), "cur": 0}
    state["perm_dev"] = torch.from_numpy(state["perm"]).to(dev)
    # feature-sharded: a rank also needs the other ranks' row ids of every global batch; their generators are replayed here
    peers = [{"rng": np.random.default_rng(1234 + r), "perm": None, "cur": 0} for r in range(world)] if sharded else []
    for r_, pr in enumerate(peers):
        pr["perm"] = pr["rng"].permutation(cells_here)
        pr["perm_dev"] = torch.from_numpy(pr["perm"] + r_ * cells_here).to(dev)
    share_dev = torch.full((), 1.0 / world, dtype=torch.float32, device=dev) if sharded else None
    adam_ev, fused_ev, exch_ev = [], [], []

    host_t = {"tiles": 0.0, "fwd": 0.0, "bwd": 0.0, "opt": 0.0}      # host enqueue time by phase (diagnostic)
    pc = time.perf_counter

    def exchange_bytes(D, grads):
        n = 0
        for g in grads:
            n += g.numel() * (2 if (sync.compress and g.numel() >= sync.big) else 4)
        if D.sink is not None:
            n += D.sink[0].numel() * 2
        return n

    def one_direction(D, tx, ty, cov, time_it):
        t0_ = pc()
        loss = D.loss(S, tx, ty, cov)
        t1_ = pc()
        D.opt.zero_grad()
        if paired:
            S.ops.backward_root(loss)
        else:
            loss.backward()
        host_t["fwd"] += t1_ - t0_
        host_t["bwd"] += pc() - t1_
        if sync is not None:
            grads = [p.grad for g in D.opt.param_groups for p in g["params"] if p.grad is not None]
            if time_it:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
                D._exch = (e0, exchange_bytes(D, grads))
            sync.start_all_reduce(grads, sink=D.sink)
        return loss

    def finish_direction(D, time_it):
        t0_ = pc()
        if sync is not None:
            sync.finish()
            if time_it:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                exch_ev.append((D._exch[0], e1, D._exch[1]))
        if time_it:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            D.opt.step(grad_scale=1.0 / world, release_grads=True)
            e1.record()
            adam_ev.append((e0, e1, D.adamw_kernel_bytes()))
        else:
            D.opt.step(grad_scale=1.0 / world, release_grads=True)
        host_t["opt"] += pc() - t0_

    def next_rows():
        if state["cur"] + a.batch > cells_here:
            state["perm"], state["cur"] = rng.permutation(cells_here), 0
            state["perm_dev"] = torch.from_numpy(state["perm"]).to(dev)        # one upload per epoch
        rows = state["perm"][state["cur"]:state["cur"] + a.batch]
        rows_dev = state["perm_dev"][state["cur"]:state["cur"] + a.batch]
        state["cur"] += a.batch
        return rows, rows_dev

    def run_models(t_rna, t_atac, cov, time_it, side, prefetch=None):
        x_rna, x_atac = t_rna, t_atac             # encoder inputs = the label tile lists, unless feature-sharded
        if sharded:
            (t_rna, x_rna), (t_atac, x_atac) = t_rna, t_atac
        if not paired:
            D = models[0]
            out = one_direction(D, t_rna, t_rna, cov, time_it)
            finish_direction(D, time_it)
            if prefetch is not None:
                prefetch()
            return (out,)
        if side is None:
            la = one_direction(A, x_rna, t_atac, cov, time_it)
            finish_direction(A, time_it)
            lb = one_direction(Bd, x_atac, t_rna, cov, time_it)
            finish_direction(Bd, time_it)
            return la, lb
        # the two predictors share nothing but the minibatch: one stream each, both after the tile-list build
        cur = torch.cuda.current_stream()
        for t in ((t_rna, t_atac, x_rna, x_atac) if sharded else (t_rna, t_atac)):
            for x in (t.ptr, t.key, t.val, t.pk, t.lib, t.nnan, t.rows):
                if x is not None:
                    for st in side:
                        x.record_stream(st)
        if cov is not None:
            for st in side:
                cov.record_stream(st)
        out = []
        for st, (D, tx, ty) in zip(side, ((A, x_rna, t_atac), (Bd, x_atac, t_rna))):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                out.append(one_direction(D, tx, ty, cov, time_it))
                finish_direction(D, time_it)
        if prefetch is not None:
            prefetch()      # the next minibatch's tile lists are built on this (otherwise idle) stream meanwhile
        for st in side:
            cur.wait_stream(st)
        return out[0], out[1]

    def global_rows():
        out, out_dev = [], []
        for r, pr in enumerate(peers):
            if pr["cur"] + a.batch > cells_here:
                pr["perm"], pr["cur"] = pr["rng"].permutation(cells_here), 0
                pr["perm_dev"] = torch.from_numpy(pr["perm"] + r * cells_here).to(dev)        # one upload per epoch
            out.append(pr["perm"][pr["cur"]:pr["cur"] + a.batch] + r * cells_here)
            out_dev.append(pr["perm_dev"][pr["cur"]:pr["cur"] + a.batch])
            pr["cur"] += a.batch
        return np.concatenate(out), (torch.cat(out_dev) if len(out_dev) > 1 else out_dev[0].contiguous())

    def build_tiles():
        rows, rows_dev = next_rows()
        t0_ = pc()
        t_rna = S.BatchTiles(rna, rows_dev, want_lib=True, rows_host=rows)
        t_atac = S.BatchTiles(atac, rows_dev, want_lib=False, rows_host=rows) if paired else None
        cov = None if cov_all is None else cov_all[rows_dev]
        if sharded:
            # encoder inputs: the GLOBAL batch over this rank's columns; t_rna / t_atac stay the labels of its own rows
            rg, rg_dev = global_rows()
            x_rna, _ = A.shard.batch_tiles(rna_cols, rg, share_dev, rows_dev=rg_dev)
            x_atac, _ = Bd.shard.batch_tiles(atac_cols, rg, share_dev, rows_dev=rg_dev)
            host_t["tiles"] += pc() - t0_
            return (t_rna, x_rna), (t_atac, x_atac), cov
        host_t["tiles"] += pc() - t0_
        return t_rna, t_atac, cov

    def step(time_it=False, side_=None):
        # The tile lists of the NEXT minibatch (the data-loader stage: row ids -> tile lists, no weights involved) are
        # built while this step's kernels run, as a prefetching loader would.
        nxt = state.pop("next_tiles", None)
        t_rna, t_atac, cov = nxt if nxt is not None else build_tiles()
        if a.no_prefetch or (paired and side_ is None):
            return run_models(t_rna, t_atac, cov, time_it, side_)

        def prefetch():
            state["next_tiles"] = build_tiles()
        return run_models(t_rna, t_atac, cov, time_it, side_, prefetch)

    if a.streams is None:
        a.streams = 1 if use_dist else 2
    side = (torch.cuda.Stream(), torch.cuda.Stream()) if (a.streams == 2 and paired) else None
    under_profiler = any(k_ in os.environ for k_ in ("ROCPROFILER_LIBRARY_CTOR", "ROCPROF_OUTPUT_PATH"))

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step(side_=side)
    barrier()
    prof = None
    if os.environ.get("SCPAIR_BENCH_CPROFILE") and rank == 0:      # diagnostic: where the host enqueue time goes
        import cProfile
        # the Python backward functions run on autograd's worker thread, which cProfile does not see: run them here
        torch.autograd.set_multithreading_enabled(False)
        prof = cProfile.Profile()
        prof.enable()
    for k_ in host_t:
        host_t[k_] = 0.0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses_t = step(side_=side)
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
        if backend == "nccl":
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        else:
            h_ = tmax.cpu()
            dist.all_reduce(h_, op=dist.ReduceOp.MAX)
            tmax = h_
    dt = float(tmax.item())
    losses = tuple(float(l_.detach()) for l_ in losses_t)
    # host cost of issuing ONE step into an idle queue (the run-ahead figure above includes time blocked on a full queue)
    t_issue = 0.0
    for _ in range(10):
        torch.cuda.synchronize()
        t1_ = time.perf_counter()
        step(side_=side)
        t_issue += time.perf_counter() - t1_
    barrier()
    host_by_phase = {k_: round(1e3 * v_ / a.steps, 3) for k_, v_ in host_t.items()}

    # ---- roofline pass (untimed, every rank: the steps hold collectives): the same step issued eagerly on ONE stream,
    # HIP events around every launch of the candidate dominant kernels on that stream, so the durations hold no time
    # queued behind the other direction's kernels and agree with rocprofv3's per-kernel durations of this command ----
    state.pop("next_tiles", None)
    if fuse or (sharded and A.fused):
        S.ops.set_fused_timer(fused_ev)
    n_roof = max(a.roofline_steps, 1)
    dec_ev = []
    S.ops.set_decoder_timer(dec_ev)
    for _ in range(n_roof):
        step(time_it=True, side_=None)
    S.ops.set_fused_timer(None)
    S.ops.set_decoder_timer(None)
    barrier()

    # ---- candidates for the dominant kernel; `roofline` = the one with the most time per step ----------------------
    # adamw_kernel: 28 B/param, 26 with bf16 gradients from the exchange buffer.
    # Fused forms (N = 1): a weight-gradient kernel with the AdamW step in its epilogue moves 24 B/param of optimiser
    # state (p, m, v read and written; the gradient stays on chip) + 2 B/param of bf16 shadow + its operands.
    n_params = sum(d_.n_trained_params() for d_ in models)
    cands = []
    tot_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in adam_ev)
    tot_bytes = sum(nb for _, _, nb in adam_ev)
    if tot_ms > 0:
        gbs = tot_bytes / (tot_ms * 1e-3) / 1e9
        cands.append({"kernel": "adamw_kernel (fused multi-tensor AdamW, 28 B/param; 26 with bf16 gradients)", "bound": "hbm",
                      "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, "traffic": None,
                      "algorithmic_bytes_per_step": tot_bytes / n_roof, "avg_ms_per_step_in_kernel": tot_ms / n_roof,
                      "launches_per_step": len(adam_ev) / n_roof})
    names = {"spmm_dw_adamw": "spmm_dw128_bf16_kernel / spmm_dw_bf16_kernel<*, ADAM> (first-layer weight gradient with the AdamW "
                              "step in its epilogue; one launch per encoder; 26 B/param + tile-list entries + bf16 dY^T)",
             "wgemm_adamw": "wgemm_kernel<true, true, EPI_ADAMW> (weight gradient of the wide G x G layer with the AdamW step "
                            "in its epilogue; 26 B/param + bf16 operands)"}
    for kind in ("spmm_dw_adamw", "wgemm_adamw"):
        ev = [e for e in fused_ev if e[4] == kind]
        if not ev:
            continue
        f_ms = sum(e0.elapsed_time(e1) for e0, e1, _, _, _ in ev)
        f_bytes = sum(26.0 * n + ob for _, _, n, ob, _ in ev)
        gbs = f_bytes / (f_ms * 1e-3) / 1e9 if f_ms > 0 else 0.0
        cands.append({"kernel": names[kind], "bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s",
                      "frac": gbs / 8000.0, "traffic": None, "algorithmic_bytes_per_step": f_bytes / n_roof,
                      "avg_ms_per_step_in_kernel": f_ms / n_roof, "launches_per_step": len(ev) / n_roof, "_kind": kind})
    cands.sort(key=lambda c: -c["avg_ms_per_step_in_kernel"])
    roof = dict(cands[0]) if cands else {"kernel": "none timed", "bound": "hbm", "achieved": 0.0, "peak": 8000.0,
                                         "unit": "GB/s", "frac": 0.0, "traffic": None}
    roof["timing"] = f"HIP events around each launch, {n_roof} eager single-stream steps after the timed region"
    roof["other_kernels"] = [{k_: v_ for k_, v_ in c.items() if not k_.startswith("_")} for c in cands[1:]]
    # HBM bytes of the kernel from PMC counters (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate rocprofv3 passes over
    # this command; profiles/*.json).  Only quoted when it was collected for this exact workload and kernel.
    for name in (f"r03_{a.workload.lower()}_traffic.json", "r02_dw_adamw_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                tr = json.load(fh)
            ref_bytes = tr.get("algorithmic_bytes_per_step", tr.get("optimizer_state_bytes_per_step"))
            want = roof.get("algorithmic_bytes_per_step", 0.0)
            if (tr.get("kind", "spmm_dw_adamw") == roof.get("_kind") and abs(ref_bytes - want) < 0.02 * want
                    and a.precision == "bf16" and not a.cov and not a.batchnorm):
                roof["traffic"] = tr["hbm_bytes_per_step"]
                roof["traffic_source"] = f"stored: profiles/{name} (separate rocprofv3 --pmc passes over this command)"
                break
        except (OSError, KeyError, ValueError, TypeError):
            pass
    roof.pop("_kind", None)
    dec_roof = decoder_roofline(a, dec_ev, n_roof)
    if dec_roof:
        roof["decoders"] = dec_roof
    if exch_ev:
        x_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in exch_ev)
        x_bytes = sum(nb for _, _, nb in exch_ev)
        alg = x_bytes / (x_ms * 1e-3) / 1e9 if x_ms > 0 else 0.0
        bus = alg * 2.0 * (world - 1) / world
        roof["exchange"] = {"what": "gradient all-reduce (RCCL), both buckets of every model, start -> wait on the compute stream",
                            "backend": backend, "message_bytes_per_step": x_bytes / n_roof, "avg_ms_per_step": x_ms / n_roof,
                            "algbw_GBs": alg, "busbw_GBs": bus, "peak_GBs": XGMI_PEAK_GBS,
                            "frac_of_xgmi": bus / XGMI_PEAK_GBS,
                            "note": None if backend == "nccl" else "gloo rehearsal through host memory: not an xGMI figure"}

    if paired:
        wl = (f"{a.workload}: {a.cells} cells x {a.genes} genes ({'ZINB, G x G zero-inflation layer live' if a.zinb else 'NB'}) x "
              f"{a.peaks} peaks (Bernoulli), batch {a.batch}/GPU, hidden {a.hidden}, dropout {a.dropout}, both directions per step")
    else:
        wl = (f"{a.workload}: unimodal VAE pretraining, {a.cells} cells x {a.genes} genes (NB), hidden [{a.hidden[0]}], latent "
              f"{a.hidden[1]}, batch {a.batch}/GPU, dropout {a.dropout}, one VAE step per batch")
    if a.cov or a.batchnorm:
        wl += f", covariate levels {a.cov}, batchnorm {a.batchnorm}"
    any_sink = any(d_.sink is not None for d_ in models)
    fs_info = None
    if sharded:
        Bg = world * a.batch
        h_a, h_b = sum(p.shape[0] for p in A.first), sum(p.shape[0] for p in Bd.first)
        fs_info = {"what": "feature-sharded first layers: per step and rank, SUM reduce-scatter of the pre-activations [B_global x H] "
                           "fp32 forward and all-gather of dY^T [H x B_global] bf16 backward, per encoder",
                   "allreduce_bytes_per_step": 4 * Bg * (h_a + h_b), "allgather_bytes_per_step": 2 * Bg * (h_a + h_b),
                   "replaces_dense_gradient_allreduce_bytes": 2 * (a.genes * h_a + a.peaks * h_b),
                   "first_layer_params_per_rank": sum(p.numel() for p in A.first + Bd.first)}
    out = {
        "metric": "training cells/sec (whole node) paired RNA+ATAC" if paired else "training cells/sec (whole node) unimodal RNA VAE",
        "value": a.batch * world * a.steps / dt,
        "unit": "paired cells/s" if paired else "cells/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16 MFMA inputs, f32 accumulate/weights" if a.precision == "bf16" else "f32 (3x bf16 split MFMA)",
        "data": "synthetic",
        "config": {"workload": wl,
                   "cells_resident_per_gpu": cells_here, "nnz_per_cell_rna": round(nnz_r, 1),
                   "nnz_per_cell_atac": round(nnz_a, 1), "precision": a.precision, "parallelism": f"dp{world}",
                   "streams": a.streams if paired else 1,
                   "first_layer": ("feature-sharded" if sharded else "replicated"), "first_layer_exchange": fs_info,
                   "grad_exchange": ("none" if sync is None else (("bf16 wide (written by the dW kernel) + fp32 rest" if any_sink
                                                                    else "bf16 wide + fp32 rest") if sync.compress else "fp32")),
                   "backend": (backend if use_dist else None),
                   "optimizer": ("AdamW; wide weights (first layers" + (", G x G layer" if a.zinb else "") + ") updated in their "
                                 "weight-gradient kernels") if fuse else
                                ("AdamW; each rank's first-layer column block updated in its weight-gradient kernel, the rest in one "
                                 "multi-tensor kernel" if (sharded and A.fused) else "AdamW, one multi-tensor kernel"),
                   "host_enqueue_ms_per_step": round(1e3 * t_issue / 10, 3),
                   "host_loop_ms_per_step": round(1e3 * t_host / a.steps, 3),
                   "host_enqueue_ms_by_phase": host_by_phase,
                   "step_issue": "eager launches (profiler attached)" if under_profiler else "eager launches",
                   "trained_params": n_params, "final_losses": losses},
        "roofline": roof,
    }

    if paired and world == 1 and not a.no_extras:
        out["validation"], out["mapping_stage"] = extras(a, S, A, Bd, rna, atac, cov_all, dev)
    if rank == 0 and world == 1 and a.cpu_steps > 0:
        if paired:
            out["cpu_baseline"] = cpu_baseline_paired(a, S, A, Bd, rna, atac, cov_all, rng)
        else:
            out["cpu_baseline"] = cpu_baseline_vae(a, S, models[0], rna, rng)
    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


@torch.no_grad()
def _val_pass(S, D, xcsr, ycsr, cov_all, n, chunk):
    """Per-epoch validation of one predictor as scPair_object._eval_split runs it (model.py:769-781): eval-mode forward
    + loss over a split in chunks, reduced to the full-split mean."""
    from scpair_amd.utils import _predictor_loss
    D.enc.eval(); D.dec.eval()
    num = torch.zeros((), dtype=torch.float64, device=xcsr.device)
    for r0 in range(0, n, chunk):
        rows = np.arange(r0, min(n, r0 + chunk))
        tx = S.BatchTiles(xcsr, rows, want_lib=False)
        ty = S.BatchTiles(ycsr, rows, want_lib=True)
        c = None if cov_all is None else cov_all[r0:r0 + len(rows)]
        lib = ty.lib.reshape(-1, 1) if D.use_lib else None
        num += _predictor_loss(tx, ty, lib, D.enc, D.dec, D.dist, c is not None, c, c).double() * len(rows)
    D.enc.train(); D.dec.train()
    return num / n


def extras(a, S, A, Bd, rna, atac, cov_all, dev):
    """The two stages SURVEY.md section 8d reports separately from the training rate, at the same shapes:
    validation (model.py:769-781: both predictors over a validation split, chunks of 4096 cells) and the latent-mapping
    stage (model.py:527-542 + 813-924: embeddings of the cells by both trained encoders, then minibatch training of the
    two Module_Module 40 -> 40 mapping nets with scaled_loss)."""
    from torch import nn
    n_val = min(rna.n_rows, 16384)
    chunk = 4096
    for _ in range(2):                    # first pass: warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        va = float(_val_pass(S, A, rna, atac, cov_all, n_val, chunk))
        vb = float(_val_pass(S, Bd, atac, rna, cov_all, n_val, chunk))
        torch.cuda.synchronize()
        t_val = time.perf_counter() - t0
    validation = {"value": n_val / t_val, "unit": "paired cells/s", "cells": n_val, "chunk": chunk,
                  "what": "eval-mode forward + loss of both predictors over a validation split (model.py:769-781)",
                  "losses": [va, vb]}

    # mapping stage: (i) embeddings
    n_emb = min(rna.n_rows, 32768)

    @torch.no_grad()
    def embed(D, csr):
        D.enc.eval()
        outs = []
        for r0 in range(0, n_emb, chunk):
            rows = np.arange(r0, min(n_emb, r0 + chunk))
            c = None if cov_all is None else cov_all[r0:r0 + len(rows)]
            outs.append(D.enc(S.BatchTiles(csr, rows, want_lib=False), c)[0])
        D.enc.train()
        return torch.cat(outs, 0)
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        er, ea = embed(A, rna), embed(Bd, atac)
        torch.cuda.synchronize()
        t_emb = time.perf_counter() - t0
    # (ii) the two mapping nets, one minibatch step each per "step" (model.py:838-883)
    nets = []
    for _ in range(2):
        S.set_seed(0)
        net = S.Module_Module(origin_module_dim=a.hidden[1], target_module_dim=a.hidden[1], translational_hidden_nodes=None,
                              non_neg=False, activation=nn.LeakyReLU(), layernorm=False, batchnorm=False, dropout_rate=0)
        net.apply(S.init_weights).to(dev)
        dp, ndp, _, _ = S.add_weight_decay(net, output_layer="bias")
        opt = S.FusedAdamW([g for g in ({"params": dp, "weight_decay": 1e-9, "lr": 1e-3},
                                        {"params": ndp, "weight_decay": 0, "lr": 1e-3}) if g["params"]])
        nets.append((net, opt))
    g_ = torch.Generator(device=dev).manual_seed(7)
    n_steps = 60

    def map_step():
        idx = torch.randint(0, n_emb, (a.batch,), device=dev, generator=g_)
        for (net, opt), (x, y) in zip(nets, ((er, ea), (ea, er))):
            loss = S.scaled_loss(y[idx], net(x[idx]))
            opt.zero_grad()
            loss.backward()
            opt.step()
        return loss
    for _ in range(5):
        map_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        l_ = map_step()
    torch.cuda.synchronize()
    t_map = (time.perf_counter() - t0) / n_steps
    mapping = {"embedding_cells_per_s": n_emb / t_emb, "embedding_cells": n_emb,
               "train_value": a.batch / t_map, "unit": "paired cells/s", "ms_per_step": 1e3 * t_map, "batch": a.batch,
               "what": "reference_embeddings of both modalities in chunks of 4096 (model.py:527-542), then one AdamW step of each "
                       "of the two 40 -> 40 mapping nets per step (model.py:813-924)",
               "final_loss": float(l_.detach())}
    return validation, mapping


def _cpu_threads():
    n = int(os.environ.get("OMP_NUM_THREADS", 0)) or min(16, os.cpu_count() or 1)
    torch.set_num_threads(n)
    return n


def cpu_baseline_paired(a, S, A, Bd, rna, atac, cov_all, rng):
    """The oracle (oracle/ref_cpu.py: dense fp32 torch CPU ops, same architecture/losses/AdamW grouping) timed on the
    host cores for `cpu_steps` paired steps of the same shapes, starting from the same weights."""
    from oracle import ref_cpu as O
    ncores = _cpu_threads()
    cpu = lambda m: {k: v.detach().cpu() for k, v in m.state_dict().items()
                     if not k.startswith("output_inverse_dispersion")}     # dead G x G layer: no grads, skip the copy
    flags = dict(batchnorm=a.batchnorm)
    dA = O.Direction(cpu(A.enc), cpu(A.dec), "ber", dict(sample_factor=False, **flags))
    dB = O.Direction(cpu(Bd.enc), cpu(Bd.dec), Bd.dist, dict(sample_factor=True, **flags))
    bsz = min(a.batch, 1024)               # bounded sample: the CPU path needs seconds per 1024 cells at these widths
    batches = []
    for _ in range(a.cpu_steps + 1):
        rows = np.sort(rng.choice(rna.n_rows, size=bsz, replace=False))
        xr = S.BatchTiles(rna, rows).to_dense().cpu()
        xa = S.BatchTiles(atac, rows).to_dense().cpu()
        c = None if cov_all is None else cov_all[torch.as_tensor(rows, device=cov_all.device)].cpu()
        batches.append((xr, xa, c))
    xr, xa, c = batches[0]
    dA.train_step(xr, xa, None, c); dB.train_step(xa, xr, xr.sum(1, keepdim=True), c)          # warm-up
    t0 = time.perf_counter()
    for xr, xa, c in batches[1:]:
        dA.train_step(xr, xa, None, c)
        dB.train_step(xa, xr, xr.sum(1, keepdim=True), c)
    dt = time.perf_counter() - t0
    return {"value": bsz * a.cpu_steps / dt, "unit": "paired cells/s", "cores": ncores, "cpu_model": cpu_model(), "kind": "port",
            "sample": f"{a.cpu_steps} paired steps of batch {bsz} at the same shapes and likelihoods (ber + {Bd.dist}) after 1 "
                      f"warm-up step, dense fp32 torch CPU ops, dropout 0"}


def cpu_baseline_vae(a, S, M, rna, rng):
    """One VAE step (pairedVAE.py:24-187 + utils.py:179-202) on the oracle's functions: forward, KL + NB loss, autograd
    gradients, the oracle's AdamW; same shapes and initial weights."""
    from oracle import ref_cpu as O
    ncores = _cpu_threads()
    enc = {k: v.detach().cpu().clone() for k, v in M.enc.state_dict().items()}
    dec = {k: v.detach().cpu().clone() for k, v in M.dec.state_dict().items()}
    params = {}
    for tag, sd in (("enc", enc), ("dec", dec)):
        for k, v in sd.items():
            if v.dtype.is_floating_point and "running_" not in k:
                v.requires_grad_(True)
                params[f"{tag}:{k}"] = v
    decay = [k for k, v in params.items() if v.ndim > 1 and not k.split(":", 1)[1].split("_")[0] in ["output"]]
    opt = O.AdamW(params, decay, lr=1e-3, wd=1e-9)
    bsz = min(a.batch, 1024)

    def one(x):
        eps = torch.randn((x.shape[0], a.hidden[1]))
        mu, sigma, z, _ = O.vae_encoder(enc, x, eps)
        scale, theta, rate, pi = O.vae_decoder(dec, z, lib=x.sum(1, keepdim=True), distribution="nb")
        loss = O.kl_normal(mu, sigma) + O.raw_loss(x, rate, theta, pi, "nb")
        names = list(params)
        grads = torch.autograd.grad(loss, [params[k] for k in names], allow_unused=True)
        opt.step(dict(zip(names, grads)))
        return float(loss.detach())
    xs = []
    for _ in range(a.cpu_steps + 1):
        rows = np.sort(rng.choice(rna.n_rows, size=bsz, replace=False))
        xs.append(S.BatchTiles(rna, rows).to_dense().cpu())
    one(xs[0])
    t0 = time.perf_counter()
    for x in xs[1:]:
        one(x)
    dt = time.perf_counter() - t0
    return {"value": bsz * a.cpu_steps / dt, "unit": "cells/s", "cores": ncores, "cpu_model": cpu_model(), "kind": "port",
            "sample": f"{a.cpu_steps} VAE steps of batch {bsz} at the same shapes (NB) after 1 warm-up step, dense fp32 torch CPU "
                      f"ops, dropout 0"}


if __name__ == "__main__":
    main()
