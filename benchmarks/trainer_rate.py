"""Rate of the product trainer itself -- `scPair_object.train_predicting_networks` (model.py:603-812 of the reference) at the C2
shapes -- as opposed to `bench.py`'s own step loop: the same kernels, but driven by the trainer's epoch loop (DataLoader over
row ids, tile lists built in line, validation + checkpoint copy per epoch; the two directions side by side on two streams,
or one after the other with --sequential).

Two runs with different epoch counts; the difference gives the cost of an epoch without the per-direction set-up (weight
initialisation, homing of the first-layer weights, optimiser construction).  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import scpair_amd as S
from scpair_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=20480)
    ap.add_argument("--val", type=int, default=2048)
    ap.add_argument("--genes", type=int, default=20000)
    ap.add_argument("--peaks", type=int, default=100000)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--epochs", type=int, nargs=2, default=[2, 8])
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--vae", action="store_true", help="the unimodal VAE trainer (scVAE_object.train) at the C4 shapes instead")
    ap.add_argument("--dp", choices=["none", "replicated", "sharded"], default="none",
                    help="run the data-parallel trainer with ONE rank over RCCL (the call sequence and host cost of the N > 1 "
                         "trainer on a one-GPU box): replicated weights, or feature-sharded first layers")
    ap.add_argument("--explicit-peak-values", action="store_true",
                    help="hold the binary peak matrices WITH a value array of ones (what the trainer's data seam did with a "
                         "binarised scipy matrix before it detected all-ones data): the value-carrying kernel paths")
    ap.add_argument("--zinb", action="store_true", help="ZINB likelihood for the RNA output (the live G x G zero-inflation layer: C3)")
    ap.add_argument("--cov", type=int, default=0, help="one-hot covariate block of this many levels (cov=[...] of the constructor)")
    ap.add_argument("--batchnorm", action="store_true")
    ap.add_argument("--hidden", type=int, nargs=2, default=[900, 40])
    ap.add_argument("--sequential", action="store_true",
                    help="train the predictors one after the other (the reference's order) instead of side by side")
    ap.add_argument("--profile", default=None, help="write a cProfile listing (by cumulative time) of the longer run here")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    sync = None
    if a.dp != "none":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29571")
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        from scpair_amd.distributed import GradSync
        S.set_precision(a.precision)
        sync = GradSync(compress=(a.precision == "bf16"))
    mk = lambda n, s: (synth.make_rna_device(n, a.genes, dev, seed=100 + s, target_density=0.10),
                       synth.make_atac_device(n, a.peaks, dev, seed=200 + s, p=0.03))
    splits = {"train": mk(a.cells, 0), "val": mk(a.val, 1), "test": mk(64, 2)}
    if a.explicit_peak_values:
        splits = {k: (r, p.with_explicit_values()) for k, (r, p) in splits.items()}
    dicts = []
    for split in ("train", "val", "test"):
        r, p = splits[split]
        cov = None
        if a.cov:
            g_ = torch.Generator().manual_seed(7 + len(dicts))
            cov = torch.nn.functional.one_hot(torch.randint(0, a.cov, (r.n_rows,), generator=g_), a.cov).float()
        dicts.append({f"Gene Expression_{split}_mtx": r, f"Peaks_{split}_mtx": p,
                      f"Gene Expression_{split}_cov": cov, f"Peaks_{split}_cov": cov})

    def run_vae(epochs):
        obj = S.scVAE_object(distribution="nb", SEED=0, hidden_dim=[900], latent_dim=40, dropout_rate=0.1, batch_size=a.batch,
                             max_epochs=epochs, early_stopping_patience=1000, device=dev, save_model=False,
                             use_mixed_precision=(a.precision == "bf16"), verbose=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        obj.train(splits["train"][0], splits["val"][0])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return dt, len(obj.history["train_loss"]), {"vae": (obj.history["train_loss"][-1], obj.history["val_loss"][-1])}, 0.0

    def run(epochs):
        if a.vae:
            return run_vae(epochs)
        obj = S.scPair_object(None, cov=(["labels"] if a.cov else None),
                              modalities={"Gene Expression": "zinb" if a.zinb else "nb", "Peaks": "ber"},
                              SEED=0, hidden_layer=list(a.hidden), batchnorm=a.batchnorm, dropout_rate=0.1, batch_size=a.batch, max_epochs=epochs,
                              early_stopping_patience=1000, device=dev, save_model=False, zero_inflated=bool(a.zinb),
                              use_mixed_precision=(a.precision == "bf16"), verbose=False, grad_sync=sync,
                              shard_first_layer=(a.dp == "sharded"), concurrent_directions=not a.sequential)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        obj.train_predicting_networks(*dicts)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steps = sum(len(h["train_loss"]) for h in obj.history.values())
        last = {k: (h["train_loss"][-1], h["val_loss"][-1]) for k, h in obj.history.items()}
        obj.reference_embeddings()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        obj.train_bidirectional_mapping()          # model.py:813-924: the two 40 -> 40 mapping nets, same epoch control
        torch.cuda.synchronize()
        return dt, steps, last, time.perf_counter() - t1

    run(1)                                   # warm-up: library load, allocator pools
    t_lo, s_lo, _, m_lo = run(a.epochs[0])
    if a.profile:
        import cProfile
        import pstats
        prof = cProfile.Profile()
        prof.enable()
    t_hi, s_hi, last, m_hi = run(a.epochs[1])
    if a.profile:
        prof.disable()
        with open(a.profile, "w") as fh:
            pstats.Stats(prof, stream=fh).sort_stats("cumulative").print_stats(60)
    d_ep = a.epochs[1] - a.epochs[0]
    per_epoch = (t_hi - t_lo) / d_ep         # both directions: one epoch of each predictor, validation and checkpoint copy included
    steps_per_epoch = (s_hi - s_lo) / d_ep   # direction steps per epoch pair
    print(json.dumps({
        "what": ("scVAE_object.train, C4 shapes (unimodal NB VAE)" if a.vae else
                 "scPair_object.train_predicting_networks, C2 shapes, both predictors " +
                 ("one after the other" if (a.sequential or a.dp != "none") else "side by side on two streams")) +
                "; marginal cost of an epoch (training steps + validation over the val split + deepcopy of the best modules) "
                "from two runs",
        "zinb": bool(a.zinb), "cov": a.cov, "batchnorm": bool(a.batchnorm), "hidden": list(a.hidden), "data_parallel": a.dp, "sequential": bool(a.sequential), "explicit_peak_values": bool(a.explicit_peak_values), "precision": a.precision, "cells": a.cells, "val_cells": a.val, "batch": a.batch, "epochs": a.epochs,
        "seconds": [round(t_lo, 4), round(t_hi, 4)],
        "ms_per_epoch_pair": round(1e3 * per_epoch, 3),
        "ms_per_direction_step_incl_validation": round(1e3 * per_epoch / steps_per_epoch, 4),
        "cells_per_s": round(a.cells / per_epoch, 1),
        "setup_s_per_run": round(t_lo - a.epochs[0] * per_epoch, 3),
        "mapping_ms_per_epoch_pair": None if a.vae else round(1e3 * (m_hi - m_lo) / d_ep, 3),
        "mapping_paired_cells_per_s": None if a.vae else round(a.cells * d_ep / (m_hi - m_lo), 1),
        "last_losses": {k: [float(v[0]), float(v[1])] for k, v in last.items()},
    }))


if __name__ == "__main__":
    main()
