"""Data-parallel trainer on the GPU, two ranks (gloo, both on cuda:0): `scPair_object(grad_sync=GradSync())` slices
every global batch by rank, sums the count-weighted gradients and must reproduce the reference's SINGLE-process
trajectory (tests/golden/traj_nb_ber.npz: per-step training loss, per-epoch validation loss) — the all-reduce
semantics of SURVEY.md section 8c item 5, pinned against the reference's own run.  Tolerance: the fp32 trajectory
tolerance (2e-4 relative) widened to 5e-4 for the different summation order of the split batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests import util as U

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)           # a stuck collective ends the rank with a traceback
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import scpair_amd as S
    from scpair_amd.distributed import GradSync
    from tests.test_trajectory_gpu import _dicts
    d = U.load("traj_nb_ber")
    c = U.cfg(d, "cfg")
    tr, va, te = _dicts(c, d)
    obj = S.scPair_object(None, cov=None, modalities=c["modalities"], SEED=0, hidden_layer=c["hidden"], dropout_rate=0.0,
                          batchnorm=False, layernorm=True, batch_size=c["batch"], max_epochs=3,
                          device=torch.device("cuda:0"), save_model=False, zero_inflated=c["zero_inflated"],
                          verbose=False, grad_sync=GradSync())
    obj.train_predicting_networks(tr, va, te)
    if rank == 0:
        out.put({k: (list(map(float, v["train_loss"])), list(map(float, v["val_loss"]))) for k, v in obj.history.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_trainer_reproduces_reference_trajectory():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    hist = out.get(timeout=110)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = U.load("traj_nb_ber")
    c = U.cfg(d, "cfg")
    names = list(c["modalities"].keys())
    steps_per_epoch = int(np.ceil(len(d["dir0/train_loss"]) / c["epochs"]))
    for di in range(2):
        tl, vl = hist[names[di] + " to " + names[1 - di]]
        ref_t = d[f"dir{di}/train_loss"][: 3 * steps_per_epoch]
        ref_v = d[f"dir{di}/val_loss"][:3]
        assert len(tl) == len(ref_t) and len(vl) == 3
        assert float(np.max(np.abs(np.array(tl) - ref_t) / np.abs(ref_t))) < 5e-4, di
        assert float(np.max(np.abs(np.array(vl) - ref_v) / np.abs(ref_v))) < 5e-4, di
