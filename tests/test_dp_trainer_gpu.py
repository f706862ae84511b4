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


def _train(grad_sync, batch, epochs, with_mapping, precision="fp32", batchnorm=False, shard=False):
    import scpair_amd as S
    from tests.test_trajectory_gpu import _dicts
    d = U.load("traj_nb_ber")
    c = U.cfg(d, "cfg")
    tr, va, te = _dicts(c, d)
    obj = S.scPair_object(None, cov=None, modalities=c["modalities"], SEED=0, hidden_layer=c["hidden"], dropout_rate=0.0,
                          batchnorm=batchnorm, layernorm=True, batch_size=batch, max_epochs=epochs,
                          device=torch.device("cuda:0"), save_model=False, zero_inflated=c["zero_inflated"],
                          verbose=False, grad_sync=grad_sync, use_mixed_precision=(precision == "bf16"),
                          shard_first_layer=shard)
    enc, _ = obj.train_predicting_networks(tr, va, te)
    if with_mapping:
        obj.reference_embeddings()
        obj.train_bidirectional_mapping()
    hist = {k: (list(map(float, v["train_loss"])), list(map(float, v["val_loss"]))) for k, v in obj.history.items()}
    if shard:
        # the returned encoders hold the full first-layer weights again (all-gather of the ranks' column blocks)
        names = list(c["modalities"].keys())
        dims = {names[0]: tr[names[0] + "_train_mtx"].shape[1], names[1]: tr[names[1] + "_train_mtx"].shape[1]}
        for k, e in enc.items():
            w = e.hidden_rep[0].weight
            assert w.shape[1] == dims[k.split(" to ")[0]] and e.hidden_rep[0].in_features == w.shape[1], (k, tuple(w.shape))
        hist["_w_sum"] = ([float(enc[k].hidden_rep[0].weight.double().abs().sum()) for k in sorted(enc)], [])
    return hist


def _worker(rank, world, port, out, batch, epochs, with_mapping, precision="fp32", compress=False, batchnorm=False,
            shard=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)           # a stuck collective ends the rank with a traceback
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scpair_amd.distributed import GradSync
    hist = _train(GradSync(compress=compress), batch, epochs, with_mapping, precision, batchnorm, shard)
    faulthandler.cancel_dump_traceback_later()
    if rank == 0:
        out.put(hist)
    dist.barrier()
    dist.destroy_process_group()


def _run_two_ranks(batch, epochs, with_mapping=False, precision="fp32", compress=False, batchnorm=False, shard=False):
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out, batch, epochs, with_mapping, precision, compress, batchnorm,
                                                shard))
             for r in range(world)]
    for p in procs:
        p.start()
    try:
        hist = out.get(timeout=110)
        for p in procs:
            p.join(timeout=60)
    finally:
        for p in procs:                         # a failed rank must not leave its peer holding the GPU and the port
            if p.is_alive():
                p.terminate()
                p.join(timeout=10)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return hist


def test_two_rank_trainer_reproduces_reference_trajectory():
    hist = _run_two_ranks(U.cfg(U.load("traj_nb_ber"), "cfg")["batch"], 3)
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


def _compare(h2, h1, tol):
    assert set(h2) == set(h1)
    for k in h1:
        for a, b in zip(h2[k], h1[k]):
            assert len(a) == len(b), k
            assert float(np.max(np.abs(np.array(a) - np.array(b)) / np.abs(np.array(b)))) < tol, (k, a, b)


def test_two_rank_trainer_with_an_empty_shard_and_the_mapping_stage():
    """410 training cells in batches of 409 leave a last global batch of ONE cell: rank 1's shard of it is empty and
    joins the collectives with zero gradients.  Predictors and the bidirectional mapping stage (also sliced by rank)
    against the single-process run of the same configuration (itself pinned to the reference by the trajectory tests)."""
    h2 = _run_two_ranks(409, 2, with_mapping=True)
    h1 = _train(None, 409, 2, True)
    _compare(h2, h1, 5e-4)


def test_two_rank_trainer_bf16_exchange_buffer():
    """Precision 'bf16' with GradSync(compress=True): the first-layer weight-gradient kernels write bf16 dW straight
    into the exchange buffer (the trainer wires it up like bench.py does); losses within the bf16 tolerance of the
    single-process bf16 run."""
    h2 = _run_two_ranks(64, 2, precision="bf16", compress=True)
    h1 = _train(None, 64, 2, False, precision="bf16")
    _compare(h2, h1, 2e-2)


def test_two_rank_trainer_with_batchnorm_uses_global_batch_statistics():
    """batchnorm=True under data parallelism: the BatchNorm layers normalise with the statistics of the GLOBAL batch
    (ops.SyncBatchNormFn: per-rank mean / M2 all-gathered and merged, the backward's two per-feature sums all-reduced
    with the ranks' loss shares), so the two-rank run follows the single-process run of the same configuration (itself
    pinned to the reference by tests/golden/traj_nb_ber_bn.npz in test_trajectory_gpu.py).  409-row batches over 410
    cells: the one-cell last batch leaves rank 1 empty and is the step that falls back to per-rank statistics."""
    b = U.cfg(U.load("traj_nb_ber"), "cfg")["batch"]
    h2 = _run_two_ranks(b, 2, batchnorm=True)
    h1 = _train(None, b, 2, False, batchnorm=True)
    _compare(h2, h1, 5e-4)
    h2 = _run_two_ranks(409, 2, batchnorm=True)
    h1 = _train(None, 409, 2, False, batchnorm=True)
    _compare(h2, h1, 5e-4)


def _fs_gpu_worker(rank, world, port, out):
    """Feature-sharded first layer with the HIP tile-list kernels as the local products (fp32-equivalent mode)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)
    import scipy.sparse as sp
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import scpair_amd as S
    from scpair_amd import ops
    from scpair_amd.distributed import GradSync, feature_sharded_first_layer, shard_bounds
    dev = torch.device("cuda:0")
    sync = GradSync()
    rng = np.random.default_rng(3)
    B, F, H = 150, 700, 72
    X = sp.random(B, F, density=0.06, format="csr", random_state=rng, dtype=np.float32)
    X.data = np.ceil(X.data * 5).astype(np.float32)
    g = torch.Generator().manual_seed(3)
    W = torch.randn(H, F, generator=g) * 0.1
    T = torch.randn(B, H, generator=g)
    f0, f1 = shard_bounds(F, world)[rank]
    lo, hi = shard_bounds(B, world)[rank]
    tiles = S.BatchTiles(S.DeviceCSR.from_scipy(X[:, f0:f1].tocsr(), dev), np.arange(B), want_lib=False)
    Wb = torch.nn.Parameter(W[:, f0:f1].contiguous().to(dev))

    def local_fwd(t, w):
        return ops.fcnn_layer(t, None, w, None, None, None, 1.0)

    def local_dw(t, dy, shp):
        wt = Wb.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            o = ops.fcnn_layer(t, None, wt, None, None, None, 1.0)
        o.backward(dy)
        return wt.grad

    share = sync.share_of(float(hi - lo), dev)
    y = feature_sharded_first_layer(Wb, tiles, sync, local_fwd, local_dw, share)
    loss = ((torch.tanh(y) - T[lo:hi].to(dev)) ** 2).mean()
    loss.backward()
    faulthandler.cancel_dump_traceback_later()
    out.put((rank, y.detach().cpu().numpy(), Wb.grad.cpu().numpy(), (f0, f1), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def test_feature_sharded_first_layer_on_the_tile_list_kernels():
    """Prototype of the feature-sharded first layer (scpair_amd/distributed.py): two ranks, each owning half of the
    columns of W and a tile list of the global batch restricted to them; rows of Y and each block's gradient against a
    dense fp32 computation of the whole batch (the kernels' fp32-equivalent mode: 1e-4)."""
    import scipy.sparse as sp
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fs_gpu_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [out.get(timeout=110) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(timeout=10)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    rng = np.random.default_rng(3)
    B, F, H = 150, 700, 72
    X = sp.random(B, F, density=0.06, format="csr", random_state=rng, dtype=np.float32)
    X.data = np.ceil(X.data * 5).astype(np.float32)
    g = torch.Generator().manual_seed(3)
    W = (torch.randn(H, F, generator=g) * 0.1).requires_grad_(True)
    T = torch.randn(B, H, generator=g)
    Y = torch.from_numpy(X.toarray()) @ W.t()
    ((torch.tanh(Y) - T) ** 2).mean().backward()
    for rank, y, gw, (f0, f1), (lo, hi) in res:
        ref_y, ref_g = Y.detach().numpy()[lo:hi], W.grad.numpy()[:, f0:f1]
        assert float(np.abs(y - ref_y).max()) < 1e-4 * float(np.abs(ref_y).max()), rank
        assert float(np.abs(gw - ref_g).max()) < 1e-4 * float(np.abs(ref_g).max()), rank


def test_two_rank_trainer_with_feature_sharded_first_layer_reproduces_reference_trajectory():
    """scPair_object(grad_sync=..., shard_first_layer=True): every rank owns a column block of the first-layer weights and
    holds those columns of all cells; the all-reduce of the pre-activations and the all-gather of dY replace the all-reduce
    of the dense first-layer gradient.  The two-rank run must follow the reference's single-process trajectory like the
    replicated data-parallel run does (5e-4), ragged last batch (26 rows -> 13 + 13, padded to 64 per rank) included."""
    hist = _run_two_ranks(U.cfg(U.load("traj_nb_ber"), "cfg")["batch"], 3, shard=True)
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


def test_two_rank_feature_sharded_bf16_fused_step_matches_single_process():
    """Precision 'bf16': each rank's block takes its AdamW step inside its weight-gradient kernel over the GLOBAL batch
    (K2' per rank, no gradient exchange for the wide weights).  Losses and the magnitude of the reassembled first-layer
    weights within the bf16 tolerance of the single-process bf16 run."""
    h2 = _run_two_ranks(64, 2, precision="bf16", shard=True)
    h1 = _train(None, 64, 2, False, precision="bf16")
    w2 = h2.pop("_w_sum")[0]
    _compare(h2, h1, 2e-2)
    import scpair_amd as S
    # single-process weights for the same two epochs
    from tests.test_trajectory_gpu import _dicts
    d = U.load("traj_nb_ber"); c = U.cfg(d, "cfg"); tr, va, te = _dicts(c, d)
    obj = S.scPair_object(None, cov=None, modalities=c["modalities"], SEED=0, hidden_layer=c["hidden"], dropout_rate=0.0,
                          batchnorm=False, layernorm=True, batch_size=64, max_epochs=2, device=torch.device("cuda:0"),
                          save_model=False, zero_inflated=c["zero_inflated"], verbose=False, use_mixed_precision=True)
    enc, _ = obj.train_predicting_networks(tr, va, te)
    w1 = [float(enc[k].hidden_rep[0].weight.double().abs().sum()) for k in sorted(enc)]
    for a, b in zip(w2, w1):
        assert abs(a - b) < 2e-3 * abs(b), (w2, w1)
