"""World-size-2 gloo tests of the data-parallel gradient exchange (scpair_amd/distributed.py) on the CPU.
The N-rank global-batch gradient must equal the single-process gradient of the concatenated batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out, compress=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scpair_amd.distributed import GradSync
    sync = GradSync(big=1000, compress=compress)
    torch.manual_seed(0)
    # a tiny model with one "big" (>= big elements) and several small tensors
    W1 = torch.nn.Parameter(torch.randn(40, 50)); b1 = torch.nn.Parameter(torch.randn(40)); W2 = torch.nn.Parameter(torch.randn(3, 40))
    params = [W1, b1, W2]
    if rank != 0:
        for p in params:
            p.data.zero_()
    sync.broadcast(params)                                    # everyone starts from rank 0's weights
    g = torch.Generator().manual_seed(123)
    X = torch.randn(10, 50, generator=g); Y = torch.randn(10, 3, generator=g)
    rows = sync.shard(np.arange(10) if world == 2 else np.arange(10))   # ragged split when world does not divide 10
    if world == 3:
        assert [len(np.array_split(np.arange(10), 3)[r]) for r in range(3)] == [4, 3, 3]
    x, y = X[rows], Y[rows]
    loss = ((torch.relu(x @ W1.t() + b1) @ W2.t() - y) ** 2).mean()
    loss.backward()
    n_valid = float(y.numel())
    sync.all_reduce_grads(params, n_valid)
    gl = sync.weighted_mean(loss.detach(), n_valid)
    num, den = sync.reduce_scalars(torch.tensor(float(rank + 1)), torch.tensor(2.0))
    if rank == 0:
        out.put(([p.grad.clone().numpy() for p in params], float(gl), float(num), float(den),
                 [p.detach().clone().numpy() for p in params]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_global_batch_gradient_equals_single_process(world):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    grads, gl, num, den, weights = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the full batch
    torch.manual_seed(0)
    W1 = torch.randn(40, 50, requires_grad=True); b1 = torch.randn(40, requires_grad=True); W2 = torch.randn(3, 40, requires_grad=True)
    for w, got in zip((W1, b1, W2), weights):
        np.testing.assert_array_equal(w.detach().numpy(), got)
    g = torch.Generator().manual_seed(123)
    X = torch.randn(10, 50, generator=g); Y = torch.randn(10, 3, generator=g)
    loss = ((torch.relu(X @ W1.t() + b1) @ W2.t() - Y) ** 2).mean()
    loss.backward()
    for ref, got in zip((W1, b1, W2), grads):
        np.testing.assert_allclose(got, ref.grad.numpy(), rtol=1e-4, atol=1e-5)
    assert abs(gl - float(loss)) < 1e-5 * abs(float(loss))
    assert num == sum(range(1, world + 1)) and den == 2.0 * world


def test_bf16_exchange_of_wide_gradients():
    """compress=True: tensors >= `big` elements are all-reduced as bf16 (the wide first-layer dW in bf16 mode), the
    rest in fp32: the wide gradient matches the single-process one to bf16 rounding, the small ones to fp32."""
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out, True)) for r in range(world)]
    for p in procs:
        p.start()
    grads, gl, num, den, weights = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    W1 = torch.randn(40, 50, requires_grad=True); b1 = torch.randn(40, requires_grad=True); W2 = torch.randn(3, 40, requires_grad=True)
    g = torch.Generator().manual_seed(123)
    X = torch.randn(10, 50, generator=g); Y = torch.randn(10, 3, generator=g)
    loss = ((torch.relu(X @ W1.t() + b1) @ W2.t() - Y) ** 2).mean()
    loss.backward()
    scale = float(W1.grad.abs().max())
    np.testing.assert_allclose(grads[0], W1.grad.numpy(), rtol=2e-2, atol=1e-2 * scale)        # bf16: 8 bits
    assert np.abs(grads[0] - W1.grad.numpy()).max() > 0                                          # really compressed
    for ref, got in zip((b1, W2), grads[1:]):
        np.testing.assert_allclose(got, ref.grad.numpy(), rtol=1e-4, atol=1e-5)


def _sink_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scpair_amd import ops
    from scpair_amd.distributed import GradSync
    sync = GradSync(big=1000, compress=True)
    Wa = torch.nn.Parameter(torch.zeros(6, 40)); Wb = torch.nn.Parameter(torch.zeros(3, 40)); small = torch.nn.Parameter(torch.zeros(5))
    flat, handles = sync.make_sink([Wa, Wb])
    assert flat.dtype == torch.bfloat16 and flat.numel() == Wa.numel() + Wb.numel()
    assert ops.grad_sink_of(Wa) is handles[0] and ops.grad_sink_of(Wb).view.shape == Wb.shape
    # what the weight-gradient kernels do on the GPU: write this rank's bf16 dW into the views and flag them
    handles[0].view.copy_(torch.full((6, 40), float(rank + 1)))
    handles[1].view.copy_(torch.arange(120.0).reshape(3, 40) * (rank + 1))
    handles[0].written = handles[1].written = True
    small.grad = torch.full((5,), 0.5 * (rank + 1))
    sync.start_all_reduce([small.grad], sink=(flat, handles))
    sync.finish()
    # a step in which no kernel wrote the buffer must not reduce it again
    before = flat.clone()
    handles[0].written = handles[1].written = False
    sync.start_all_reduce([small.grad], sink=(flat, handles))
    sync.finish()
    if rank == 0:
        out.put((handles[0].view.float().numpy(), handles[1].view.float().numpy(), small.grad.numpy(),
                 bool(torch.equal(flat, before))))
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_exchange_buffer_written_by_the_kernels():
    """GradSync.make_sink: the wide first-layer gradients live as bf16 in one flat exchange buffer that is all-reduced
    in place with one collective (the GPU kernels write into its views; here the test does)."""
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sink_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    a, b, s, untouched = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(a, np.full((6, 40), 3.0, dtype=np.float32))
    np.testing.assert_allclose(b, np.arange(120.0).reshape(3, 40) * 3, rtol=2.0 ** -7)           # summed in bf16
    np.testing.assert_allclose(s, np.full(5, 1.5 + 0.75 * 2))      # 0.5 + 1.0 summed once, then summed again (1.5 * 2)
    assert untouched


def _exchange_worker(rank, world, port, out, n_rows):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scpair_amd.distributed import GradSync
    sync = GradSync(big=1000)
    torch.manual_seed(0)
    W1 = torch.nn.Parameter(torch.randn(40, 50)); b1 = torch.nn.Parameter(torch.randn(40)); W2 = torch.nn.Parameter(torch.randn(3, 40))
    dead = torch.nn.Parameter(torch.randn(7))                # never receives a gradient (like the unused G x G layer)
    params = [W1, b1, dead, W2]
    g = torch.Generator().manual_seed(123)
    X = torch.randn(10, 50, generator=g); Y = torch.randn(10, 3, generator=g)
    Y[0, 1] = float("nan")                                     # a masked label: counts differ from rows x features
    glob = np.arange(n_rows)
    rows = sync.shard(glob)
    some_empty = sync.some_rank_empty(len(glob))
    loss, n_valid = None, 0.0
    if len(rows):
        x, y = X[rows], Y[rows]
        pred = torch.relu(x @ W1.t() + b1) @ W2.t()
        mask = ~torch.isnan(y)
        loss = ((pred[mask] - y[mask]) ** 2).mean()
        loss.backward()
        n_valid = float(mask.sum())
    gl = sync.exchange(params, n_valid, loss, some_rank_empty=some_empty)
    out.put((rank, [None if p.grad is None else p.grad.clone().numpy() for p in params], float(gl), len(rows)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_rows", [1, 5])
def test_exchange_with_ragged_and_empty_shards(n_rows):
    """GradSync.exchange: one collective for (count, loss), count-weighted gradients; a global batch smaller than the
    world leaves a rank without rows (n_train % batch_size < world, drop_last=False: model.py:713-717) -- it joins
    the same collectives with zero gradients instead of hanging the others."""
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, out, n_rows)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted([out.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    assert [r[3] for r in res] == ([1, 0] if n_rows == 1 else [3, 2])
    torch.manual_seed(0)
    W1 = torch.randn(40, 50, requires_grad=True); b1 = torch.randn(40, requires_grad=True); W2 = torch.randn(3, 40, requires_grad=True)
    _ = torch.randn(7)
    g = torch.Generator().manual_seed(123)
    X = torch.randn(10, 50, generator=g)[:n_rows]; Y = torch.randn(10, 3, generator=g)[:n_rows]
    Y[0, 1] = float("nan")
    pred = torch.relu(X @ W1.t() + b1) @ W2.t()
    mask = ~torch.isnan(Y)
    loss = ((pred[mask] - Y[mask]) ** 2).mean()
    loss.backward()
    for rank, grads, gl, _ in res:
        assert grads[2] is None                                  # the dead parameter stays without gradient everywhere
        for ref, got in zip((W1, b1, W2), (grads[0], grads[1], grads[3])):
            np.testing.assert_allclose(got, ref.grad.numpy(), rtol=1e-4, atol=1e-5)
        assert abs(gl - float(loss)) < 1e-5 * abs(float(loss))


def _bn_worker(rank, world, port, out):
    """The statistics half of synchronised BatchNorm on two gloo ranks (the kernels themselves need a GPU: the full
    forward / backward is covered by tests/test_dp_trainer_gpu.py): per-rank (mean, M2, n) gathered and merged must
    equal the statistics of the whole batch, and the backward's formula with all-reduced sums must equal autograd
    through torch's batch_norm on the whole batch."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scpair_amd.distributed import GradSync
    from scpair_amd.ops import combine_bn_stats
    sync = GradSync()
    g = torch.Generator().manual_seed(5)
    H, eps = 7, 1e-5
    X = torch.randn(11, H, generator=g) * 3 + 1
    dY = torch.randn(11, H, generator=g)
    gamma = torch.rand(H, generator=g) + 0.5
    rows = sync.shard(np.arange(11))                         # 6 + 5 rows
    x = X[rows]
    mean_r = x.mean(0)
    st = sync.gather_rows(torch.cat([mean_r, (x - mean_r).square().sum(0), x.new_full((1,), float(len(rows)))]))
    mean, var, N = combine_bn_stats(st, H)
    rstd = 1.0 / torch.sqrt(var + eps)
    # global loss = mean over ALL rows of sum(dY * y): rank r's own loss is the mean over its rows, its share w_r = n_r / N
    w = sync.share_of(float(len(rows)), torch.device("cpu"))
    dy = dY[rows] / len(rows)                                # gradient of this rank's own loss
    xhat = (x - mean) * rstd
    S = sync._sum(torch.cat([dy.sum(0), (dy * xhat).sum(0)]) * w)
    dx = gamma * rstd * dy - gamma * rstd / (N * w) * (S[:H] + xhat * S[H:])
    out.put((rank, rows, mean.numpy(), var.numpy(), (dx * w).numpy()))      # w_r * local gradient = global gradient
    dist.barrier()
    dist.destroy_process_group()


def test_bn_statistics_and_backward_over_two_ranks_equal_the_whole_batch():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bn_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    g = torch.Generator().manual_seed(5)
    H, eps = 7, 1e-5
    X = (torch.randn(11, H, generator=g) * 3 + 1).requires_grad_(True)
    dY = torch.randn(11, H, generator=g)
    gamma = torch.rand(H, generator=g) + 0.5
    y = torch.nn.functional.batch_norm(X, None, None, gamma, torch.zeros(H), True, 0.1, eps)
    ((dY * y).sum() / 11).backward()
    np.testing.assert_allclose(res[0][2], X.detach().mean(0).numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(res[0][3], X.detach().var(0, unbiased=False).numpy(), rtol=1e-5)
    np.testing.assert_allclose(res[0][2], res[1][2], rtol=0, atol=0)          # identical on both ranks
    dx = np.concatenate([res[0][4], res[1][4]])
    np.testing.assert_allclose(dx, X.grad.numpy(), rtol=2e-4, atol=1e-6)


def _fs_worker(rank, world, port, out):
    """Feature-sharded first layer on two gloo ranks with plain matmuls as the local products: rows of Y and the
    gradient of each rank's column block must equal autograd on the whole batch and the whole weight."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scpair_amd.distributed import GradSync, feature_sharded_first_layer, shard_bounds
    sync = GradSync()
    g = torch.Generator().manual_seed(11)
    B, F, H = 11, 30, 5                                      # 6 + 5 rows; columns 0:15 / 15:30
    X = torch.randn(B, F, generator=g); W = torch.randn(H, F, generator=g); T = torch.randn(B, H, generator=g)
    f0, f1 = shard_bounds(F, world)[rank]
    lo, hi = shard_bounds(B, world)[rank]
    Wb = torch.nn.Parameter(W[:, f0:f1].clone())
    share = sync.share_of(float(hi - lo), torch.device("cpu"))
    y = feature_sharded_first_layer(Wb, X[:, f0:f1], sync, lambda x, w: x @ w.t(), lambda x, dy, shp: dy.t() @ x, share)
    loss = ((torch.tanh(y) - T[lo:hi]) ** 2).mean()         # this rank's own loss: mean over its rows
    loss.backward()
    out.put((rank, y.detach().numpy(), Wb.grad.numpy(), (f0, f1), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def test_feature_sharded_first_layer_equals_the_whole_batch():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fs_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    g = torch.Generator().manual_seed(11)
    B, F, H = 11, 30, 5
    X = torch.randn(B, F, generator=g); W = torch.randn(H, F, generator=g).requires_grad_(True); T = torch.randn(B, H, generator=g)
    Y = X @ W.t()
    ((torch.tanh(Y) - T) ** 2).mean().backward()            # the global-batch loss: mean over ALL rows
    for rank, y, gw, (f0, f1), (lo, hi) in res:
        np.testing.assert_allclose(y, Y.detach().numpy()[lo:hi], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(gw, W.grad.numpy()[:, f0:f1], rtol=1e-5, atol=1e-7)
