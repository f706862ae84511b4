"""scp_colsum_noalloc (the one entry point that allocates nothing: its [32][cols] partial sums live in a device array
inside the library) against scp_colsum with a caller workspace, through the C ABI: bit-exact, on two calls in a row, the
second with a smaller input (stale partials of the first call must not leak), plus the size limit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(L, _lib, X, rows, cols, ld):
    out_ref = torch.full((cols,), float("nan"), device=X.device)
    ws = torch.empty(32 * cols, device=X.device)
    _lib.check(L.scp_colsum(X.data_ptr(), ld, rows, cols, out_ref.data_ptr(), ws.data_ptr(), _lib.stream()), "scp_colsum")
    out = torch.full((cols,), float("nan"), device=X.device)
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    _lib.check(L.scp_colsum_noalloc(X.data_ptr(), ld, rows, cols, out.data_ptr(), _lib.stream()), "scp_colsum_noalloc")
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    return out, out_ref


def test_colsum_noalloc_matches_colsum():
    from scpair_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    X1 = torch.randn(1024, 900, generator=g).to(dev)
    out, ref = _both(L, _lib, X1, 1024, 900, 900)
    assert torch.equal(out, ref)
    assert float((out.double().cpu() - X1.double().cpu().sum(0)).abs().max()) < 1e-3
    # second call, smaller input (fewer columns, fewer than 32 rows -> one chunk only, padded leading dimension)
    X2 = torch.randn(7, 40, generator=g).to(dev)
    out, ref = _both(L, _lib, X2, 7, 33, 40)
    assert torch.equal(out, ref)
    assert float((out.double().cpu() - X2[:, :33].double().cpu().sum(0)).abs().max()) < 1e-5
    # and a larger one again
    X3 = torch.randn(130, 16384, generator=g).to(dev)
    out, ref = _both(L, _lib, X3, 130, 16384, 16384)
    assert torch.equal(out, ref)
    # beyond the built-in workspace: clear error, nothing launched
    assert L.scp_colsum_noalloc_max_cols() == 16384
    X4 = torch.zeros(2, 16385, device=dev)
    o4 = torch.zeros(16385, device=dev)
    rc = L.scp_colsum_noalloc(X4.data_ptr(), 16385, 2, 16385, o4.data_ptr(), _lib.stream())
    assert rc == -2 and b"exceeds the built-in workspace" in L.scp_last_error()
