"""Device-resident CSR count matrices and per-minibatch tile lists.

Replaces the reference's dense CPU tensors + LazyGPUDataset/DataLoader batch gather
(model.py:26-38, 492-510, 704-717): a split (train/val/test) of one modality stays in HBM as CSR for
the whole run, and a minibatch is just a vector of its row ids.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


def rows_to_device(rows_host: np.ndarray, device):
    """Row ids of a minibatch -> int64 device tensor without stalling the host: a copy from pageable memory waits for the
    stream to drain (every kernel of the previous step), which leaves the GPU idle while the next step is enqueued; staged
    through pinned memory (torch's caching host allocator keeps the block until the copy has run) it is stream-ordered
    only."""
    t = torch.from_numpy(np.ascontiguousarray(rows_host, dtype=np.int64))
    device = torch.device(device)
    if device.type == "cuda":
        return t.pin_memory().to(device, non_blocking=True)
    return t.to(device)


def scalar64(v, device):
    """A float64 device scalar from a Python number (a fill kernel: no host-to-device copy, which from pageable memory would
    wait for the stream to drain) or from a tensor."""
    if torch.is_tensor(v):
        return v.detach().to(device=device, dtype=torch.float64).reshape(())
    return torch.full((), float(v), dtype=torch.float64, device=device)


class DeviceCSR:
    """CSR cell x feature matrix in GPU memory (int64 indptr, int32 indices, fp32 values or implicit ones)."""

    def __init__(self, indptr, indices, values, shape, device):
        self.shape = (int(shape[0]), int(shape[1]))
        ip = np.ascontiguousarray(np.asarray(indptr, dtype=np.int64))
        self.indptr_host = ip
        self.device = torch.device(device)
        self.indptr = torch.from_numpy(ip).to(self.device)
        idx = np.ascontiguousarray(np.asarray(indices, dtype=np.int32))
        self.nnz = int(idx.shape[0])
        if self.nnz == 0:                     # keep device pointers non-null for an all-zero matrix
            idx = np.zeros(1, np.int32)
            values = None if values is None else np.zeros(1, np.float32)
        self.indices = torch.as_tensor(idx).to(self.device)
        self.values = None if values is None else torch.as_tensor(
            np.ascontiguousarray(np.asarray(values, dtype=np.float32))).to(self.device)
        if ip.shape[0] != self.shape[0] + 1:
            raise ValueError("indptr length does not match the row count")
        if self.shape[1] >= (1 << 25):
            raise ValueError("more than 2^25 features are not supported by the tile-list key format")

    @classmethod
    def from_device(cls, indptr, indices, values, shape):
        """Wrap CSR arrays that already live in GPU memory (int64 indptr, int32 indices, fp32 values or None)."""
        self = cls.__new__(cls)
        self.shape = (int(shape[0]), int(shape[1]))
        self.device = indptr.device
        self.indptr = indptr.to(torch.int64).contiguous()
        self.indptr_host = self.indptr.cpu().numpy()
        self.indices = indices.to(torch.int32).contiguous()
        self.nnz = int(self.indices.shape[0])
        self.values = None if values is None else values.to(torch.float32).contiguous()
        if self.nnz == 0:
            self.indices = torch.zeros(1, dtype=torch.int32, device=self.device)
            self.values = None if values is None else torch.zeros(1, dtype=torch.float32, device=self.device)
        if self.shape[1] >= (1 << 25):
            raise ValueError("more than 2^25 features are not supported by the tile-list key format")
        return self

    @classmethod
    def from_scipy(cls, m, device="cuda", binary=False):
        """Canonical CSR required (duplicates summed, indices sorted): scipy's .sum_duplicates() does both.
        binary: True = every stored entry is taken to be 1 (no value array: the kernels' implicit-ones path, less traffic
        and no value scatter); None = decided from the data (all stored values equal to 1, e.g. binarised peaks)."""
        import scipy.sparse as sp
        m = sp.csr_matrix(m)
        if not m.has_canonical_format:
            m = m.copy()
            m.sum_duplicates()
        if binary is None:
            binary = bool(m.data.size) and bool(np.all(m.data == 1))
        vals = None if binary else m.data
        return cls(m.indptr, m.indices, vals, m.shape, device)

    @classmethod
    def from_dense(cls, x, device="cuda", binary=False):
        """Dense tensor / array -> CSR.  NaN entries (missing labels, loss.py:24,38) are kept as explicit entries.
        binary=None: implicit ones when every non-zero entry is 1 (see from_scipy)."""
        import scipy.sparse as sp
        a = x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
        mask = (a != 0) | np.isnan(a)
        rows, cols = np.nonzero(mask)
        m = sp.csr_matrix((a[rows, cols].astype(np.float32), (rows, cols)), shape=a.shape)
        if binary is None:
            binary = bool(m.data.size) and bool(np.all(m.data == 1))
        return cls(m.indptr, m.indices, None if binary else m.data, m.shape, device)

    def with_explicit_values(self):
        """The same matrix with a value array (ones) where the entries are implicit ones; self otherwise."""
        if self.values is not None:
            return self
        import copy
        v = copy.copy(self)
        v.values = torch.ones(max(self.nnz, 1), dtype=torch.float32, device=self.device)
        return v

    # Optional input pre-transform (opt-in; the reference feeds raw counts): None, ("log1p", 1.0) or
    # ("log1p_norm", scale).  It is applied to every minibatch tile list built from this object, on the batch's own
    # entries; the resident CSR keeps the raw counts, and `.lib` of a tile list stays the raw library size.
    transform = None

    def with_transform(self, mode, scale=1e4):
        """A view of the same device arrays whose minibatches are pre-transformed: mode 'log1p' -> log1p(x),
        'log1p_norm' -> log1p(x * scale / library_size(cell)).  Use it for the INPUT role only; labels of the
        count likelihoods must stay raw counts."""
        if mode is None:
            return self
        if mode not in ("log1p", "log1p_norm"):
            raise ValueError("input transform must be None, 'log1p' or 'log1p_norm'")
        if self.values is None:
            raise ValueError("input transforms need explicit values (binary matrices with implicit ones have none)")
        if mode == "log1p_norm" and not scale > 0:
            raise ValueError("log1p_norm needs scale > 0")
        import copy
        v = copy.copy(self)
        v.transform = (mode, float(scale) if mode == "log1p_norm" else 1.0)
        return v

    def col_block(self, lo: int, hi: int, pad_row: bool = False):
        """Columns [lo, hi) of every row as a new DeviceCSR (column ids re-based to 0), built on the device.  The
        column-sharded view of a split that a rank holds when it owns the first-layer weights W[:, lo:hi) (feature-sharded
        data parallelism).  pad_row: one empty row is appended (its id, .n_rows - 1, pads per-rank slices of a global
        batch to a common length)."""
        lo, hi = int(lo), int(hi)
        if not 0 <= lo < hi <= self.shape[1]:
            raise ValueError("col_block: need 0 <= lo < hi <= n_cols")
        n = self.shape[0]
        if self.nnz:
            idx = self.indices[:self.nnz]
            keep = (idx >= lo) & (idx < hi)
            counts = torch.diff(self.indptr)
            row_of = torch.repeat_interleave(torch.arange(n, device=self.device), counts)
            new_counts = torch.bincount(row_of[keep], minlength=n)
            indices = (idx[keep] - lo).to(torch.int32)
            values = None if self.values is None else self.values[:self.nnz][keep]
            del row_of, keep
        else:
            new_counts = torch.zeros(n, dtype=torch.int64, device=self.device)
            indices = torch.zeros(0, dtype=torch.int32, device=self.device)
            values = None if self.values is None else torch.zeros(0, dtype=torch.float32, device=self.device)
        indptr = torch.zeros(n + 1 + (1 if pad_row else 0), dtype=torch.int64, device=self.device)
        indptr[1:n + 1] = torch.cumsum(new_counts, 0)
        if pad_row:
            indptr[n + 1] = indptr[n]
        out = DeviceCSR.from_device(indptr, indices, values, (n + (1 if pad_row else 0), hi - lo))
        out.transform = self.transform
        return out

    @classmethod
    def vstack(cls, parts):
        """Rows of `parts` (same column count, same device, all with or all without values) stacked into one DeviceCSR."""
        parts = list(parts)
        if len(parts) == 1:
            return parts[0]
        ncol = parts[0].shape[1]
        if any(p.shape[1] != ncol or (p.values is None) != (parts[0].values is None) for p in parts):
            raise ValueError("vstack: parts differ in columns or in having values")
        ips, off = [parts[0].indptr], int(parts[0].indptr_host[-1])
        for p in parts[1:]:
            ips.append(p.indptr[1:] + off)
            off += int(p.indptr_host[-1])
        indices = torch.cat([p.indices[:p.nnz] for p in parts])
        values = None if parts[0].values is None else torch.cat([p.values[:p.nnz] for p in parts])
        out = cls.from_device(torch.cat(ips), indices, values, (sum(p.shape[0] for p in parts), ncol))
        out.transform = parts[0].transform
        return out

    @property
    def n_rows(self):
        return self.shape[0]

    @property
    def n_cols(self):
        return self.shape[1]

    def max_batch_nnz(self, batch: int) -> int:
        """Upper bound of the entries of any minibatch of `batch` distinct rows (the sum of the `batch` longest rows):
        the fixed tile-list capacity of a training step captured in a HIP graph."""
        lens = np.diff(self.indptr_host)
        if batch >= lens.shape[0]:
            return int(lens.sum())
        return int(np.partition(lens, lens.shape[0] - batch)[lens.shape[0] - batch:].sum())

    def nnz_of(self, rows_host: np.ndarray) -> int:
        ip = self.indptr_host
        return int((ip[rows_host + 1] - ip[rows_host]).sum())

    def to_dense(self, rows=None):
        """Dense fp32 [len(rows), n_cols] tensor on the device (labels of the Gaussian / ZINB losses, small
        inference), densified on the GPU through the tile list."""
        rows = np.arange(self.shape[0]) if rows is None else np.asarray(rows, dtype=np.int64)
        out = []
        for r0 in range(0, len(rows), 16384):
            out.append(BatchTiles(self, rows[r0:r0 + 16384], want_lib=False).to_dense())
        if not out:
            return torch.zeros((0, self.shape[1]), dtype=torch.float32, device=self.device)
        return out[0] if len(out) == 1 else torch.cat(out, 0)


class BatchTiles:
    """Tile list of one minibatch of one modality (built on the GPU by scp_tiles_build).

    ptr : int32 [ncb*F + 2]   entries of (cell block cb, feature f) are key/val[ptr[cb*F+f] : ptr[cb*F+f+1]]
    key : int32 [nnz]         (feature << 6) | cell-in-block
    val : fp32  [nnz] or None (implicit ones)
    lib : fp32  [B]           row sums (library size, model.py:509-510)
    nnan: int32 [1]           number of NaN values (missing labels)
    """

    def __init__(self, csr: DeviceCSR, rows, want_lib=True, rows_host=None, capacity=None):
        """`capacity`: build without looking at the row ids on the host (a step captured in a HIP graph: `rows` is a
        device tensor whose CONTENTS change between replays): the entry buffers get `capacity` slots
        (>= csr.max_batch_nnz(len(rows))) and the caller guarantees distinct, in-range row ids."""
        L = _lib.lib()
        dev = csr.device
        if capacity is not None:
            if not (isinstance(rows, torch.Tensor) and rows.is_cuda and rows.dtype == torch.int64 and rows.is_contiguous()):
                raise ValueError("fixed-capacity tile lists need a contiguous int64 device tensor of row ids")
            rows_dev = rows
            self.B = int(rows.shape[0])
            self.nnz = int(capacity)
        else:
            if isinstance(rows, torch.Tensor):
                rows_dev = rows.to(device=dev, dtype=torch.int64).contiguous()
                if rows_host is None:
                    rows_host = rows.detach().cpu().numpy()
            else:
                rows_host = np.asarray(rows, dtype=np.int64)
                rows_dev = rows_to_device(rows_host, dev)
            rows_host = np.asarray(rows_host, dtype=np.int64)
            self.B = int(rows_host.shape[0])
            if self.B and (rows_host.min() < 0 or rows_host.max() >= csr.n_rows):
                raise IndexError("minibatch row id out of range")
            self.nnz = csr.nnz_of(rows_host) if self.B else 0
        self.F = csr.n_cols
        if self.B == 0:
            raise ValueError("empty minibatch")
        self.rows = rows_dev
        self.ptr = torch.empty(L.scp_tiles_ptr_len(self.B, self.F), dtype=torch.int32, device=dev)
        self.key = torch.empty(max(self.nnz, 1), dtype=torch.int32, device=dev)
        self.val = None if csr.values is None else torch.empty(max(self.nnz, 1), dtype=torch.float32, device=dev)
        # packed (bf16 value, low key bits) words: what the bf16 first-layer kernels read for value-carrying data
        self.pk = None if csr.values is None else torch.empty(max(self.nnz, 1), dtype=torch.int32, device=dev)
        tf = csr.transform
        need_lib = want_lib or (tf is not None and tf[0] == "log1p_norm")
        self.lib = torch.empty(self.B, dtype=torch.float32, device=dev) if need_lib else None
        self.nnan = torch.empty(1, dtype=torch.int32, device=dev)   # cleared by scp_tiles_build
        ws = torch.empty(L.scp_tiles_ws_len(self.B, self.F), dtype=torch.int32, device=dev)
        _lib.check(L.scp_tiles_build(_lib.ptr(csr.indptr), _lib.ptr(csr.indices), _lib.ptr(csr.values),
                                     _lib.ptr(rows_dev), self.B, self.F, _lib.ptr(self.ptr), _lib.ptr(self.key),
                                     _lib.ptr(self.val), _lib.ptr(self.pk), _lib.ptr(self.lib), _lib.ptr(self.nnan),
                                     _lib.ptr(ws), _lib.stream()), "scp_tiles_build")
        if tf is not None:
            _lib.check(L.scp_tiles_transform(self.B, self.F, _lib.ptr(self.ptr), _lib.ptr(self.key), _lib.ptr(self.val),
                                             _lib.ptr(self.pk), _lib.ptr(self.lib), 2 if tf[0] == "log1p_norm" else 1, tf[1],
                                             _lib.stream()), "scp_tiles_transform")
        self.csr = csr

    @property
    def shape(self):
        return (self.B, self.F)

    def to_dense(self):
        """Dense [B, F] fp32 tensor on the device (labels of the ZINB / Gaussian likelihoods, tiny inference)."""
        out = torch.empty((self.B, self.F), dtype=torch.float32, device=self.ptr.device)
        _lib.check(_lib.lib().scp_tiles_to_dense(self.B, self.F, _lib.ptr(self.ptr), _lib.ptr(self.key),
                                                 _lib.ptr(self.val), _lib.ptr(out), self.F, _lib.stream()),
                   "scp_tiles_to_dense")
        return out
