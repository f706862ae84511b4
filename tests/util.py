"""Shared helpers for the tests: golden loading and state-dict extraction."""
import json
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def sub(d, prefix, tensor=True):
    """All entries of npz `d` under 'prefix/' as {rest: tensor}."""
    out = {}
    p = prefix + "/"
    for k in d.files:
        if k.startswith(p):
            v = d[k]
            out[k[len(p):]] = torch.tensor(v) if tensor else v
    return out


def opt(d, key):
    return torch.tensor(d[key]) if key in d.files else None


def cfg(d, key):
    return json.loads(str(d[key]))


def split_batches(rows, sizes):
    out, o = [], 0
    for s in sizes:
        out.append(np.asarray(rows[o:o + s]))
        o += s
    return out


class MiniAnnData:
    """The slice of the AnnData interface scPair's data_loader_builder touches (model.py:450-516): `.X`, `.obs`,
    `.var`, `.copy()`, boolean row indexing `data[mask]` and column indexing `data[:, mask]`.  anndata itself is not
    installed in the image; this stand-in lets the reference's own builder run for the golden fixture and the
    build's builder run in the tests on the same object."""

    def __init__(self, X, obs, var):
        self.X, self.obs, self.var = X, obs, var

    def copy(self):
        return MiniAnnData(self.X.copy(), self.obs.copy(), self.var.copy())

    def __getitem__(self, idx):
        import numpy as np
        if isinstance(idx, tuple):
            rows, cols = idx
        else:
            rows, cols = idx, slice(None)
        r = np.arange(self.X.shape[0]) if isinstance(rows, slice) else np.flatnonzero(np.asarray(rows))
        c = np.arange(self.X.shape[1]) if isinstance(cols, slice) else np.flatnonzero(np.asarray(cols))
        return MiniAnnData(self.X[r][:, c], self.obs.iloc[r], self.var.iloc[c])


def make_mini_adata(seed=0, n=60, g=12, p=20, sparse=True):
    """Small paired object: counts + binary peaks, a categorical and a numeric covariate, 70/15/15 split."""
    import numpy as np
    import pandas as pd
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    rna = rng.poisson(0.6, size=(n, g)).astype(np.float32)
    atac = (rng.uniform(size=(n, p)) < 0.15).astype(np.float32)
    X = np.concatenate([rna, atac], 1)
    # interleave the feature order so the modality masks are not contiguous
    perm = rng.permutation(g + p)
    X = X[:, perm]
    modality = np.array(["Gene Expression"] * g + ["Peaks"] * p)[perm]
    split = rng.choice(["train", "val", "test"], size=n, p=[0.7, 0.15, 0.15])
    obs = pd.DataFrame({"scPair_split": split,
                        "labels": pd.Categorical(rng.choice(["a", "b", "c"], size=n)),
                        "depth": rng.normal(5.0, 2.0, size=n)},
                       index=[f"cell{i}" for i in range(n)])
    var = pd.DataFrame({"modality": modality}, index=[f"f{i}" for i in range(g + p)])
    return MiniAnnData(sp.csr_matrix(X) if sparse else X, obs, var)
