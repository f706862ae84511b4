#!/usr/bin/env python3
"""Golden fixture of the reference's training_split / split_data / check_distribution (utils.py:341-408), generated
by importing the reference in this container on a MiniAnnData stand-in (tests/util.py).  merge_paired_data needs the
real anndata package (absent here) and is not covered.  Writes tests/golden/data_split.npz."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.argv = [sys.argv[0], "--import-only"]
import make_goldens as MG  # noqa: E402  (registers the import stubs and imports the reference)
import scipy.sparse as sp  # noqa: E402
from tests.util import make_mini_adata  # noqa: E402

ref = MG.ref
out = {}
cases = {"plain": dict(fracs=[0.8, 0.1, 0.1], seed=0), "seed7": dict(fracs=[0.7, 0.2, 0.1], seed=7),
         "notest": dict(fracs=[0.75, 0.25, 0.0], seed=1), "grouped": dict(fracs=[0.6, 0.2, 0.2], seed=2, batch_key="labels"),
         "grouped2": dict(fracs=[0.5, 0.25, 0.25], seed=5, batch_key=["labels"])}
for tag, kw in cases.items():
    ad = make_mini_adata(seed=4, n=90)
    ad.obs = ad.obs.drop(columns=["scPair_split"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = ref.utils.training_split(ad, **kw)
    out[f"{tag}/split"] = np.array(list(res.obs["scPair_split"]))
    out[f"{tag}/index"] = np.array(list(res.obs.index))
ad = make_mini_adata(seed=4, n=90)
names = list(ad.obs.index)
res = ref.utils.training_split(ad, pre_split=(names[:50], names[50:60], names[70:]))
out["pre/split"] = np.array(list(res.obs["scPair_split"]))
# check_distribution: binarisation of a count matrix under 'ber' (sparse and dense)
for tag, sparse in (("sparse", True), ("dense", False)):
    ad = make_mini_adata(seed=5, n=30, sparse=sparse)
    ref.utils.check_distribution("Peaks", ad, "ber")
    X = ad.X.toarray() if sp.issparse(ad.X) else np.asarray(ad.X)
    out[f"ber_{tag}/X"] = X.astype(np.int64)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "data_split.npz"), **out)
print("wrote data_split.npz with", len(out), "arrays")
