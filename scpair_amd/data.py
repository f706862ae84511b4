"""Ingest helpers in front of data_loader_builder, with the reference's names and behaviour (utils.py:341-408):
distribution checks / binarisation, merging two single-modality objects into one paired object, and the
train / val / test split labels in `.obs['scPair_split']`.

They work on AnnData objects when anndata is installed and on any object with the same small surface
(`.X`, `.obs`, `.var`, `.copy()`), and keep `.X` sparse: nothing here densifies a count matrix (the reference's
`.min()/.max()` calls do not either)."""
from __future__ import annotations

import warnings

import numpy as np
import pandas as pd
import scipy.sparse as sp


def _value_range(X):
    """(min, max) of a dense or scipy-sparse matrix without densifying it."""
    return X.min(), X.max()


def _binary_problem(lo, hi):
    return lo < 0 or hi > 1


def check_distribution(name, obj, dist):
    """Counterpart of utils.py:341-353: validates one modality's matrix against the likelihood chosen for it.

      'ber'         values outside [0, 1] -> the matrix is replaced, in place, by its 0/1 indicator (X > 0), with a notice
      'zinb' / 'nb' any negative value is an error; a non-integer dtype only warns
      'gau'         an error when the smallest value is positive (the upstream test is `X.min() > 0`; kept, although its
                    message speaks of non-negative values)
    Other distribution names are not checked, as upstream."""
    lo, hi = _value_range(obj.X)
    family = {"ber": "Bernoulli", "zinb": "(Zero-Inflated) Negative Binomial", "nb": "(Zero-Inflated) Negative Binomial",
              "gau": "Gaussian"}.get(dist)
    if family is None:
        return
    head = f"{family} distribution was set for {name} data, but {name} data"
    if dist == "ber":
        if _binary_problem(lo, hi):
            print(f"{head} is not binary. Converting to binary...")
            obj.X = (obj.X > 0).astype(int)
        return
    if dist == "gau":
        if lo > 0:
            raise ValueError(f"{head} has non-negative values.")
        return
    if lo < 0:
        raise ValueError(f"{head} has negative values.")
    if obj.X.dtype != "int":
        warnings.warn(f"{head} is not integer.")


def _concat_features(a, b):
    """anndata.concat([a.T, b.T], merge='same').T for AnnData-like objects: cells = inner join of the two cell
    indices (order of `a`), features = a's then b's, `.var` = shared columns of both `.var`s, `.obs` = the columns
    that exist in both objects with identical values."""
    cells = a.obs.index[a.obs.index.isin(b.obs.index)]
    ia = a.obs.index.get_indexer(cells)
    ib = b.obs.index.get_indexer(cells)
    Xa, Xb = a.X[ia], b.X[ib]
    if sp.issparse(Xa) or sp.issparse(Xb):
        X = sp.hstack([sp.csr_matrix(Xa), sp.csr_matrix(Xb)], format="csr")
    else:
        X = np.concatenate([np.asarray(Xa), np.asarray(Xb)], axis=1)
    shared_var = [c for c in a.var.columns if c in b.var.columns]
    var = pd.concat([a.var[shared_var], b.var[shared_var]], axis=0)
    oa, ob = a.obs.iloc[ia], b.obs.iloc[ib]
    same = [c for c in oa.columns if c in ob.columns and oa[c].astype(object).equals(ob[c].astype(object).set_axis(oa.index))]
    return type(a)(X, oa[same].copy(), var)


def merge_paired_data(paired_objects=[None, None], modality_names=["Gene Expression", "Peaks"],
                      modality_distributions=["zinb", "ber"]):
    """Counterpart of utils.py:356-366: one paired object from two single-modality ones.  Each object's `.var` gets the
    columns 'modality' (its name) and 'modality_distribution', its matrix is validated / binarised by
    check_distribution, and the two are joined along the feature axis over their common cells."""
    if not (len(paired_objects) == len(modality_names) == len(modality_distributions) == 2):
        raise ValueError("merge_paired_data expects two objects, two modality names and two distributions")
    for obj, name, dist in zip(paired_objects, modality_names, modality_distributions):
        print(f"processing modality: {name}")
        obj.var["modality"] = name
        obj.var["modality_distribution"] = dist
        check_distribution(name, obj, dist)
    first, second = paired_objects
    try:
        import anndata
    except ImportError:
        anndata = None
    if anndata is not None and isinstance(first, anndata.AnnData):
        return anndata.concat([first.T, second.T], merge="same").T
    return _concat_features(first, second)


def _sample_per_group(frame, batch_key, frac, seed):
    """Cell names drawn group by group: `frac` of every batch_key group, pandas sampling with the given seed."""
    picked = frame.groupby(batch_key).apply(lambda grp: grp.sample(frac=frac, random_state=seed))
    return picked["cell_index"].tolist()


def split_data(scobj, train_frac, val_frac, test_frac, seed, batch_key=None):
    """utils.py:369-384 -> (train, val, test) cell names.  Without a batch key: sklearn's train_test_split on the
    cell index, train vs rest first, then the rest into val / test (all of it to val when test_frac is 0).  With a
    batch key (a column name or a list of them): the fractions are drawn inside every group; whatever is left after
    train and val is the test set."""
    held_out = val_frac + test_frac
    if batch_key is not None:
        frame = scobj.obs.copy()
        frame["cell_index"] = frame.index.tolist()
        train_set = _sample_per_group(frame, batch_key, train_frac, seed)
        rest = frame.loc[~frame.index.isin(train_set)]
        val_set = _sample_per_group(rest, batch_key, val_frac / held_out, seed)
        test_set = rest.loc[~rest.index.isin(val_set)]["cell_index"].tolist()
        return train_set, val_set, test_set
    from sklearn.model_selection import train_test_split
    train_set, rest = train_test_split(scobj.obs.index, test_size=held_out, random_state=seed)
    if test_frac == 0:
        return train_set, rest, []
    val_set, test_set = train_test_split(rest, test_size=test_frac / held_out, random_state=seed)
    return train_set, val_set, test_set


def training_split(scobj, fracs=[0.8, 0.1, 0.1], seed=0, batch_key=None, pre_split=None):
    """utils.py:387-408: label every cell in `.obs['scPair_split']` ('train' / 'val' / 'test', 'TBD' for cells in
    none of the sets) and return a copy of the object.  `pre_split` = (train, val, test) name lists skips the draw.
    As upstream, numpy's global RNG is seeded, and when `.obs` cannot take the labels only the table is returned."""
    assert np.isclose(np.sum(fracs), 1, atol=1e-9), "train_frac + val_frac + test_frac should be approximately 1"
    assert len(fracs) == 3, "fracs should be a list of 3 fractions [train_frac, val_frac, test_frac]"
    assert min(fracs) >= 0, "train_frac, val_frac, and test_frac should be >= 0"
    np.random.seed(seed)
    if pre_split is None:
        print("No pre-split data provided. Splitting data based on provided fractions...")
        pre_split = split_data(scobj, fracs[0], fracs[1], fracs[2], seed, batch_key)
    labelled = scobj.obs.copy()
    labelled["scPair_split"] = "TBD"
    for label, cells in zip(("train", "val", "test"), pre_split):
        labelled.loc[cells, "scPair_split"] = label
    if not scobj.obs.index.equals(labelled.index):
        print("The provided metadata does not match the input AnnData object. Returning the metadata only...")
        return labelled
    scobj.obs = labelled.copy()
    return scobj.copy()
