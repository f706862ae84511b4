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


def check_distribution(name, obj, dist):
    """utils.py:341-353.  'ber': anything outside [0, 1] is binarised in place (X > 0); 'zinb'/'nb': negative
    values raise, non-integer dtype warns; 'gau': the reference raises when every value is positive (its test is
    `X.min() > 0`; kept as is)."""
    if dist == "ber":
        if obj.X.min() < 0 or obj.X.max() > 1:
            print(f"Bernoulli distribution was set for {name} data, but {name} data is not binary. Converting to binary...")
            obj.X = (obj.X > 0).astype(int)
    elif dist in ("zinb", "nb"):
        if obj.X.min() < 0:
            raise ValueError(f"(Zero-Inflated) Negative Binomial distribution was set for {name} data, but {name} data has negative values.")
        elif obj.X.dtype != "int":
            warnings.warn(f"(Zero-Inflated) Negative Binomial distribution was set for {name} data, but {name} data is not integer.")
    elif dist == "gau":
        if obj.X.min() > 0:
            raise ValueError(f"Gaussian distribution was set for {name} data, but {name} data has non-negative values.")


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
    """utils.py:356-366: tag each object's features with its modality name and distribution, check / binarise, and
    concatenate the two along the feature axis."""
    paired_obj_dict = dict(zip(modality_names, paired_objects))
    paired_dist_dict = dict(zip(modality_names, modality_distributions))
    for name, obj in paired_obj_dict.items():
        obj.var["modality"] = name
        print(f"processing modality: {name}")
        obj.var["modality_distribution"] = paired_dist_dict[name]
        check_distribution(name, obj, paired_dist_dict[name])
    a, b = paired_obj_dict[modality_names[0]], paired_obj_dict[modality_names[1]]
    try:
        import anndata
    except ImportError:
        anndata = None
    if anndata is not None and isinstance(a, anndata.AnnData):
        return anndata.concat([a.T, b.T], merge="same").T
    return _concat_features(a, b)


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
