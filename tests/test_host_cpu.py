"""Host-side logic that needs no GPU: reference-identical construction/initialisation, parameter names,
weight-decay grouping, batch order, argument validation, and the no-CPU-fallback rule."""
import json
import os

import numpy as np
import pytest
import torch
from torch import nn

import scpair_amd as S
from tests import util as U


def _obj(mods, zi, **kw):
    return S.scPair_object(None, modalities=mods, cov=None, hidden_layer=[16, 8], dropout_rate=0.0, device="cpu",
                           save_model=False, zero_inflated=zi, verbose=False, **kw)


@pytest.mark.parametrize("name,mods,zi", [("traj_zinb_ber", {"Gene Expression": "zinb", "Peaks": "ber"}, True),
                                          ("traj_nb_ber", {"Gene Expression": "nb", "Peaks": "ber"}, False)])
def test_initial_weights_equal_reference(name, mods, zi):
    """Same construction + seeding order as model.py:620-694 => bit-identical initial state_dicts."""
    d = U.load(name)
    o = _obj(mods, zi)
    for di, (i, oo, a, b) in enumerate([("Gene Expression", "Peaks", 64, 128), ("Peaks", "Gene Expression", 128, 64)]):
        enc, dec, _ = o._build_predictor(i, oo, a, b, 0)
        gold_e, gold_d = U.sub(d, f"dir{di}/enc_init", tensor=False), U.sub(d, f"dir{di}/dec_init", tensor=False)
        assert sorted(enc.state_dict()) == sorted(gold_e) and sorted(dec.state_dict()) == sorted(gold_d)
        for k, v in enc.state_dict().items():
            np.testing.assert_array_equal(v.numpy(), gold_e[k])
        for k, v in dec.state_dict().items():
            np.testing.assert_array_equal(v.numpy(), gold_d[k])


def test_batch_order_equals_reference():
    """The row-id DataLoader consumes the global torch RNG like the reference's DataLoader (model.py:712-717)."""
    from scpair_amd.model import _index_loader
    d = U.load("traj_nb_ber")
    c = U.cfg(d, "cfg")
    gold = U.split_batches(d["dir0/batch_rows"], d["dir0/batch_sizes"])
    S.set_seed(0)
    loader = _index_loader(len(d["emb/Gene Expression_train"]), c["batch"])
    got = []
    for _ in range(c["epochs"]):
        got.extend(b[0].numpy() for b in loader)
    assert len(got) == len(gold)
    for a, b in zip(got, gold):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("n,bs,drop_last", [(1000, 130, False), (1000, 130, True), (7, 3, False), (5, 8, False)])
def test_index_loader_is_the_shuffling_dataloader(n, bs, drop_last):
    """_index_loader yields slices of the sampler's permutation without the DataLoader's per-row work; batches, their order
    over several epochs, and the state the global torch RNG is left in are those of DataLoader(TensorDataset(arange(n)),
    batch_size, shuffle=True), which is what the reference builds (model.py:713-717)."""
    from scpair_amd.model import _index_loader
    ds = torch.utils.data.TensorDataset(torch.arange(n))
    torch.manual_seed(11)
    ref = torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=True, num_workers=0, drop_last=drop_last)
    want = [[b[0].numpy().copy() for b in ref] for _ in range(3)]
    rng_want = torch.get_rng_state()
    from scpair_amd.model import _FastIndexLoader, _dataloader_over_indices, _fast_loader_matches_dataloader
    assert _fast_loader_matches_dataloader() and isinstance(_index_loader(n, bs, drop_last), _FastIndexLoader)
    for make in (_index_loader, _dataloader_over_indices):        # the lean iterator in use, and the fall-back behind it
        torch.manual_seed(11)
        mine = make(n, bs, drop_last)
        got = [[b[0].numpy().copy() for b in mine] for _ in range(3)]
        assert bool((torch.get_rng_state() == rng_want).all())
        for e_want, e_got in zip(want, got):
            assert len(e_want) == len(e_got)
            for a, b in zip(e_want, e_got):
                assert a.dtype == b.dtype
                np.testing.assert_array_equal(a, b)


def test_all_ones_matrices_are_held_without_values():
    """The trainers' data seam (model._to_csr) keeps a matrix whose stored entries are all 1 (binarised peaks) without a value
    array, so that it runs the kernels' implicit-ones path; counts, explicit zeros and NaN labels keep their values."""
    import scipy.sparse as sp
    from scpair_amd.model import _to_csr
    rng = np.random.default_rng(0)
    ones = sp.random(50, 40, density=0.1, format="csr", random_state=1, dtype=np.float32)
    ones.data[:] = 1.0
    assert _to_csr(ones, "cpu").values is None
    assert _to_csr(torch.tensor(ones.toarray()), "cpu").values is None
    assert S.DeviceCSR.from_scipy(ones, "cpu").values is not None          # the explicit constructor keeps its default
    counts = ones.copy()
    counts.data[3] = 2.0
    assert _to_csr(counts, "cpu").values is not None
    nan = ones.copy()
    nan.data[5] = np.nan
    assert _to_csr(nan, "cpu").values is not None
    dense_nan = ones.toarray()
    dense_nan[0, 0] = np.nan
    assert _to_csr(dense_nan, "cpu").values is not None
    empty = sp.csr_matrix((4, 7), dtype=np.float32)
    assert _to_csr(empty, "cpu").values is not None                        # nothing stored: nothing to decide from
    c = _to_csr(ones, "cpu")
    e = c.with_explicit_values()
    assert e.values is not None and float(e.values.sum()) == c.nnz and e.indices is c.indices
    assert c.with_explicit_values().with_transform("log1p").transform == ("log1p", 1.0)
    del rng


def test_checkpoint_snapshot_equals_deepcopy():
    """model._snapshot: the epoch loops' best-so-far copy -- a deep copy the first time, overwritten in place afterwards:
    the values of a deep copy (parameters and buffers, BatchNorm's counters included), no aliasing with the live module."""
    import copy
    from scpair_amd.model import _snapshot
    torch.manual_seed(0)
    live = nn.Sequential(nn.Linear(6, 5), nn.BatchNorm1d(5), nn.LeakyReLU(), nn.Linear(5, 3))
    live.train()
    live(torch.randn(8, 6))
    snap = _snapshot(live)
    ids = [id(p) for p in snap.parameters()]
    assert all(a is not b and torch.equal(a, b) for a, b in zip(snap.state_dict().values(), live.state_dict().values()))
    with torch.no_grad():
        for p in live.parameters():
            p.add_(1.0)
    live(torch.randn(8, 6))
    want = copy.deepcopy(live)
    assert not torch.equal(snap[0].weight, live[0].weight)                  # the snapshot did not follow the live module
    snap2 = _snapshot(live.eval(), snap)
    assert snap2 is snap and [id(p) for p in snap.parameters()] == ids and not snap.training
    for (k, a), (_, b) in zip(snap.state_dict().items(), want.state_dict().items()):
        assert torch.equal(a, b), k
    other = nn.Sequential(nn.Linear(6, 4))
    assert _snapshot(other, snap) is not snap                               # another architecture: a fresh deep copy


def test_weight_decay_groups_equal_reference():
    with open(os.path.join(U.GOLD, "weight_decay_groups.json")) as f:
        groups = json.load(f)
    H, G, P = [16, 8], 32, 64
    enc = S.Input_Module(input_dim=G, input_batch_num=0, hidden_layer=H, activation=nn.LeakyReLU(), layernorm=True,
                         batchnorm=True, dropout_rate=0.1, add_linear_layer=True, infer_library_size=True)
    raw = S.Output_Module_Raw(output_dim=G, output_batch_num=0, hidden_layer=H, infer_library_size=False,
                              sample_factor=True, feature_factor=True, zero_inflated=True, dispersion="feature-cell")
    ber = S.Output_Module_Ber(output_dim=P, output_batch_num=0, hidden_layer=H, infer_library_size=True,
                              sample_factor=False, feature_factor=True)
    gau = S.Output_Module_Gau(output_dim=G, output_batch_num=0, hidden_layer=H, infer_library_size=False, feature_factor=True)
    mp = S.Module_Module(origin_module_dim=8, target_module_dim=8, translational_hidden_nodes=[12], non_neg=False,
                         layernorm=True, activation=nn.LeakyReLU(), batchnorm=False, dropout_rate=0)
    for key, mod, ol in [("Input_Module", enc, ["output"]), ("Output_Module_Raw", raw, ["output"]),
                         ("Output_Module_Ber", ber, ["output"]), ("Output_Module_Gau", gau, ["output"]),
                         ("Module_Module[bias]", mp, "bias")]:
        _, _, dn, ndn = S.add_weight_decay(mod, output_layer=ol)
        assert dn == groups[key]["decay"] and ndn == groups[key]["nodecay"], key
        assert {k: v.ndim for k, v in mod.named_parameters()} == groups[key]["ndim"]


def test_module_parameter_names_match_goldens():
    d = U.load("mod_vae")
    enc = S.VAE_ENCODER(32, 0, [16], 6, infer_library_size=True)
    assert sorted(enc.state_dict()) == sorted(U.sub(d, "c1/enc_sd"))
    dec = S.VAE_DECODER(32, 0, [16], 6, infer_library_size=False, distribution="zinb", dispersion="feature-cell")
    assert sorted(dec.state_dict()) == sorted(k for k in U.sub(d, "c4/dec_sd"))
    m = U.load("mod_gau_map")
    mp = S.Module_Module(8, 8, [12, 10], non_neg=False, layernorm=False, batchnorm=True, activation=nn.LeakyReLU())
    assert sorted(mp.state_dict()) == sorted(U.sub(m, "map3/sd"))


def test_constructor_validation_like_reference():
    with pytest.raises(ValueError):
        S.scPair_object(None, add_linear_layer=True, mapping_non_neg=True, device="cpu")
    with pytest.raises(ValueError):
        S.scPair_object(None, sample_factor_rna=False, infer_library_size_rna=False, device="cpu")
    with pytest.raises(ValueError):
        S.scPair_object(None, sample_factor_rna=True, infer_library_size_rna=True, device="cpu")
    dec = S.Output_Module_Raw(8, 0, [4], sample_factor=False)
    with pytest.raises(Exception, match="library size is missing"):
        dec.fused_loss(torch.zeros(2, 4), None, None, None, None, "nb")
    with pytest.raises(NotImplementedError):
        S.FCNN([4, 4], activation=nn.Tanh())(torch.zeros(1, 4))


def test_no_cpu_fallback():
    o = _obj({"Gene Expression": "nb", "Peaks": "ber"}, False)
    with pytest.raises(RuntimeError, match="GPU"):
        o.train_predicting_networks({}, {}, {})
    with pytest.raises(RuntimeError):
        S.scaled_loss(torch.zeros(3), torch.zeros(3))
    src = open(os.path.join(os.path.dirname(S.__file__), "ops.py")).read() + \
        open(os.path.join(os.path.dirname(S.__file__), "model.py")).read()
    assert "oracle" not in src, "the product must never import the oracle"


def test_synthetic_generator_is_deterministic():
    a = S.synth.make_paired(64, 20, 30, seed=3, p_atac=0.1) if hasattr(S, "synth") else None
    from scpair_amd import synth
    a, b = synth.make_paired(64, 20, 30, seed=3, p_atac=0.1), synth.make_paired(64, 20, 30, seed=3, p_atac=0.1)
    assert (a["rna"] != b["rna"]).nnz == 0 and (a["atac"] != b["atac"]).nnz == 0
    assert a["rna"].has_canonical_format and a["atac"].has_canonical_format
    assert len(set(a["train"]) | set(a["val"]) | set(a["test"])) == 64


@pytest.mark.parametrize("tag,cov", [("cov", ["labels", "depth"]), ("nocov", None)])
def test_data_loader_builder_matches_reference(tag, cov):
    """data_loader_builder on an AnnData-like object (model.py:450-516): same keys, matrices (kept sparse here),
    covariate blocks, library sizes and metadata as the reference's builder (golden: tools/make_golden_loader.py)."""
    import scipy.sparse as sp
    import scpair_amd as S
    from tests.util import load, make_mini_adata
    gold = load("data_loader")
    ad = make_mini_adata(seed=3)
    obj = S.scPair_object(scobj=ad, cov=cov, modalities={"Gene Expression": "zinb", "Peaks": "ber"},
                          device=torch.device("cpu"), save_model=False)
    tr, va, te = obj.data_loader_builder()
    n_checked = 0
    for split, d in (("train", tr), ("val", va), ("test", te)):
        for k, v in d.items():
            gk = f"{tag}/{k}"
            if v is None:
                assert gk not in gold
                continue
            got = np.asarray(v.todense()) if sp.issparse(v) else v.numpy()
            np.testing.assert_allclose(got, gold[gk], rtol=1e-6, atol=1e-6, err_msg=gk)
            if k.endswith("_mtx"):
                assert sp.issparse(v) and v.has_sorted_indices
            n_checked += 1
        meta = getattr(obj, f"{split}_metadata")
        assert list(meta.index) == list(gold[f"{tag}/{split}_index"])
        assert [str(c) for c in meta.columns] == list(gold[f"{tag}/{split}_columns"])
    assert n_checked == len([k for k in gold.files if k.startswith(tag + "/") and not k.endswith(("_index", "_columns"))])


def test_training_split_matches_reference():
    """training_split / split_data (utils.py:369-408) against labels the reference produced on the same object."""
    import warnings
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "data_split.npz"))
    cases = {"plain": dict(fracs=[0.8, 0.1, 0.1], seed=0), "seed7": dict(fracs=[0.7, 0.2, 0.1], seed=7),
             "notest": dict(fracs=[0.75, 0.25, 0.0], seed=1),
             "grouped": dict(fracs=[0.6, 0.2, 0.2], seed=2, batch_key="labels"),
             "grouped2": dict(fracs=[0.5, 0.25, 0.25], seed=5, batch_key=["labels"])}
    for tag, kw in cases.items():
        ad = U.make_mini_adata(seed=4, n=90)
        ad.obs = ad.obs.drop(columns=["scPair_split"])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = S.training_split(ad, **kw)
        assert list(res.obs.index) == list(d[f"{tag}/index"])
        assert list(res.obs["scPair_split"]) == list(d[f"{tag}/split"]), tag
        assert list(ad.obs["scPair_split"]) == list(d[f"{tag}/split"])         # written in place too, as upstream
    ad = U.make_mini_adata(seed=4, n=90)
    names = list(ad.obs.index)
    res = S.training_split(ad, pre_split=(names[:50], names[50:60], names[70:]))
    assert list(res.obs["scPair_split"]) == list(d["pre/split"])
    assert set(res.obs["scPair_split"][60:70]) == {"TBD"}
    with pytest.raises(AssertionError):
        S.training_split(ad, fracs=[0.5, 0.1, 0.1])


def test_check_distribution_and_merge():
    """check_distribution (utils.py:341-353) against the reference's binarised matrices; merge_paired_data on two
    single-modality objects (the reference needs the anndata package for this one: parity unpinned, the expected
    result is stated from anndata.concat's documented inner-join / merge='same' semantics)."""
    import pandas as pd
    import scipy.sparse as sp
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "data_split.npz"))
    for tag, sparse in (("sparse", True), ("dense", False)):
        ad = U.make_mini_adata(seed=5, n=30, sparse=sparse)
        S.check_distribution("Peaks", ad, "ber")
        assert sp.issparse(ad.X) == sparse                               # stays sparse
        X = ad.X.toarray() if sparse else np.asarray(ad.X)
        assert np.array_equal(X.astype(np.int64), d[f"ber_{tag}/X"])
    neg = U.make_mini_adata(seed=5, n=10, sparse=False); neg.X = neg.X - 1.0
    with pytest.raises(ValueError):
        S.check_distribution("Gene Expression", neg, "zinb")
    pos = U.make_mini_adata(seed=5, n=10, sparse=False); pos.X = pos.X + 1.0
    with pytest.raises(ValueError):
        S.check_distribution("x", pos, "gau")
    with pytest.warns(UserWarning):
        S.check_distribution("Gene Expression", U.make_mini_adata(seed=5, n=10), "nb")
    # merge: RNA counts + ATAC counts (binarised by the check), one cell missing from the second object
    rng = np.random.default_rng(0)
    n, g, p = 12, 5, 7
    cells = [f"c{i}" for i in range(n)]
    rna = U.MiniAnnData(sp.csr_matrix(rng.poisson(1.0, (n, g)).astype(np.float32)),
                        pd.DataFrame({"donor": ["a", "b"] * 6, "qc": np.arange(n)}, index=cells),
                        pd.DataFrame({"chrom": ["1"] * g}, index=[f"g{i}" for i in range(g)]))
    keep = [i for i in range(n) if i != 3]
    atac_counts = rng.poisson(0.4, (n, p)).astype(np.float32)
    atac = U.MiniAnnData(sp.csr_matrix(atac_counts[keep]),
                         pd.DataFrame({"donor": (["a", "b"] * 6)[:3] + (["a", "b"] * 6)[4:], "qc": np.arange(11) * 2},
                                      index=[cells[i] for i in keep]),
                         pd.DataFrame({"chrom": ["2"] * p, "width": 500}, index=[f"p{i}" for i in range(p)]))
    rna_X = rna.X.toarray()
    with pytest.warns(UserWarning):
        m = S.merge_paired_data([rna, atac], ["Gene Expression", "Peaks"], ["zinb", "ber"])
    assert sp.issparse(m.X) and m.X.shape == (n - 1, g + p)
    assert list(m.obs.index) == [cells[i] for i in keep]
    assert list(m.obs.columns) == ["donor"]                              # 'qc' differs between the two objects
    assert list(m.var["modality"]) == ["Gene Expression"] * g + ["Peaks"] * p
    assert list(m.var["modality_distribution"]) == ["zinb"] * g + ["ber"] * p
    assert list(m.var.columns) == ["chrom", "modality", "modality_distribution"]
    M = m.X.toarray()
    assert np.array_equal(M[:, :g], rna_X[keep]) and np.array_equal(M[:, g:], (atac_counts[keep] > 0))
