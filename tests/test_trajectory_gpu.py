"""End-to-end parity on the GPU: the reference's own training loops (recorded in tests/golden/traj_*.npz by
tools/make_goldens.py) against scpair_amd on the same synthetic inputs, seeds and initialisation.

Tolerances:
  fp32 path : per-step training loss rel 1e-5, per-epoch validation loss rel 1e-4, embeddings max-abs 1e-3 (the values
              SURVEY 8d states; measured on MI355X: <= 3e-6, <= 2.4e-6, <= 4e-4 with BatchNorm and <= 3.1e-5 without;
              the test prints them)
  bf16 path : losses rel 2e-2, embeddings cosine median >= 0.999, 5th percentile >= 0.99 and min >= 0.99 (0.95 with
              BatchNorm; 8-dim toy embeddings)
"""
import numpy as np
import pytest
import torch

import scpair_amd as S
from scpair_amd import synth
from tests import util as U

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _dicts(c, d):
    data = synth.make_paired(c["n"], c["g"], c["p"], seed=c["seed"], p_atac=c["p_atac"])
    out = []
    for split in ("train", "val", "test"):
        idx = data[split]
        dd = {}
        for name, mat in (("Gene Expression", data["rna"]), ("Peaks", data["atac"])):
            dd[f"{name}_{split}_mtx"] = mat[idx]
            dd[f"{name}_{split}_cov"] = torch.tensor(d["cov"][idx]) if c["with_cov"] else None
        out.append(dd)
    return out


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.abs(b)))


def _run(name, precision, **kw):
    d = U.load(name)
    c = U.cfg(d, "cfg")
    tr, va, te = _dicts(c, d)
    args = dict(cov=(["labels"] if c["with_cov"] else None), modalities=c["modalities"], SEED=0,
                hidden_layer=c["hidden"], dropout_rate=0.0, batchnorm=c["batchnorm"], layernorm=True,
                batch_size=c["batch"], max_epochs=c["epochs"], device=DEV, save_model=False,
                zero_inflated=c["zero_inflated"], use_mixed_precision=(precision == "bf16"), verbose=False)
    if "patience" in c:        # epoch-control fixtures (tools/make_goldens.py `epoch`)
        args.update(early_stopping_patience=c["patience"], weight_decay=c["scheduler"],
                    learning_rate_prediction=c["lr"], mapping_learning_rate=c["lr"])
    args.update(kw)
    obj = S.scPair_object(None, **args)
    obj.train_predicting_networks(tr, va, te)
    return d, c, obj


def test_trajectory_with_captured_steps_is_bitwise_the_eager_one(monkeypatch):
    """(experimental path: refused unless SCPAIR_EXPERIMENTAL_GRAPH is set, which is checked first)
    scPair_object(capture_step=True): full-size minibatch steps replayed from a HIP graph, the ragged last batch of
    each epoch eager -- the same training and validation losses, bit for bit, as the all-eager run."""
    monkeypatch.delenv("SCPAIR_EXPERIMENTAL_GRAPH", raising=False)
    with pytest.raises(ValueError):
        S.scPair_object(None, cov=None, device=DEV, save_model=False, capture_step=True)
    monkeypatch.setenv("SCPAIR_EXPERIMENTAL_GRAPH", "1")
    _, c, eager = _run("traj_nb_ber_bn", "fp32")
    _, _, graph = _run("traj_nb_ber_bn", "fp32", capture_step=True)
    for k in eager.history:
        assert eager.history[k]["train_loss"] == graph.history[k]["train_loss"], k
        assert eager.history[k]["val_loss"] == graph.history[k]["val_loss"], k


@pytest.mark.parametrize("name,precision", [("traj_nb_ber", "fp32"), ("traj_nb_ber_bn", "bf16"), ("traj_zinb_ber", "bf16")])
def test_directions_side_by_side_equal_one_after_the_other(name, precision):
    """The trainer runs the two predictors interleaved on two streams by default (scPair_object(concurrent_directions=True));
    per-step training losses, per-epoch validation losses and the returned weights are bit for bit those of the reference's
    order -- direction 0 to completion, then direction 1 (model.py:612) -- because each direction keeps its own stream of
    batch draws (dropout 0 here: with dropout only the numbering of the mask seeds differs)."""
    _, _, side = _run(name, precision)
    rng_side = torch.get_rng_state()
    _, _, seq = _run(name, precision, concurrent_directions=False)
    assert bool((torch.get_rng_state() == rng_side).all())          # the global RNG is left in the same state, too
    assert list(side.history) == list(seq.history) and list(side.encoder_dict) == list(seq.encoder_dict)
    for k in seq.history:
        assert side.history[k]["train_loss"] == seq.history[k]["train_loss"], k
        assert side.history[k]["val_loss"] == seq.history[k]["val_loss"], k
    for k in seq.encoder_dict:
        for (n1, a), (n2, b) in zip(side.encoder_dict[k].state_dict().items(), seq.encoder_dict[k].state_dict().items()):
            assert n1 == n2 and torch.equal(a, b), (k, n1)
        for (n1, a), (n2, b) in zip(side.decoder_dict[k].state_dict().items(), seq.decoder_dict[k].state_dict().items()):
            assert n1 == n2 and torch.equal(a, b), (k, n1)


@pytest.mark.parametrize("name", ["traj_nb_ber", "traj_zinb_ber", "traj_nb_ber_cov", "traj_nb_ber_bn"])
def test_trajectory_fp32(name):
    d, c, obj = _run(name, "fp32")
    names = list(c["modalities"].keys())
    for di in range(2):
        h = obj.history[names[di] + " to " + names[1 - di]]
        assert len(h["train_loss"]) == len(d[f"dir{di}/train_loss"])
        print(f"[measured] {name} dir{di}: train loss rel {_rel(h['train_loss'], d[f'dir{di}/train_loss']):.2e}, "
              f"val loss rel {_rel(h['val_loss'], d[f'dir{di}/val_loss']):.2e}", flush=True)
        assert _rel(h["train_loss"], d[f"dir{di}/train_loss"]) < 1e-5, (di, _rel(h["train_loss"], d[f"dir{di}/train_loss"]))
        assert _rel(h["val_loss"], d[f"dir{di}/val_loss"]) < 1e-4, (di, _rel(h["val_loss"], d[f"dir{di}/val_loss"]))
    emb, _ = obj.reference_embeddings()
    for k, v in emb.items():
        print(f"[measured] {name} embedding {k}: max abs {float((v - torch.tensor(d['emb/' + k])).abs().max()):.2e}", flush=True)
        assert float((v - torch.tensor(d["emb/" + k])).abs().max()) < 1e-3, k
    if name in ("traj_nb_ber", "traj_zinb_ber"):
        obj.train_bidirectional_mapping()
        for mi in range(2):
            h = obj.history[names[mi] + "_to_" + names[1 - mi]]
            assert _rel(h["val_loss"], d[f"map{mi}/val_loss"]) < 1e-3
        memb, _ = obj.mapped_embeddings()
        for k, v in memb.items():
            assert float((v - torch.tensor(d["memb/" + k])).abs().max()) < 5e-3, k
        pred = obj.predict()
        for k, v in pred.items():
            g = d["pred/" + k]
            assert v.shape == g.shape
            assert float(np.abs(v - g).max()) < 2e-3 * max(1.0, float(np.abs(g).max())), k
        # atlas-scale output options: bf16 predictions, top-k per cell
        p16 = obj.predict(out_dtype=torch.bfloat16)
        top = obj.predict(topk=5)
        for k, v in pred.items():
            assert p16[k].dtype == torch.bfloat16 and torch.equal(p16[k], torch.tensor(v).bfloat16()), k
            tv, ti = torch.topk(torch.tensor(v), 5, dim=1)
            assert np.array_equal(top[k]["values"], tv.numpy()), k
            assert np.array_equal(np.take_along_axis(v, top[k]["indices"], 1), tv.numpy()), k


@pytest.mark.parametrize("name", ["traj_nb_ber", "traj_zinb_ber", "traj_nb_ber_bn"])
def test_trajectory_bf16(name):
    d, c, obj = _run(name, "bf16")
    names = list(c["modalities"].keys())
    for di in range(2):
        h = obj.history[names[di] + " to " + names[1 - di]]
        assert _rel(h["train_loss"], d[f"dir{di}/train_loss"]) < 2e-2
        assert _rel(h["val_loss"], d[f"dir{di}/val_loss"]) < 2e-2
    emb, _ = obj.reference_embeddings()
    for k, v in emb.items():
        g = torch.tensor(d["emb/" + k])
        cos = torch.nn.functional.cosine_similarity(v, g, dim=1)
        # BatchNorm (batch statistics of bf16-rounded activations) widens the tail: a few of the 8-dim toy embeddings
        # land at 0.96-0.97; 5th percentile and median stay where the other configurations are
        floor = 0.95 if name == "traj_nb_ber_bn" else 0.99
        assert float(cos.median()) > 0.999 and float(cos.quantile(0.05)) > 0.99 and float(cos.min()) > floor, \
            (k, float(cos.median()), float(cos.quantile(0.05)), float(cos.min()))
    S.set_precision("fp32")


def test_vae_trajectory_fp32():
    """train_VAE / eval_VAE (utils.py:179-228) replayed with the recorded batch order and noise stream."""
    d = U.load("traj_vae_nb")
    c = U.cfg(d, "cfg")
    S.set_precision("fp32")
    rna = synth.make_rna(c["n"], c["g"], seed=c["seed"])
    csr = S.DeviceCSR.from_scipy(rna, DEV)
    enc = S.VAE_ENCODER(c["g"], 0, c["hid"], c["L"], layernorm=True, dropout_rate=0, infer_library_size=False)
    dec = S.VAE_DECODER(c["g"], 0, c["hid"], c["L"], layernorm=True, dropout_rate=0, infer_library_size=False,
                        sample_factor=True, feature_factor=False, distribution="nb", dispersion="feature-cell")
    enc.load_state_dict({k: torch.tensor(v) for k, v in U.sub(d, "enc_init", tensor=False).items()})
    dec.load_state_dict({k: torch.tensor(v) for k, v in U.sub(d, "dec_init", tensor=False).items()})
    enc.to(DEV); dec.to(DEV)
    opt = S.FusedAdamW(list(enc.parameters()) + list(dec.parameters()), lr=1e-3, weight_decay=1e-9)
    eps_all = torch.tensor(d["eps"]).to(DEV)
    pos = [0]

    def eps_source(shape, device):
        n = int(np.prod(shape))
        e = eps_all[pos[0]:pos[0] + n].reshape(shape)
        pos[0] += n
        return e
    enc.eps_source = eps_source
    batches = U.split_batches(d["rows"], d["batch_sizes"])
    per_epoch = len(batches) // c["epochs"]
    all_rows = np.arange(c["n"])
    for ep in range(c["epochs"]):
        enc.train(); dec.train()
        for rows in batches[ep * per_epoch:(ep + 1) * per_epoch]:
            t = S.BatchTiles(csr, rows)
            S.train_VAE(t, t.lib.reshape(-1, 1), enc, dec, opt, kl_weight=c["kl_weight"], likelihood_type="nb")
        enc.eval(); dec.eval()
        t = S.BatchTiles(csr, all_rows)
        tot, rec, kl = S.eval_VAE(t, t.lib.reshape(-1, 1), enc, dec, kl_weight=c["kl_weight"], likelihood_type="nb")
        got = np.array([float(tot), float(rec), float(kl)])
        assert np.max(np.abs(got - d["eval"][ep]) / np.abs(d["eval"][ep])) < 5e-4, (ep, got, d["eval"][ep])
    for k, v in enc.state_dict().items():
        assert float((v.cpu() - torch.tensor(d["enc_final/" + k])).abs().max()) < 2e-3, k


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_vae_trainer_replays_reference_trajectory(precision):
    """(precision 'bf16': same replay with bf16 MFMA operands; losses within 2e-2, weights within 2e-2 of their scale.)
    scVAE_object (the package's epoch loop around train_VAE / eval_VAE, shaped like model.py:727-799) replaying the
    reference's recorded VAE run: same initial weights, batch order and noise stream -> per-epoch validation
    (total, reconstruction, KL) within 5e-4 relative and final encoder weights within 2e-3."""
    d = U.load("traj_vae_nb")
    c = U.cfg(d, "cfg")
    rna = synth.make_rna(c["n"], c["g"], seed=c["seed"])
    obj = S.scVAE_object(distribution="nb", SEED=0, hidden_dim=c["hid"], latent_dim=c["L"], dropout_rate=0, batch_size=c["batch"],
                         L2_lambda=1e-9, max_epochs=c["epochs"], kl_weight=c["kl_weight"], device=DEV, sample_factor=True,
                         infer_library_size=False, verbose=False, weight_decay_groups=False,
                         use_mixed_precision=(precision == "bf16"))
    ltol, wtol = (5e-4, 2e-3) if precision == "fp32" else (2e-2, None)
    eps_all = torch.tensor(d["eps"]).to(DEV)
    pos = [0]

    def eps_source(shape, device):
        n = int(np.prod(shape))
        e = eps_all[pos[0]:pos[0] + n].reshape(shape)
        pos[0] += n
        return e
    obj.eps_source = eps_source
    init = ({k: torch.tensor(v) for k, v in U.sub(d, "enc_init", tensor=False).items()},
            {k: torch.tensor(v) for k, v in U.sub(d, "dec_init", tensor=False).items()})
    # the package builds the same initial weights from SEED alone (construction order of model.py:620-694)
    enc0, dec0 = obj.build(c["g"])
    for k, v in enc0.state_dict().items():
        assert torch.equal(v, init[0][k]), k
    for k, v in dec0.state_dict().items():
        assert torch.equal(v, init[1][k]), k
    obj.train(rna, rna, batch_replay=U.split_batches(d["rows"], d["batch_sizes"]))
    assert pos[0] == eps_all.numel()
    got = np.stack([obj.history["val_loss"], obj.history["val_recon"], obj.history["val_kl"]], 1)
    assert got.shape == d["eval"].shape
    assert np.max(np.abs(got - d["eval"]) / np.abs(d["eval"])) < ltol, (got, d["eval"])
    for k, v in obj.last_encoder.state_dict().items():
        ref = torch.tensor(d["enc_final/" + k])
        tol = wtol if wtol is not None else 2e-2 * max(float(ref.abs().max()), 1e-3)
        assert float((v.cpu() - ref).abs().max()) < tol, k
    assert len(obj.history["train_loss"]) == len(d["batch_sizes"])
    obj.encoder.eps_source = None                      # the recorded noise stream is used up
    emb = obj.embeddings(rna)
    assert emb.shape == (c["n"], c["L"]) and bool(torch.isfinite(emb).all())
    S.set_precision("fp32")


def test_run_from_anndata_like_object():
    """The reference's top-level workflow `scPair_object(scobj=adata, cov=[...]).run()` (model.py:517-526) on an
    AnnData-like object: data_loader_builder keeps the matrices sparse, the predictors, embeddings and the
    bidirectional mapping train on the GPU; every output has the reference's keys and shapes and finite values."""
    import scpair_amd as S
    from tests.util import make_mini_adata
    ad = make_mini_adata(seed=5, n=400, g=40, p=60)
    obj = S.scPair_object(scobj=ad, cov=["labels", "depth"], modalities={"Gene Expression": "zinb", "Peaks": "ber"},
                          hidden_layer=[32, 8], batch_size=64, max_epochs=3, early_stopping_patience=2,
                          device=torch.device("cuda:0"), save_model=False)
    enc, dec, maps, emb, emb_mapped = obj.run()
    n_train = int((ad.obs["scPair_split"] == "train").sum())
    assert set(enc.keys()) == {"Gene Expression to Peaks", "Peaks to Gene Expression"} == set(dec.keys())
    assert emb["Gene Expression_train"].shape == (n_train, 8) and emb["Peaks_train"].shape == (n_train, 8)
    _, emb_df = obj.reference_embeddings()
    assert list(emb_df["Gene Expression_train"].index) == list(obj.train_metadata.index)
    for v in list(emb.values()) + list(emb_mapped.values()):
        assert np.isfinite(np.asarray(v)).all()
    pred = obj.predict()
    assert pred["Peaks_test"].shape[1] == 60 and np.isfinite(pred["Peaks_test"]).all()


def test_run_with_input_transform():
    """Opt-in log1p / library-size normalisation of the RNA encoder input (no counterpart upstream, default off):
    the workflow runs end to end, only the RNA -> ATAC predictor sees different inputs, and the count labels of the
    ATAC -> RNA predictor stay raw (its first-epoch loss is unchanged)."""
    import scpair_amd as S
    from tests.util import make_mini_adata
    hist = {}
    for tag, tf in (("raw", None), ("norm", {"Gene Expression": ("log1p_norm", 1e3)})):
        ad = make_mini_adata(seed=5, n=400, g=40, p=60)
        obj = S.scPair_object(scobj=ad, cov=None, modalities={"Gene Expression": "nb", "Peaks": "ber"},
                              hidden_layer=[32, 8], batch_size=64, max_epochs=2, early_stopping_patience=2,
                              dropout_rate=0.0, device=torch.device("cuda:0"), save_model=False, input_transform=tf)
        obj.run()
        hist[tag] = obj.history
        for v in obj.embeddings.values():
            assert np.isfinite(np.asarray(v.cpu() if hasattr(v, "cpu") else v)).all()
    a, b = hist["raw"], hist["norm"]
    assert abs(a["Peaks to Gene Expression"]["train_loss"][0] - b["Peaks to Gene Expression"]["train_loss"][0]) < 1e-6
    assert abs(a["Gene Expression to Peaks"]["train_loss"][3] - b["Gene Expression to Peaks"]["train_loss"][3]) > 1e-6
    with pytest.raises(ValueError):
        S.scPair_object(scobj=None, modalities={"Gene Expression": "nb", "Peaks": "ber"}, save_model=False,
                        device=torch.device("cuda:0"), input_transform={"Gene Expression": "sqrt"})


@pytest.mark.parametrize("golden", ["augment", "augment_cov"])
def test_augment_matches_reference(golden):
    """augment() (model.py:925-1036) from the reference's trained RNA encoder on extra unimodal cells, against the
    four embedding tables and the fine-tuned encoder the reference produced (tests/golden/augment.npz,
    tools/make_golden_augment.py).  Dropout 0 and BatchNorm on make the fine-tuning deterministic: the train-mode
    batch statistics are the learning signal.  augment_cov: the paired data carry a one-hot covariate, the unimodal
    cells none, so their covariate block is all zeros (model.py:936-940)."""
    import json
    import os
    import scipy.sparse as sp
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", golden + ".npz"))
    c = json.loads(str(d["cfg"]))
    with_cov = "cov" in d.files
    data = synth.make_paired(c["n"], c["g"], c["p"], seed=c["seed"], p_atac=0.1)
    dicts = []
    for split in ("train", "val", "test"):
        idx = data[split]
        dicts.append({f"{nm}_{split}_{k}": (mat[idx] if k == "mtx" else (torch.tensor(d["cov"][idx]) if with_cov else None))
                      for nm, mat in (("Gene Expression", data["rna"]), ("Peaks", data["atac"])) for k in ("mtx", "cov")})
    obj = S.scPair_object(None, cov=(["labels"] if with_cov else None), modalities={"Gene Expression": "nb", "Peaks": "ber"},
                          SEED=0, hidden_layer=c["hidden"], dropout_rate=0.0, batchnorm=True, layernorm=True,
                          batch_size=c["batch"], max_epochs=c["aug_epochs"], device=DEV, save_model=False, verbose=False)
    obj.set_data(*dicts)
    enc, _, _ = obj._build_predictor("Gene Expression", "Peaks", c["g"], c["p"], 4 if with_cov else 0)
    sd = {k[len("enc_trained/"):]: torch.tensor(d[k]) for k in d.files if k.startswith("enc_trained/")}
    enc.load_state_dict(sd)
    obj.encoder_dict = {"Gene Expression to Peaks": enc.eval()}
    uni = sp.csr_matrix((d["uni/data"], d["uni/indices"], d["uni/indptr"]), shape=(c["n_uni"], c["g"]))
    aug_d, aug_emb = obj.augment(type("Uni", (), {"X": uni})(), {"Gene Expression": "nb"})
    assert list(aug_d.keys()) == [str(d["aug_key"])]
    assert set(aug_emb.keys()) == {k[len("emb/"):] for k in d.files if k.startswith("emb/")}
    for k, v in aug_emb.items():
        ref = d["emb/" + k]
        assert v.shape == ref.shape
        assert float(np.abs(v - ref).max()) < 2e-3 * max(1.0, float(np.abs(ref).max())), k
    for k, v in list(aug_d.values())[0].state_dict().items():
        ref = torch.tensor(d["enc_augmented/" + k])
        if ref.dtype.is_floating_point:
            assert float((v.cpu() - ref).abs().max()) < 2e-3 * max(1.0, float(ref.abs().max())), k
    # the fine-tuning moved the encoder away from the one it started from
    moved = max(float((v.cpu() - sd[k]).abs().max()) for k, v in list(aug_d.values())[0].state_dict().items()
                if sd[k].dtype.is_floating_point)
    assert moved > 1e-3
