#!/usr/bin/env python3
"""Golden fixture of the reference's augment() (model.py:925-1036): a small nb+ber object is trained with the
reference's own loop for a few epochs, then its RNA encoder is fine-tuned on extra unimodal cells (dropout 0 and batchnorm on: the train-mode batch statistics are the only learning signal, and it is deterministic).  Recorded: the data,
the trained encoder the fine-tuning starts from, the fine-tuned encoder and the four embedding tables augment() returns.
Writes tests/golden/augment.npz; with `cov` as argument the paired data carry a one-hot covariate and the unimodal cells
none (the one covariate combination that runs upstream, model.py:936-940: their covariate block is all zeros), written to
tests/golden/augment_cov.npz."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
WITH_COV = "cov" in sys.argv[1:]
sys.argv = [sys.argv[0], "--import-only"]
import make_goldens as MG  # noqa: E402  (registers the import stubs and imports the reference)
import torch  # noqa: E402
from scpair_amd import synth  # noqa: E402

ref = MG.ref
cfg = dict(n=384, g=48, p=96, hidden=[16, 8], batch=64, epochs=3, aug_epochs=4, n_uni=200, seed=3)
data = synth.make_paired(cfg["n"], cfg["g"], cfg["p"], seed=cfg["seed"], p_atac=0.1)
if WITH_COV:
    data["cov"] = np.eye(4, dtype=np.float32)[np.random.default_rng(7).integers(0, 4, size=cfg["n"])]
tr, va, te = MG.dense_dicts(data, WITH_COV)
modalities = {"Gene Expression": "nb", "Peaks": "ber"}
obj = ref.scPair_object(scobj=None, cov=(["labels"] if WITH_COV else None), modalities=modalities, SEED=0, hidden_layer=cfg["hidden"],
                        dropout_rate=0.0, batchnorm=True, layernorm=True, batch_size=cfg["batch"],
                        max_epochs=cfg["epochs"], device=torch.device("cpu"), save_model=False)
enc_d, dec_d = obj.train_predicting_networks(tr, va, te)
obj.encoder_dict, obj.decoder_dict = enc_d, dec_d
obj.data_loader_dict_train, obj.data_loader_dict_val, obj.data_loader_dict_test = tr, va, te
uni = synth.make_paired(cfg["n_uni"], cfg["g"], cfg["p"], seed=cfg["seed"] + 1, p_atac=0.1)["rna"]


class Uni:
    X = uni
    shape = uni.shape


if WITH_COV:
    import pandas as pd
    C = data["cov"].shape[1]
    obj.cov_dummy = pd.DataFrame(np.zeros((cfg["n"], C)), columns=[f"l{i}" for i in range(C)])   # columns / width only
    Uni.obs = pd.DataFrame(index=[f"u{i}" for i in range(cfg["n_uni"])])


obj.max_epochs = cfg["aug_epochs"]
aug_d, aug_emb = obj.augment(Uni(), {"Gene Expression": "nb"})
out = {"cfg": np.array(json.dumps(cfg)), "__torch_version__": np.array(torch.__version__),
       "uni/data": uni.data.astype(np.float32), "uni/indices": uni.indices.astype(np.int32),
       "uni/indptr": uni.indptr.astype(np.int64)}
for k, v in enc_d["Gene Expression to Peaks"].state_dict().items():
    out[f"enc_trained/{k}"] = v.numpy()
for k, v in list(aug_d.values())[0].state_dict().items():
    out[f"enc_augmented/{k}"] = v.numpy()
out["aug_key"] = np.array(list(aug_d.keys())[0])
if WITH_COV:
    out["cov"] = data["cov"]
for k, v in aug_emb.items():
    out[f"emb/{k}"] = np.asarray(v)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "augment_cov.npz" if WITH_COV else "augment.npz"), **out)
print("wrote", "augment_cov.npz" if WITH_COV else "augment.npz", ":", {k: np.asarray(v).shape for k, v in aug_emb.items()})
