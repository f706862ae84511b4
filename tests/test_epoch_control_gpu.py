"""Epoch control of the trainer loops (reference model.py:592-602 scheduler, 719-799 early stopping + best-epoch deep copy,
803-807 / 916-919 saved files) against two runs recorded from the reference itself (tools/make_goldens.py `epoch`):
  traj_earlystop : patience 2, lr 0.02 -- the toy overfits, so early stopping FIRES in both predictors (after 4 and 6
                   epochs) and in one mapping net (21 of 40 epochs); save_model=True: the file names and sizes are in cfg
  traj_steplr    : weight_decay='StepLR' (step 25, gamma 0.1; the only scheduler that works upstream) over 29 epochs, i.e.
                   across its first decay
Tolerances (fp32 mode): number of epochs run and best epoch equal; per-epoch validation loss rel 1e-3 (lr is 5-20 x the
default, which amplifies summation-order noise over the epochs; measured values are printed); best-epoch weights max-abs
5e-3."""
import copy
import os

import numpy as np
import pytest
import torch
from torch import nn

import scpair_amd as S
from tests import util as U
from tests.test_trajectory_gpu import _rel, _run

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _check_history(name, d, c, obj, map_too=True):
    names = list(c["modalities"].keys())
    for di in range(2):
        key = names[di] + " to " + names[1 - di]
        h, g = obj.history[key], d[f"dir{di}/val_loss"]
        assert len(h["val_loss"]) == len(g), (key, len(h["val_loss"]), len(g))           # stopped after the same epoch
        assert int(np.argmin(h["val_loss"])) == int(np.argmin(g)), key                    # same best epoch
        print(f"[measured] {name} {key}: epochs {len(g)}, best {int(np.argmin(g))}, val rel {_rel(h['val_loss'], g):.2e}, "
              f"train rel {_rel(h['train_loss'], d[f'dir{di}/train_loss']):.2e}", flush=True)
        assert _rel(h["val_loss"], g) < 1e-3 and _rel(h["train_loss"], d[f"dir{di}/train_loss"]) < 1e-3, key
        # what the loop returns is the deep copy taken at the best epoch, not the last weights
        for mod, tag in ((obj.encoder_dict[key], "enc_final"), (obj.decoder_dict[key], "dec_final")):
            for k, v in mod.state_dict().items():
                ref = torch.tensor(d[f"dir{di}/{tag}/{k}"])
                if ref.dtype.is_floating_point:
                    assert float((v.cpu() - ref).abs().max()) < 5e-3 * max(1.0, float(ref.abs().max())), (key, k)
    if not map_too:
        return
    obj.reference_embeddings()
    obj.train_bidirectional_mapping()
    for mi in range(2):
        key = names[mi] + "_to_" + names[1 - mi]
        h, g = obj.history[key], d[f"map{mi}/val_loss"]
        print(f"[measured] {name} {key}: epochs {len(h['val_loss'])} (reference {len(g)})", flush=True)
        assert len(h["val_loss"]) == len(g), (key, len(h["val_loss"]), len(g))
        n = min(len(g), 10)
        assert _rel(h["val_loss"][:n], g[:n]) < 2e-2, key


def test_early_stopping_fires_and_files_are_saved(tmp_path):
    d = U.load("traj_earlystop")
    c = U.cfg(d, "cfg")
    assert len(d["dir0/val_loss"]) < c["epochs"] and len(d["dir1/val_loss"]) < c["epochs"]        # it did fire upstream
    _, _, obj = _run("traj_earlystop", "fp32", save_model=True, save_path=str(tmp_path))
    _check_history("traj_earlystop", d, c, obj)
    # the files the reference leaves behind (model.py:803-807, 916-919): same names; whole-module pickles whose size is
    # the parameter bytes plus pickle overhead (the reference's own sizes are recorded next to the names)
    want = c["saved_files"]
    got = {f: os.path.getsize(tmp_path / f) for f in sorted(os.listdir(tmp_path))}
    assert set(got) == set(want), (sorted(got), sorted(want))
    # sizes: the parameter bytes plus pickle overhead.  (The reference's own files are larger than their parameters --
    # 34 KB for a 4.6 KB Bernoulli decoder: its pickled modules carry more than the state_dict -- so its recorded sizes
    # are an upper bound here, not a target.)
    mods = {"encoder_": obj.encoder_dict, "decoder_": obj.decoder_dict}
    for f, sz in got.items():
        assert sz <= want[f] + 8192, (f, sz, want[f])
        for pre, dct in mods.items():
            if f.startswith(pre):
                m = dct[f[len(pre):-3].replace("_to_", " to ")]
                nb = sum(v.numel() * v.element_size() for v in m.state_dict().values())
                assert nb <= sz < 1.3 * nb + 16384, (f, sz, nb)
    key = "Gene Expression to Peaks"
    back = torch.load(tmp_path / "encoder_Gene Expression_to_Peaks.pt", weights_only=False)       # a file this test wrote
    for k, v in obj.encoder_dict[key].state_dict().items():
        assert torch.equal(back.state_dict()[k].cpu(), v.cpu()), k


def test_steplr_scheduler_matches_reference():
    d = U.load("traj_steplr")
    c = U.cfg(d, "cfg")
    assert c["scheduler"] == "StepLR" and len(d["dir0/val_loss"]) == 29
    _, _, obj = _run("traj_steplr", "fp32")
    _check_history("traj_steplr", d, c, obj, map_too=False)


def _plain_encoder(G, hidden, infer_lib):
    """A plain torch.nn tree with the reference's submodule layout (model.py:60-92, utils.py:54-64), i.e. what a
    reference-side `load_state_dict` targets."""
    def fcnn():
        return nn.Sequential(nn.Linear(G, hidden[0]), nn.LayerNorm(hidden[0]), nn.LeakyReLU(), nn.Dropout(0.0),
                             nn.Linear(hidden[0], hidden[1]), nn.LayerNorm(hidden[1]), nn.LeakyReLU(), nn.Dropout(0.0))
    m = nn.Module()
    m.hidden_rep = fcnn()
    m.extra_linear_rep = nn.Linear(hidden[1], hidden[1])
    if infer_lib:
        m.hidden_lib = nn.Sequential(fcnn(), nn.Linear(hidden[1], 1))
    return m


def test_checkpoint_of_packed_encoder_round_trips(tmp_path):
    """Precision 'bf16' packs the two first-layer weights of an encoder with a library branch into one [2 H, F] block
    (their Parameters become views of it).  deepcopy (the best-epoch checkpoint) and torch.save must neither alias the
    live weights nor write the block twice, and the state_dict must load -- with weights_only=True -- into a plain
    nn.Module tree with the reference's parameter names and reproduce the encoder's outputs."""
    from scpair_amd import synth
    G, P, hidden, B = 600, 900, [132, 16], 96
    data = synth.make_paired(B, G, P, seed=3, p_atac=0.05)
    S.set_precision("bf16")
    try:
        S.set_seed(0)
        enc = S.Input_Module(input_dim=G, input_batch_num=0, hidden_layer=hidden, activation=nn.LeakyReLU(), layernorm=True,
                             batchnorm=False, dropout_rate=0.0, add_linear_layer=True, infer_library_size=True)
        enc.apply(S.init_weights).to(DEV).eval()
        t = S.BatchTiles(S.DeviceCSR.from_scipy(data["rna"], DEV), np.arange(B))
        rep, lat = enc(t, None)                                   # bf16 mode, two branches: packs the first layers
        w_rep, w_lib = enc.hidden_rep[0].weight, enc.hidden_lib[0][0].weight
        assert w_rep.data.untyped_storage().data_ptr() == w_lib.data.untyped_storage().data_ptr(), "first layers are not packed"
        ck = copy.deepcopy(enc)
        assert ck.hidden_rep[0].weight.data_ptr() != w_rep.data_ptr()
        w0 = w_rep.detach().clone()
        with torch.no_grad():
            w_rep.add_(1.0)                                       # the live weights move on; the checkpoint must not
        assert torch.equal(ck.hidden_rep[0].weight.detach(), w0)
        assert torch.equal(ck.hidden_lib[0][0].weight.detach(), w_lib.detach())
        with torch.no_grad():
            w_rep.copy_(w0)
        n_bytes = sum(v.numel() * v.element_size() for v in enc.state_dict().values())
        torch.save(enc, tmp_path / "enc_module.pt")
        torch.save(enc.state_dict(), tmp_path / "enc_sd.pt")
        for f in ("enc_module.pt", "enc_sd.pt"):
            sz = os.path.getsize(tmp_path / f)
            assert n_bytes <= sz < 1.15 * n_bytes + 65536, (f, sz, n_bytes)          # not the packed block twice
        sd = torch.load(tmp_path / "enc_sd.pt", weights_only=True)
        plain = _plain_encoder(G, hidden, True)
        missing, unexpected = plain.load_state_dict(sd, strict=True)
        assert not missing and not unexpected
        x = torch.tensor(np.asarray(data["rna"].todense(), dtype=np.float32))
        with torch.no_grad():
            pr = plain.extra_linear_rep(plain.hidden_rep(x))
            pl = plain.hidden_lib(x)
        # bf16 MFMA operands in the first layer vs fp32 torch: 2e-2 of the output scale
        assert float((rep.cpu() - pr).abs().max()) < 2e-2 * max(1.0, float(pr.abs().max()))
        assert float((lat.cpu() - pl).abs().max()) < 2e-2 * max(1.0, float(pl.abs().max()))
        # and back: the reference-side state_dict loads into the HIP-backed module
        enc2 = S.Input_Module(input_dim=G, input_batch_num=0, hidden_layer=hidden, activation=nn.LeakyReLU(), layernorm=True,
                              batchnorm=False, dropout_rate=0.0, add_linear_layer=True, infer_library_size=True)
        enc2.load_state_dict(plain.state_dict())
        enc2.to(DEV).eval()
        rep2, lat2 = enc2(t, None)
        assert torch.equal(rep2, rep) and torch.equal(lat2, lat)
    finally:
        S.set_precision("fp32")
