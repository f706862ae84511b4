"""A training step captured in a HIP graph (scpair_amd.graph.CapturedStep) against the same step issued eagerly:
with dropout off the replayed steps are bitwise the eager ones (losses, weights, AdamW moments and step counters)
although the minibatch, the AdamW bias corrections and the dropout seeds change from replay to replay; with dropout
on, every replay draws a new mask."""
import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _build(S, G, P, hidden, dropout, seed=0, batchnorm=False):
    S.set_seed(seed)
    enc = S.Input_Module(input_dim=G, input_batch_num=0, hidden_layer=hidden, activation=nn.LeakyReLU(), layernorm=True,
                         batchnorm=batchnorm, dropout_rate=dropout, add_linear_layer=True, infer_library_size=True)
    S.set_seed(seed)
    dec = S.Output_Module_Ber(output_dim=P, output_batch_num=0, hidden_layer=hidden, infer_library_size=True,
                              sample_factor=False, feature_factor=False)
    enc.apply(S.init_weights).to(DEV).train()
    dec.apply(S.init_weights).to(DEV).train()
    de, ne, _, _ = S.add_weight_decay(enc, output_layer=["output"])
    dd, nd, _, _ = S.add_weight_decay(dec, output_layer=["output"])
    opt = S.FusedAdamW([{"params": de, "weight_decay": 1e-2, "lr": 1e-3}, {"params": dd + ne + nd, "weight_decay": 0, "lr": 1e-3}])
    return enc, dec, opt


def _state(enc, dec, opt):
    out = {}
    for nm, m in (("enc", enc), ("dec", dec)):
        for k, v in m.state_dict().items():
            out[f"{nm}.{k}"] = v.detach().clone()
        for k, p in m.named_parameters():
            if len(opt.state[p]):
                out[f"{nm}.{k}.m"] = opt.state[p]["exp_avg"].clone()
                out[f"{nm}.{k}.v"] = opt.state[p]["exp_avg_sq"].clone()
                out[f"{nm}.{k}.step"] = torch.tensor(opt.state[p]["step"])
    return out


@pytest.mark.parametrize("precision,fuse,batchnorm", [("fp32", False, False), ("bf16", True, False), ("bf16", False, True)])
def test_captured_step_is_the_eager_step(precision, fuse, batchnorm):
    import scpair_amd as S
    from scpair_amd import synth
    from scpair_amd.graph import CapturedStep
    from scpair_amd.utils import _predictor_loss
    S.set_precision(precision)
    try:
        n, G, P, B = 700, 96, 260, 130
        data = synth.make_paired(n, G, P, seed=3, p_atac=0.08)
        rna, atac = S.DeviceCSR.from_scipy(data["rna"], DEV), S.DeviceCSR.from_scipy(data["atac"], DEV, binary=True)
        rng = np.random.default_rng(0)
        batches = [torch.from_numpy(rng.permutation(n)[:B]).to(DEV) for _ in range(7)]
        res = {}
        for mode in ("eager", "graph"):
            enc, dec, opt = _build(S, G, P, [32, 8], 0.0, batchnorm=batchnorm)
            if fuse:
                opt.fuse_first_layer(enc.hidden_rep[0].weight, enc.hidden_lib[0][0].weight)
            cap_x, cap_y = rna.max_batch_nnz(B), atac.max_batch_nnz(B)

            def fn(rows, fixed):
                tx = S.BatchTiles(rna, rows, want_lib=False, capacity=cap_x if fixed else None)
                ty = S.BatchTiles(atac, rows, want_lib=False, capacity=cap_y if fixed else None)
                loss = _predictor_loss(tx, ty, None, enc, dec, "ber", False, None, None)
                opt.zero_grad()
                loss.backward()
                opt.step()
                return loss.detach()

            losses = []
            for rows in batches[:2]:                       # eager warm-up in both modes (lazy state)
                losses.append(float(fn(rows, False)))
            if mode == "graph":
                cs = CapturedStep(lambda rows: fn(rows, True), [opt], B, DEV).capture(batches[2])
                for rows in batches[2:6]:
                    losses.append(float(cs(rows)))
                losses.append(float(fn(batches[6], False)))          # an eager step after replays (ragged last batch)
                assert cs.replays == 4
            else:
                for rows in batches[2:]:
                    losses.append(float(fn(rows, False)))
            res[mode] = (losses, _state(enc, dec, opt))
        assert res["eager"][0] == res["graph"][0], (res["eager"][0], res["graph"][0])
        for k, v in res["eager"][1].items():
            assert torch.equal(v, res["graph"][1][k]), k
        assert int(res["graph"][1]["enc.hidden_rep.0.weight.step"]) == 7
    finally:
        S.set_precision("fp32")


def test_captured_step_draws_new_dropout_masks():
    import scpair_amd as S
    from scpair_amd import synth
    from scpair_amd.graph import CapturedStep
    from scpair_amd.utils import _predictor_loss
    S.set_precision("fp32")
    n, G, P, B = 300, 64, 128, 130
    data = synth.make_paired(n, G, P, seed=4, p_atac=0.08)
    rna, atac = S.DeviceCSR.from_scipy(data["rna"], DEV), S.DeviceCSR.from_scipy(data["atac"], DEV, binary=True)
    enc, dec, opt = _build(S, G, P, [32, 8], 0.5)
    for g in opt.param_groups:
        g["lr"] = 0.0                                       # weights frozen: only the mask changes the loss
    rows = torch.arange(B, device=DEV)

    def fn(rows_, fixed=True):
        tx = S.BatchTiles(rna, rows_, want_lib=False, capacity=rna.max_batch_nnz(B) if fixed else None)
        ty = S.BatchTiles(atac, rows_, want_lib=False, capacity=atac.max_batch_nnz(B) if fixed else None)
        loss = _predictor_loss(tx, ty, None, enc, dec, "ber", False, None, None)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss.detach()

    fn(rows, False)
    cs = CapturedStep(fn, [opt], B, DEV).capture(rows)
    vals = [float(cs(rows)) for _ in range(5)]
    assert len(set(vals)) == 5, vals                         # same cells, same weights: a new mask every replay
    assert max(vals) - min(vals) < 0.5 * abs(np.mean(vals))
