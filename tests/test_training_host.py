"""Control flow of the reference's training driver (train / generate_self_train_data / save_ckp / load_ckp)
with a stand-in model on CPU (the numeric step itself is covered by the -m gpu tests)."""
import os
import sys
import random

import pandas as pd
import pytest
import torch
import torch.nn as nn

from carel_vae_amd import training as T
from carel_vae_amd.drl_classifier import make_opt


class TinyModel(nn.Module):
    def __init__(self):
        super().__init__()
        self.w = nn.Parameter(torch.zeros(3))
        self.calls = []

    def forward(self, ids, att, tt, emo, cau, labels, bow, iteration):
        self.calls.append(int(iteration))
        return (self.w.sum() - 1.0) ** 2 + labels.mean() * 0

    def get_pair_preds(self, ids, att, tt):
        return [[float(i % 2)] for i in range(ids.shape[0])]


def batches(n, bs):
    for s in range(0, n, bs):
        m = min(bs, n - s)
        yield {"input_ids": torch.zeros((m, 8), dtype=torch.long), "attention_masks": torch.ones((m, 8), dtype=torch.long),
               "token_type_ids": torch.zeros((m, 8), dtype=torch.long), "labels": torch.tensor([[float(i % 2)] for i in range(s, s + m)]),
               "emo_labels": torch.zeros((m, 1), dtype=torch.long), "cau_labels": torch.zeros((m, 1)), "bow_reps": torch.zeros((m, 5))}


def test_prf1_matches_definition():
    p, r, f = T._prf1([[1], [0], [1], [1]], [[1.0], [1.0], [0.0], [1.0]])
    assert (p, r) == (2 / 3, 2 / 3) and abs(f - 2 / 3) < 1e-12
    assert T._prf1([[0]], [[0.0]]) == (0.0, 0.0, 0.0)


def test_train_loop_checkpoint_and_iteration_counter(tmp_path):
    opt = make_opt(epochs=2, self_epochs=1, best_model_path=str(tmp_path / "ckpt"), model_id="t1")
    model = TinyModel()
    optim = torch.optim.SGD(model.parameters(), lr=0.1)
    logs = []
    best = T.train(list(batches(10, 4)), list(batches(6, 6)), model, [optim], "cpu", num_unpred_pairs=2, opt=opt, log=logs.append)
    assert model.calls == [0, 1, 2, 0, 1, 2]           # `iteration` restarts every epoch (ref :823, quirk Q4)
    assert os.path.exists(tmp_path / "ckpt" / "t1.pt") and best is model
    assert float(model.w.sum()) != 0.0
    out = T.train(list(batches(4, 4)), list(batches(6, 6)), model, [optim], "cpu", 2, self_metrics=[0.0, 0.0, 0.0], self_train=True,
                  opt=opt, log=logs.append)
    assert len(out) == 4 and out[0] is model


def test_save_load_roundtrip(tmp_path):
    m = TinyModel()
    with torch.no_grad():
        m.w.copy_(torch.tensor([1.0, 2.0, 3.0]))
    T.save_ckp(m.state_dict(), str(tmp_path), "abc")
    m2 = TinyModel()
    T.load_ckp(str(tmp_path / "abc.pt"), m2)
    assert torch.equal(m2.w, m.w)


def test_generate_self_train_data_random_strategy():
    random.seed(0)
    df = pd.DataFrame({"pair": ["p%d" % i for i in range(7)], "label": [0] * 7, "emotion": [i % 6 for i in range(7)]})
    loader = [{"input_ids": torch.zeros((7, 8), dtype=torch.long), "attention_masks": torch.ones((7, 8), dtype=torch.long),
               "token_type_ids": torch.zeros((7, 8), dtype=torch.long)}]
    out = T.generate_self_train_data([3, 1, 3], df, loader, TinyModel(), "random", device="cpu")
    assert list(out.columns) == ["pair", "label", "emotion"]
    assert list(out["label"]) == [1, 0, 1, 0]          # the single-pair document yields nothing (ref :782)
    assert out["pair"][0] == "p1" and out["pair"][2] == "p5"   # highest predicted score per document


class TinyViModel(TinyModel):
    """Stand-in with the VI ablation's surface: forward -> (e, c, aprx_loss, vae_loss), get_ec_upper_loss."""

    def __init__(self):
        super().__init__()
        self.a = nn.Parameter(torch.ones(2))
        self.betas = []

    def forward(self, ids, att, tt, emo, cau, labels, bow, iteration):
        loss = super().forward(ids, att, tt, emo, cau, labels, bow, iteration)
        e = self.w[:2] * 1.0
        return e, e.detach() + 1.0, (self.a * e.detach()).pow(2).sum(), loss

    def get_ec_upper_loss(self, e, c):
        return (e * c).sum()


def test_train_loop_vi_two_phase(tmp_path):
    opt = make_opt(epochs=3, best_model_path=str(tmp_path / "ckpt"), model_id="vi")
    model = TinyViModel()
    aprx_opt = torch.optim.SGD([model.a], lr=0.1)
    main_opt = torch.optim.SGD([model.w], lr=0.1)
    a0 = model.a.detach().clone()
    T.train(list(batches(8, 4)), list(batches(6, 6)), model, [aprx_opt, main_opt], "cpu", num_unpred_pairs=0, opt=opt, log=lambda *_: None)
    assert model.calls == [0, 1] * 3
    assert not torch.equal(model.a.detach(), a0)            # the approximation optimiser stepped
    assert float(model.w.abs().sum()) > 0.0                 # and so did the main one


class TinyEnModel(nn.Module):
    """Stand-in with the surface of drl_classifier_en.py: forward -> seven losses, five discriminator groups + the rest,
    get_pair_preds -> raw logits tensor."""

    def __init__(self):
        super().__init__()
        self.d = nn.ParameterList([nn.Parameter(torch.ones(2)) for _ in range(5)])
        self.w = nn.Parameter(torch.zeros(3))
        self.emo_dtype = None

    def forward(self, ids, att, tt, emo, cau, labels, bow, iteration):
        self.emo_dtype = emo.dtype
        z = self.w.detach()                                  # discriminators see detached features (:425-515)
        dl = [((d * (z[:2] + 1.0)).sum() - 1.0) ** 2 for d in self.d]
        vae = (self.w.sum() - 1.0) ** 2 + 0.1 * sum((d * (z[:2] + 1.0)).sum() for d in self.d)      # entropy-like term on the discriminators
        return dl[0], dl[0] * 0.5, dl[1], dl[3], dl[2], dl[4], vae

    def get_pair_preds(self, ids, att, tt):
        return torch.tensor([[3.0 if i % 2 else -3.0] for i in range(ids.shape[0])])


def test_train_loop_english_six_optimisers(tmp_path):
    opt = make_opt(epochs=2, best_model_path=str(tmp_path / "ckpt"), model_id="en")
    model = TinyEnModel()
    opts = [torch.optim.RMSprop([d], lr=0.01) for d in model.d] + [torch.optim.Adam([model.w], lr=0.01)]
    d0 = [d.detach().clone() for d in model.d]
    logs = []
    T.train(list(batches(8, 4)), list(batches(6, 6)), model, opts, "cpu", num_unpred_pairs=0, opt=opt, log=logs.append)
    assert model.emo_dtype == torch.float32                  # the English dataset's emotion label is a float (:132, :909)
    assert all(not torch.equal(d.detach(), d0[i]) for i, d in enumerate(model.d)) and float(model.w.abs().sum()) > 0
    f1_lines = [l for l in logs if "f1 socre" in str(l)]
    assert len(f1_lines) == 2 and "1.0000" in str(f1_lines[0])      # sigmoid(logits).round() equals the alternating labels
    # each discriminator's .grad after the step = its own loss's gradient + the vae loss's share (zero_grad order of :919-939)
    g = model.d[1].grad.clone()
    z = model.w.detach()[:2] + 1.0
    assert g.abs().sum() > 0 and torch.isfinite(g).all() and z.numel() == 2


def test_bench_launches_its_own_ranks_and_checks_the_count(monkeypatch):
    """`python bench.py --gpus N` without a launcher must start N rank processes as a child (before any GPU call),
    relay rank 0's JSON line and fail unless exactly N ranks reported (it used to run one GPU and say so)."""
    import importlib
    import json
    import subprocess
    import types
    import pytest
    bench = importlib.import_module("bench")
    seen = {}

    def fake_run(cmd, stdout=None, text=None):
        seen["cmd"] = cmd
        return types.SimpleNamespace(returncode=0, stdout="RCCL banner\n" + json.dumps({"metric": "clause-pairs/sec (training step)", "n_gpus": seen["n"]}) + "\n")
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    a = types.SimpleNamespace(gpus=4)
    seen["n"] = 4
    bench.launch_ranks(a)
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    seen["n"] = 1                                    # a run that silently used one GPU is an error, not a result
    with pytest.raises(SystemExit):
        bench.launch_ranks(a)


def test_mpnet_host_logic_bucket_map_and_state_dict_names():
    """CPU: the relative-position bucket map the model hands to the kernels is the oracle's (= transformers') function, entry by entry,
    and the MPNet sentence model speaks MPNetModel key names (round trip through load_state_dict / state_dict, no token types)."""
    import torch
    from carel_vae_amd import drl_classifier as M
    from carel_vae_amd import sentence_transformer as S
    from oracle import carel_oracle as O
    cfg = M.encoder_config("mpnet", vocab_size=64, layers=1)
    model = S.SentenceTransformer(cfg, seed=0)
    assert model.mpnet and model.normalize
    rel = model._m._rel_buffers()
    want = O.mpnet_relative_position_bucket(torch.arange(-127, 129))
    assert torch.equal(rel.bucket.cpu().long(), want)
    assert rel.dist.shape == (12, 256) and rel.ddist.shape == (12, 256)
    assert model._m._rel_buffers(batch=16).ddist.shape == (16 * 12, 256)      # the gradient by distance: one row per (sample, head)
    sd = model.state_dict()
    assert "encoder.relative_attention_bias.weight" in sd and sd["encoder.relative_attention_bias.weight"].shape == (32, 12)
    assert "encoder.layer.0.attention.attn.q.weight" in sd and "encoder.layer.0.attention.LayerNorm.bias" in sd
    assert not any("token_type" in k or ".self." in k for k in sd)
    sd2 = {k: torch.full_like(v, 0.25) for k, v in sd.items()}
    model.load_state_dict(sd2)
    back = model.state_dict()
    assert all(torch.equal(back[k], sd2[k]) for k in sd2)
    assert float(model._m._named[S.TT_KEY].detach().abs().max()) == 0.0          # the placeholder type row is not loadable / stays zero
    names = [k for k, _ in model.named_parameters()]
    assert len(names) == len(set(names)) and "pooler.dense.weight" not in names and "embeddings.token_type_embeddings.weight" not in names
    # the string constructor of the reference scripts selects the architecture (weights are never fetched)
    assert S.SentenceTransformer("sentence-transformers/all-mpnet-base-v2", seed=0).mpnet


def _golden_selftrain():
    import json
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "selftrain.json")))


@pytest.mark.parametrize("idx", range(9))
def test_generate_self_train_data_equals_the_reference_function(idx):
    """pinned to the reference's own generate_self_train_data (ref :734-799; fixtures by
    tests/golden/gen_golden_selftrain.py): three strategies x {rounded predictions as the reference's model returns them, fractional
    scores, all-zero scores}, same `random.seed` -- same rows in the same order, incl. the emotion column ("extreme" and "threshold"
    leave it None, :789-793) and the one-pair documents ("random" skips them, "extreme" emits the pair twice)."""
    c = _golden_selftrain()["self_train"][idx]
    n = len(c["scores"])
    df = pd.DataFrame({"pair": c["pairs"], "label": [0] * n, "emotion": c["emotions"]})
    z = torch.zeros((n, 4), dtype=torch.long)
    loader = [{"input_ids": z, "attention_masks": z + 1, "token_type_ids": z}]

    class Fixed:
        def eval(self): pass
        def get_pair_preds(self, ids, att, tt): return [[float(s)] for s in c["scores"]]
    random.seed(c["seed"])
    out = T.generate_self_train_data(c["sizes"], df, loader, Fixed(), c["strategy"], device="cpu")
    assert list(out.columns) == c["columns"]
    rows = [[r["pair"], int(r["label"]), None if r["emotion"] is None or pd.isna(r["emotion"]) else int(r["emotion"])] for _, r in out.iterrows()]
    assert rows == c["rows"], c["name"]


def test_prf1_equals_sklearn_fixtures():
    """ref :868-870 (sklearn precision / recall / f1, average="binary", zero_division -> 0.0): values generated with sklearn."""
    for m in _golden_selftrain()["metrics"]:
        p, r, f = T._prf1([[v] for v in m["labels"]], [[float(v)] for v in m["preds"]])
        assert abs(p - m["precision"]) < 1e-12 and abs(r - m["recall"]) < 1e-12 and abs(f - m["f1"]) < 1e-12, m


def test_running_loss_prints_the_reference_lines_in_order_without_reading_the_loss_each_step():
    """training.RunningLoss (round 4): the reference's `running_loss += loss.item()` ... every 10 iterations print running_loss / 10 (:845-851),
    kept as a tensor sum and read back in blocks of ten.  Same numbers, same order, same format; a partial last block prints nothing (the
    reference's `if iteration % 10 == 9`); more blocks pending than read-back slots forces an early flush, never a lost line."""
    logs = []
    rl = T.RunningLoss("cpu", every=10, log=logs.append, slots=4)
    vals = [0.5 * i - 3.0 for i in range(57)]
    for it, v in enumerate(vals):
        rl.add(torch.tensor(v), 2, it)
    rl.flush(wait=True)
    want = ["[%d, %5d] training loss: %.4f" % (2, 10 * (b + 1), sum(vals[10 * b:10 * b + 10]) / 10) for b in range(5)]
    assert logs == want and len(rl.values) == 5
    assert abs(rl.values[2] - sum(vals[20:30]) / 10) < 1e-6


def test_train_loop_logs_the_running_loss_like_the_reference(tmp_path):
    opt = make_opt(epochs=1, self_epochs=1, best_model_path=str(tmp_path / "ckpt"), model_id="t2")
    model = TinyModel()
    optim = torch.optim.SGD(model.parameters(), lr=0.0)
    logs = []
    T.train(list(batches(25 * 2, 2)), list(batches(4, 4)), model, [optim], "cpu", num_unpred_pairs=0, opt=opt, log=logs.append)
    lines = [l for l in logs if "training loss" in l]
    assert lines == ["[1,    10] training loss: 1.0000", "[1,    20] training loss: 1.0000"]       # 25 steps: two full blocks, like ref :848-851
    assert logs.index(lines[-1]) < next(i for i, l in enumerate(logs) if "precision" in l)        # printed before the epoch's evaluation line
