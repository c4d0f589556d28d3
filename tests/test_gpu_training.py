"""The reference's driver end to end on the GPU: read_ECPE_data -> ECPEDataset -> DataLoader -> train() (step body of
drl_classifier_ec_mmd_final_mul.py :823-845 / the two-phase VI step) -> evaluation with get_pair_preds -> checkpoint ->
generate_self_train_data, on the committed ECPE sample with a 2-layer encoder and a character tokenizer stand-in."""
import os

import numpy as np
import pytest
import torch

from carel_vae_amd import data as D
from carel_vae_amd import drl_classifier as M
from carel_vae_amd import training as T
from tests.test_host_data import FakeTokenizer, char_segmenter

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ECPE = os.path.join(HERE, "golden", "ecpe")


def _loaders(bs):
    import random
    random.seed(42)
    train_df, _, _ = D.read_ECPE_data(os.path.join(ECPE, "sample_zh_train.txt"), test=False, language="zh")
    test_df, sizes, unpred = D.read_ECPE_data(os.path.join(ECPE, "sample_zh_test.txt"), test=True, language="zh")
    bow = D.get_bow_zh(os.path.join(ECPE, "sample_zh_train.txt"), segmenter=char_segmenter)
    tok = FakeTokenizer()
    tr = D.ECPEDataset(train_df, tokenizer=tok, bow=bow, max_len=128, segmenter=char_segmenter)
    te = D.ECPEDataset(test_df, tokenizer=tok, bow=bow, max_len=128, segmenter=char_segmenter)
    return (torch.utils.data.DataLoader(tr, batch_size=bs, shuffle=True, num_workers=0, drop_last=True),
            torch.utils.data.DataLoader(te, batch_size=len(te), shuffle=False, num_workers=0), test_df, sizes, unpred, len(bow))


@pytest.mark.parametrize("mode", ["mmd", "vi"])
def test_train_driver_end_to_end(tmp_path, mode):
    torch.manual_seed(0)
    train_loader, test_loader, test_df, sizes, unpred, V = _loaders(bs=4)
    assert len(train_loader) >= 2
    kw = dict(disentangle="vi", emotion_head="ce") if mode == "vi" else {}
    opt = M.make_opt(epochs=2, pair_bow_dim=V, best_model_path=str(tmp_path / "ckpt"), model_id="e2e", vae_lr=1e-4, **kw)
    model = M.DrlClassifier(opt, M.encoder_config("zh", vocab_size=1300, layers=2), seed=1).to("cuda")
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    if mode == "vi":
        aprx, other = model.get_params()
        optimizers = [torch.optim.Adam(aprx, lr=opt.aprx_lr), M.FusedAdam(model, lr=opt.vae_lr, fuse_into_backward=True)]
    else:
        optimizers = [M.FusedAdam(model, lr=opt.vae_lr, fuse_into_backward=True)]
    logs = []
    best = T.train(train_loader, test_loader, model, optimizers, "cuda", num_unpred_pairs=unpred, opt=opt, log=logs.append)
    torch.cuda.synchronize()
    assert best is model
    after = model.state_dict()
    moved = [k for k in before if not torch.equal(before[k], after[k])]
    assert any(k.startswith("encoder.encoder.layer.0.") for k in moved) and "decoder.weight" in moved
    assert all(torch.isfinite(v).all() for v in after.values())
    assert sum("f1 socre" in str(l) for l in logs) == 2                    # one evaluation per epoch
    # checkpoint written when F1 improved, and it loads back into a fresh model with the reference's key names
    ck = tmp_path / "ckpt" / "e2e.pt"
    if ck.exists():
        sd = torch.load(str(ck), map_location="cpu", weights_only=True)
        assert set(sd) == set(after)
        fresh = M.DrlClassifier(opt, M.encoder_config("zh", vocab_size=1300, layers=2), seed=2)
        T.load_ckp(str(ck), fresh)
    # pseudo-labelling pass over the test documents (:734-799)
    df = T.generate_self_train_data(sizes, test_df, test_loader, model, "random")
    assert list(df.columns) == ["pair", "label", "emotion"] and set(df["label"]) <= {0, 1}
    preds = model.get_pair_preds(*(next(iter(test_loader))[k].cuda() for k in ("input_ids", "attention_masks", "token_type_ids")))
    assert len(preds) == len(test_df) and all(p[0] in (0.0, 1.0) for p in preds)


def test_train_driver_english_adversarial(tmp_path):
    """drl_classifier_en.py's loop (six optimisers, :904-947) through the same driver: discriminators on fused RMSprop, the
    rest on fused Adam inside backward; evaluation thresholds sigmoid(logits) (:975-976)."""
    from carel_vae_amd import drl_classifier_en as ME
    torch.manual_seed(0)
    train_loader, test_loader, test_df, sizes, unpred, V = _loaders(bs=4)
    ds = train_loader.dataset
    ds.emo_labels = (np.asarray(ds.emo_labels) > 2).astype(np.int64)           # the English script's emotion label is binary (:132)
    opt = ME.make_opt(epochs=2, pair_bow_dim=V, best_model_path=str(tmp_path / "ckpt"), model_id="e2e_en", vae_lr=1e-4)
    model = ME.DrlClassifier(opt, M.encoder_config("en", vocab_size=1300, layers=2), seed=1).to("cuda")
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    optimizers = list(model.make_fused_optimizers(fuse_into_backward=True))
    logs = []
    best = T.train(train_loader, test_loader, model, optimizers, "cuda", num_unpred_pairs=unpred, opt=opt, log=logs.append)
    torch.cuda.synchronize()
    assert best is model
    after = model.state_dict()
    moved = {k for k in before if not torch.equal(before[k], after[k])}
    for k in ("content_disc.weight", "ec_disc.bias", "emotion_disc.weight", "content_classifier.weight", "decoder.bias",
              "encoder.encoder.layer.0.output.dense.weight", "encoder.embeddings.word_embeddings.weight"):
        assert k in moved, k
    assert "content_mu.weight" not in moved and "cause_log_var.bias" not in moved       # in no optimiser group (:357-376)
    assert all(torch.isfinite(v).all() for v in after.values())
    assert sum("f1 socre" in str(l) for l in logs) == 2
    df = T.generate_self_train_data(sizes, test_df, test_loader, model, "extreme")
    assert set(df["label"]) <= {0, 1}


@pytest.mark.parametrize("depth", [2, 3])
def test_prefetch_loader_yields_the_wrapped_loaders_batches_on_the_device(depth):
    """PrefetchLoader (pinned ring, one async H2D per batch on a copy stream, bag-of-words as triples expanded by
    carel_bow_expand): every field bit-identical to the wrapped loader's batch, over two epochs (slots recycled several
    times, GPU kept busy between batches so a too-early slot reuse would show), odd batch size, short last batch."""
    from carel_vae_amd.data import PrefetchLoader, synthetic_ecpe_batch
    batches = [synthetic_ecpe_batch(15, 32, 100, 700, seed=10 + i, shape="B") for i in range(7)]
    batches.append(synthetic_ecpe_batch(5, 32, 100, 700, seed=99, shape="B"))
    pl = PrefetchLoader(batches, "cuda", depth=depth)
    assert len(pl) == 8
    busy = torch.zeros(1 << 22, device="cuda")
    for epoch in range(2):
        seen = 0
        for got, want in zip(pl, batches):
            assert got["seq_lengths"] == want["attention_masks"].sum(1).tolist()
            for _ in range(20):
                busy.add_(1.0)                         # work that is still running when the next batch is requested
            for k, v in want.items():
                assert got[k].is_cuda and got[k].dtype == v.dtype and got[k].shape == v.shape, k
                assert torch.equal(got[k].cpu(), v), (epoch, seen, k)
            seen += 1
        assert seen == len(batches)
    # binary float emotion labels (the `_en` dataset) take the float slot layout
    fb = [synthetic_ecpe_batch(8, 16, 50, 90, seed=3 + i, shape="A", binary_emotion=True) for i in range(3)]
    for got, want in zip(PrefetchLoader(fb, "cuda"), fb):
        assert got["emo_labels"].dtype == torch.float32 and torch.equal(got["emo_labels"].cpu(), want["emo_labels"])
        assert torch.equal(got["bow_reps"].cpu(), want["bow_reps"])


def test_prefetch_loader_around_batch_loader_ships_entry_lists():
    """PrefetchLoader(BatchLoader): the bag-of-words targets travel as the dataset's padded entry lists (no dense gather, no
    non-zero scan) and come out as the same dense rows a plain BatchLoader yields; short last batch included."""
    import pandas as pd

    class SynthDataset(D.ECPEDataset):
        def __init__(self, n, V, seed):
            b = D.synthetic_ecpe_batch(n, 32, 100, V, seed=seed, shape="B")
            self.pairs = pd.Series(["x"] * n)
            self.labels = b["labels"].view(-1).numpy(); self.emo_labels = b["emo_labels"].view(-1).numpy(); self.cau_labels = self.labels
            self.max_len, self.bow_features, self.tokenizer = 32, [None] * V, object()
            self.bow_representations = list(b["bow_reps"].numpy())
            self._cache = (b["input_ids"], b["attention_masks"], b["token_type_ids"])
    ds = SynthDataset(77, 900, 4)
    want = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()} for b in D.BatchLoader(ds, batch_size=16)]
    pl = D.PrefetchLoader(D.BatchLoader(ds, batch_size=16), "cuda", depth=3)
    n = 0
    for got, w in zip(pl, want):
        assert set(got) == set(w), (set(got) ^ set(w))
        assert got["seq_lengths"] == w["seq_lengths"]
        for k, v in w.items():
            if torch.is_tensor(v):
                assert got[k].is_cuda and torch.equal(got[k].cpu(), v), (n, k)
        n += 1
    assert n == len(want) == 5


def test_train_driver_with_prefetch_loader_reproduces_the_batch_loader_run():
    """The reference's driver loop fed by PrefetchLoader (device-resident batches, packing arrays shipped with the batch) gives
    the same weights as the same loop fed by the BatchLoader it wraps: same batches, same packed-token path (the embedding
    backward accumulates with float atomics, so two runs of EITHER loader agree to fp32 summation order, not bit for bit)."""
    import pandas as pd

    class SynthDataset(D.ECPEDataset):
        def __init__(self, n, V, seed):
            b = D.synthetic_ecpe_batch(n, 32, 300, V, seed=seed, shape="B")
            self.pairs = pd.Series(["x"] * n)
            self.labels = b["labels"].view(-1).numpy(); self.emo_labels = b["emo_labels"].view(-1).numpy(); self.cau_labels = self.labels
            self.max_len, self.bow_features, self.tokenizer = 32, [None] * V, object()
            self.bow_representations = list(b["bow_reps"].numpy())
            self._cache = (b["input_ids"], b["attention_masks"], b["token_type_ids"])
    V = 257
    tr_ds, te_ds = SynthDataset(72, V, 11), SynthDataset(24, V, 12)
    outs = []
    for wrap in (False, True):
        opt = M.make_opt(epochs=1, pair_bow_dim=V, best_model_path="/tmp/carel_ckpt_pf", model_id="pf", dropout=0.0)
        model = M.DrlClassifier(opt, M.encoder_config("zh", vocab_size=300, layers=2, hidden_dropout=0.0, attn_dropout=0.0), seed=3).to("cuda")
        optim = M.FusedAdam(model, lr=1e-3)
        tr = D.BatchLoader(tr_ds, batch_size=16, shuffle=False)
        if wrap:
            tr = D.PrefetchLoader(tr, "cuda", depth=2)
        torch.manual_seed(5)                                   # the reparameterisation noise stream
        T.train(tr, D.BatchLoader(te_ds, batch_size=24), model, [optim], "cuda", num_unpred_pairs=0, opt=opt, log=lambda *_: None)
        outs.append({k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()})
    for k in outs[0]:
        # Adam moves every element by ~lr per step whatever the gradient's size: an element whose gradient is fp32 summation
        # noise around 0 may step the other way, so a few elements may differ by up to 2 * steps * lr; the rest must agree
        d = (outs[0][k].double() - outs[1][k].double()).abs()
        assert float(d.max()) <= 2 * 5 * 1e-3 * 1.01, k
        if not k.endswith("key.bias"):                 # analytically zero gradient (softmax shift invariance): pure rounding noise
            assert float((d <= 2e-4).double().mean()) >= 0.9, (k, float((d <= 2e-4).double().mean()))   # 20 % of one lr-sized move
