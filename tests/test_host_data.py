"""Host data path (read_ECPE_data / ECPEDataset / BoW vocabulary) against golden vectors produced by the
reference's own read_ECPE_data (tests/golden/gen_golden_data.py)."""
import hashlib
import json
import os
import random

import numpy as np
import pytest
import torch

from carel_vae_amd import data as D

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "ecpe_data.json"), encoding="utf8"))


def digest(df):
    h = hashlib.sha1()
    for p, l, e in zip(df["pair"], df["label"], df["emotion"]):
        h.update(("%s|%d|%d\n" % (p, int(l), int(e))).encode("utf8"))
    return h.hexdigest()


@pytest.mark.parametrize("name", list(GOLD["samples"]))
def test_read_ecpe_samples_match_reference(name):
    g = GOLD["samples"][name]
    random.seed(42)
    df, sizes, unpred = D.read_ECPE_data(os.path.join(HERE, "golden", "ecpe", name), test=g["test"], language=g["language"])
    assert len(df) == g["rows"] and sizes == g["docs_pair_size"] and unpred == g["num_unpred"]
    assert [[p, int(l), int(e)] for p, l, e in zip(df["pair"][:6], df["label"][:6], df["emotion"][:6])] == g["head"]
    assert digest(df) == g["digest"]
    assert list(df.columns) == ["pair", "label", "emotion"]


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference checkout not present (GPU box)")
@pytest.mark.parametrize("rel", list(GOLD["reference_files"]))
def test_read_ecpe_reference_files(rel):
    g = GOLD["reference_files"][rel]
    random.seed(42)
    df, sizes, unpred = D.read_ECPE_data(os.path.join("/root/reference", rel), test=g["test"], language=g["language"])
    assert (len(df), len(sizes), unpred) == (g["rows"], g["docs"], g["num_unpred"])
    assert digest(df) == g["digest"]


class FakeTokenizer:
    """HF `encode_plus` interface: [CLS]=101, one id per character (ord % 1000 + 200), [SEP]=102, pad 0."""

    def encode_plus(self, text, text_pair=None, add_special_tokens=True, max_length=128, padding="max_length",
                    return_token_type_ids=True, truncation=True, return_attention_mask=True, return_tensors="pt"):
        ids = [101] + [ord(c) % 1000 + 200 for c in text][:max_length - 2] + [102]
        att = [1] * len(ids) + [0] * (max_length - len(ids))
        ids = ids + [0] * (max_length - len(ids))
        t = lambda v: torch.tensor([v])
        return {"input_ids": t(ids), "attention_mask": t(att), "token_type_ids": t([0] * max_length)}


def char_segmenter(text):
    return list(text)


def test_dataset_item_contract_and_bow():
    random.seed(42)
    df, _, _ = D.read_ECPE_data(os.path.join(HERE, "golden", "ecpe", "sample_zh_train.txt"))
    bow = sorted({c for p in df["pair"] for c in D._NON_ZH.sub("", p)})
    for pre in (True, False):
        ds = D.ECPEDataset(df, FakeTokenizer(), bow, max_len=128, segmenter=char_segmenter, pretokenize=pre)
        assert len(ds) == len(df)
        it = ds[3]
        assert set(it) == {"input_ids", "attention_masks", "token_type_ids", "labels", "emo_labels", "cau_labels", "bow_reps"}
        assert it["input_ids"].dtype == torch.int64 and it["input_ids"].shape == (128,)
        assert it["attention_masks"].dtype == torch.int64 and it["token_type_ids"].shape == (128,)
        assert it["labels"].dtype == torch.float32 and it["labels"].shape == (1,)
        assert it["emo_labels"].dtype == torch.int64 and it["cau_labels"].dtype == torch.float32
        assert torch.equal(it["labels"], it["cau_labels"])                     # ref :92
        assert it["bow_reps"].dtype == torch.float32 and it["bow_reps"].shape == (len(bow),)
        assert abs(float(it["bow_reps"].sum()) - 1.0) < 1e-6                   # count / max(sum, 1)   ref :116
        # reference semantics of one representation, recomputed the slow way (list.index per character)
        txt = D._NON_ZH.sub("", df["pair"][3])
        ref = np.zeros(len(bow), dtype=np.float32)
        for ch in txt:
            if ch in bow:
                ref[bow.index(ch)] += 1
        ref /= max(ref.sum(), 1)
        np.testing.assert_allclose(it["bow_reps"].numpy(), ref)
    loader = torch.utils.data.DataLoader(ds, batch_size=5, shuffle=False, num_workers=0)
    b = next(iter(loader))
    assert b["input_ids"].shape == (5, 128) and b["bow_reps"].shape == (5, len(bow)) and b["labels"].shape == (5, 1)


def test_english_bow_targets_are_all_zero():
    """non-CJK characters are stripped before the BoW lookup, so English targets are zero."""
    random.seed(42)
    df, _, _ = D.read_ECPE_data(os.path.join(HERE, "golden", "ecpe", "sample_en_train.txt"), language="en")
    ds = D.ECPEDataset(df, FakeTokenizer(), ["letter", "storm"], segmenter=char_segmenter)
    assert float(ds[0]["bow_reps"].abs().sum()) == 0.0


def test_bow_vocabularies():
    p = os.path.join(HERE, "golden", "ecpe", "sample_zh_train.txt")
    v = D.get_bow_zh(p, segmenter=char_segmenter)
    assert v == sorted(set(v)) and "雨" in v and all(len(w) == 1 for w in v)
    ven = D.get_bow_en(os.path.join(HERE, "golden", "ecpe", "sample_en_train.txt"))
    assert len(ven) == 7            # spaces are removed first (bow_util.py:70): one "word" per clause
    vopt = D.get_bow_en(os.path.join(HERE, "golden", "ecpe", "sample_en_train.txt"), bow_optimize=True)
    assert "sep" in vopt and "storm" in vopt and "villagers" in vopt


def test_batch_loader_equals_stock_dataloader():
    """BatchLoader yields what DataLoader(dataset, batch_size, shuffle, num_workers=0) yields: same keys, dtypes, shapes,
    values and (same torch seed) sample order; plus the host-side attended lengths."""
    random.seed(42)
    df, _, _ = D.read_ECPE_data(os.path.join(HERE, "golden", "ecpe", "sample_zh_train.txt"), test=False, language="zh")
    bow = D.get_bow_zh(os.path.join(HERE, "golden", "ecpe", "sample_zh_train.txt"), segmenter=char_segmenter)
    ds = D.ECPEDataset(df, tokenizer=FakeTokenizer(), bow=bow, max_len=64, segmenter=char_segmenter)
    for shuffle, drop_last, bs in ((False, False, 4), (True, False, 4), (True, True, 5)):
        torch.manual_seed(123)
        ref = list(torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=shuffle, num_workers=0, drop_last=drop_last))
        torch.manual_seed(123)
        fast = D.BatchLoader(ds, batch_size=bs, shuffle=shuffle, drop_last=drop_last, pin_memory=False)
        got = list(fast)
        assert len(got) == len(ref) == len(fast)
        for a, b in zip(ref, got):
            assert set(b) == set(a) | {"seq_lengths"}
            for k in a:
                assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and torch.equal(a[k], b[k]), (k, shuffle)
            assert b["seq_lengths"] == a["attention_masks"].sum(1).tolist()


@pytest.mark.parametrize("name", list(GOLD["samples_en_script"]))
def test_english_script_reader_and_dataset(name):
    """drl_classifier_en.py's own read_ECPE_data (:748-813: two return values, no emotion column) and dataset (:77-138: the
    emotion label is the constant 1, as a FloatTensor) -- against the reference function run by gen_golden_data.py."""
    from carel_vae_amd import drl_classifier_en as ME
    g = GOLD["samples_en_script"][name]
    random.seed(42)
    df, sizes = ME.read_ECPE_data(os.path.join(HERE, "golden", "ecpe", name.split(":")[0]), test=g["test"])
    assert list(df.columns) == g["columns"] and len(df) == g["rows"] and sizes == g["docs_pair_size"]
    assert [[p, int(l)] for p, l in zip(df["pair"][:6], df["label"][:6])] == g["head"]
    h = hashlib.sha1()
    for p, l in zip(df["pair"], df["label"]):
        h.update(("%s|%d\n" % (p, int(l))).encode("utf8"))
    assert h.hexdigest() == g["digest"]
    ds = ME.ECPEDataset(df, tokenizer=FakeTokenizer(), bow=["the", "letter"], max_len=32, segmenter=char_segmenter)
    item = ds[0]
    assert item["emo_labels"].dtype == torch.float32 and float(item["emo_labels"]) == 1.0
    assert item["labels"].dtype == torch.float32 and item["input_ids"].shape == (32,)
    assert set(item) == {"input_ids", "attention_masks", "token_type_ids", "labels", "emo_labels", "cau_labels", "bow_reps"}


def test_native_host_batch_packer_matches_the_stacked_arrays():
    """carel_host_pack_batch (the GIL-free gather PrefetchLoader's background thread calls): every field of the staging block
    equals the row gather of the dataset's stacked arrays; bag-of-words entry lists with -1 padding; repeated indices."""
    import ctypes as C
    import pandas as pd
    from carel_vae_amd import _lib as L

    class SynthDataset(D.ECPEDataset):
        def __init__(self, n, V, seed):
            b = D.synthetic_ecpe_batch(n, 16, 100, V, seed=seed, shape="B")
            self.pairs = pd.Series(["x"] * n)
            self.labels = b["labels"].view(-1).numpy(); self.emo_labels = b["emo_labels"].view(-1).numpy(); self.cau_labels = self.labels
            self.max_len, self.bow_features, self.tokenizer = 16, [None] * V, object()
            self.bow_representations = list(b["bow_reps"].numpy())
            self._cache = (b["input_ids"], b["attention_masks"], b["token_type_ids"])
    bl = D.BatchLoader(SynthDataset(50, 300, 1), batch_size=8).enable_sparse_bow()
    f, (cols, vals, V) = bl.fields, bl._sparse
    B, S, M = 8, 16, cols.shape[1]
    # the entry lists reproduce the dense rows
    dense = torch.zeros(50, V)
    for r in range(50):
        for m in range(M):
            if int(cols[r, m]) >= 0:
                dense[r, int(cols[r, m])] = vals[r, m]
    assert torch.equal(dense, f["bow_reps"])
    lay, words = D.PrefetchLoader.pack_layout(B, S, False, B * M)
    assert all(o % 2 == 0 for o, _ in lay.values())               # int64 views need 8-byte alignment
    dst = torch.full((words,), -7, dtype=torch.int32)
    idx = torch.tensor([3, 49, 0, 7, 7, 12, 30, 41])
    emo, lab, cau = (f[k].reshape(-1).contiguous() for k in ("emo_labels", "labels", "cau_labels"))
    a = L.HostPackArgs()
    a.input_ids, a.attention_masks, a.token_type_ids = (f[k].data_ptr() for k in ("input_ids", "attention_masks", "token_type_ids"))
    a.labels, a.cau_labels, a.emo_labels = lab.data_ptr(), cau.data_ptr(), emo.data_ptr()
    a.bow_cols, a.bow_vals, a.idx, a.dst = cols.data_ptr(), vals.data_ptr(), idx.data_ptr(), dst.data_ptr()
    a.n_samples, a.batch, a.seq_len, a.bow_entries, a.emo_is_float = 50, B, S, M, 0
    a.off_input_ids, a.off_attention_masks, a.off_token_type_ids = lay["input_ids"][0], lay["attention_masks"][0], lay["token_type_ids"][0]
    a.off_labels, a.off_cau_labels, a.off_emo_labels, a.off_trip = lay["labels"][0], lay["cau_labels"][0], lay["emo_labels"][0], lay["trip"][0]
    L.check(L.load().carel_host_pack_batch(C.byref(a)))
    for k in ("input_ids", "attention_masks", "token_type_ids"):
        o, n = lay[k]
        assert torch.equal(dst[o:o + n].view(torch.int64).view(B, S), f[k][idx]), k
    o, n = lay["emo_labels"]
    assert torch.equal(dst[o:o + n].view(torch.int64), emo[idx])
    o, n = lay["labels"]
    assert torch.equal(dst[o:o + n].view(torch.float32), lab[idx])
    o, _ = lay["trip"]
    nn = B * M
    assert torch.equal(dst[o:o + nn].view(B, M), torch.arange(B, dtype=torch.int32).view(B, 1).expand(B, M))
    assert torch.equal(dst[o + nn:o + 2 * nn].view(B, M), cols[idx])
    assert torch.equal(dst[o + 2 * nn:o + 3 * nn].view(torch.float32).view(B, M), vals[idx])
    # token packing arrays: the same cu / tok_row DrlClassifier._pack_info builds from the length list
    lens = torch.as_tensor(bl.lengths, dtype=torch.int32)
    a.lengths, a.batch_padded, a.off_cu, a.off_tok = lens.data_ptr(), 8, lay["cu"][0], lay["tok"][0]
    L.check(L.load().carel_host_pack_batch(C.byref(a)))
    ll = [bl.lengths[i] for i in idx.tolist()]
    t_eff = sum(ll)
    assert a.t_eff == t_eff and a.t_pad == (t_eff + 127) // 128 * 128
    cu = dst[lay["cu"][0]:lay["cu"][0] + 9]
    assert cu.tolist() == [sum(ll[:b]) for b in range(8)] + [t_eff]
    tok = dst[lay["tok"][0]:lay["tok"][0] + min(int(a.t_pad), B * S)]
    want = [b * S + s_ for b in range(8) for s_ in range(ll[b])]
    assert tok[:t_eff].tolist() == want and (tok[t_eff:] == -1).all()
    idx[0] = 50
    assert L.load().carel_host_pack_batch(C.byref(a)) != 0        # out-of-range index is refused, not read
