"""Host data path of the hot path's callers: `read_ECPE_data`, `ECPEDataset`, bag-of-words vocabulary.

Mirrors drl_classifier_ec_mmd_final_mul.py :86-146 (ECPEDataset), :631-731 (read_ECPE_data) and
bow_util.py :20-81 of the reference -- same names, argument meaning, outputs and quirks -- but written for
current pandas / scikit-learn (the reference relies on `DataFrame.append` and `get_feature_names`, both
removed) and without the reference's module globals: `tokenizer`, `bow` and `opt` are explicit arguments.
This stays on the host by design: it runs once per dataset, not per step.
"""
import ast
import random
import re

import numpy as np
import pandas as pd
import torch

_DOC_HEADER = re.compile(r"[0-9]{1,4}[\s][0-9]{1,2}")      # the reference's document-header test (:639)
_NON_ZH = re.compile(u"[^一-龥]")                   # non-Chinese unicode range (:102, bow_util.py:14)


# ----------------------------------------------------------------------------------------------
# bag-of-words vocabulary (bow_util.py)
# ----------------------------------------------------------------------------------------------
def _default_segmenter():
    try:
        import jieba
    except ImportError as e:      # jieba is what the reference uses (:105); it is not vendored here
        raise ImportError("Chinese word segmentation needs `jieba` (as in the reference) or an explicit "
                          "`segmenter=` callable returning a list of words") from e
    return jieba.lcut


def tokenize_zh(text, segmenter=None):
    """bow_util.py:13-17: strip every non-CJK character, then segment."""
    seg = segmenter or _default_segmenter()
    return seg(_NON_ZH.sub(r"", text))


def _corpus_sentences(file_path, keep_spaces=False):
    """Clause texts of an ECPE-format file in order (bow_util.py:21-35 / :58-75)."""
    out = []
    with open(file_path, encoding="utf8") as f:
        while True:
            line = f.readline()
            if not line:
                break
            if _DOC_HEADER.search(line):
                doc_len = int(line.strip().split(" ")[1])
                f.readline()                                  # pair info
                for _ in range(doc_len):
                    txt = f.readline().strip().split(",")[3]
                    out.append(txt if keep_spaces else txt.replace(" ", ""))
    return out


def _count_vocabulary(corpus, tokenizer=None):
    """Sorted feature names of sklearn's CountVectorizer fitted on `corpus` (get_feature_names() in the
    reference, bow_util.py:39 / :80; get_feature_names_out() in scikit-learn >= 1.0)."""
    from sklearn.feature_extraction.text import CountVectorizer
    vec = CountVectorizer(tokenizer=tokenizer, token_pattern=None) if tokenizer else CountVectorizer()
    vec.fit(corpus)
    return list(vec.get_feature_names_out())


def get_bow_zh(file_path, segmenter=None):
    """bow_util.py:20-40."""
    return _count_vocabulary(_corpus_sentences(file_path), tokenizer=lambda t: tokenize_zh(t, segmenter))


def bow_tokenize(sentence, tokenizer=None):
    """bow_util.py:42-48."""
    sentence = re.sub(r"[^\w\s]", "", sentence.lower())
    return [x for x in [t.replace("Ġ", "") for t in sentence.split(" ")] if x != ""]


def get_bow_en(file_path, bow_optimize=False, tokenizer=None):
    """bow_util.py:50-81 (note: with bow_optimize=False the spaces are removed, so each clause is one token)."""
    if not bow_optimize:
        corpus = _corpus_sentences(file_path)
    else:
        corpus = {"sep"}
        for line in _corpus_sentences(file_path, keep_spaces=True):
            corpus.update(bow_tokenize(line, tokenizer))
        corpus = sorted(corpus)
    return _count_vocabulary(corpus)


# ----------------------------------------------------------------------------------------------
# read_ECPE_data (:631-731)
# ----------------------------------------------------------------------------------------------
def _parse_pairs(line, language):
    if language == "zh":       # " (5,4), (6,5)"  -> eval of each ", "-separated chunk (:647-648)
        return [tuple(ast.literal_eval(x)) for x in line.strip().split(", ")]
    pairs = ast.literal_eval("[" + line.strip() + "]")          # en: " (2, 1)," (:650)
    return [(e, c) for e, c in pairs]


def read_ECPE_data(file_path, test=False, language="zh", rng=random):
    """Returns (df[pair,label,emotion], docs_pair_size, num_unpred_emotions) exactly like the reference:
    same row order, same use of `random.sample` for the training negatives (so `random.seed(42)`, :27,
    reproduces the reference's rows), same treatment of predicted emotions in test files."""
    rows, docs_pair_size, num_unpred = [], [], 0
    with open(file_path, encoding="utf8") as f:
        while True:
            line = f.readline()
            if not line:
                break
            if not _DOC_HEADER.search(line):
                continue
            doc_len = int(line.strip().split(" ")[1])
            pos_pairs = _parse_pairs(f.readline(), language)
            sentence_list, pred_emotions, sen_emo = [], [], {}
            for _ in range(doc_len):
                sentence = f.readline()
                sentence_list.append(sentence)
                parts = sentence.strip().split(",")
                sen_emotion, sen_id = int(parts[1]), int(parts[0])
                if sen_emotion != 6:
                    sen_emo[sen_id] = sen_emotion
                    pred_emotions.append(sen_id)
            if not test:
                emotions = list(dict.fromkeys(e for e, _ in pos_pairs))
            else:                                   # keep only pairs whose emotion clause was predicted (:669-682)
                keep, pre_e = [], -1
                for i, (e, _) in enumerate(pos_pairs):
                    if e not in pred_emotions and e != pre_e:
                        num_unpred += 1
                    elif e == pre_e:
                        keep.append(i)
                    else:
                        keep.append(i)
                        pred_emotions.remove(e)
                        pre_e = e
                pos_pairs = [pos_pairs[i] for i in keep]
                emotions = list(dict.fromkeys(e for e, _ in pos_pairs))
            causes = [c for _, c in pos_pairs]
            non_causes = [i + 1 for i in range(doc_len) if (i + 1) not in causes]
            neg_pairs = [(e, nc) for e in emotions for nc in non_causes]
            if not test:
                k = min(len(pos_pairs), len(neg_pairs))
                neg_pairs = rng.sample(neg_pairs, k)                     # (:699-701)
            else:
                for e in pred_emotions:                                  # (:704-708)
                    for c in range(1, doc_len + 1):
                        neg_pairs.append((e, c))

            def text(i):
                return sentence_list[i - 1].strip().split(",")[3].replace(" ", "")
            for e, c in pos_pairs:
                rows.append((text(e) + "[SEP]" + text(c), 1, sen_emo[e]))
            for e, c in neg_pairs:
                rows.append((text(e) + "[SEP]" + text(c), 0, sen_emo[e]))
            docs_pair_size.append(len(pos_pairs) + len(neg_pairs))
    df = pd.DataFrame(rows, columns=["pair", "label", "emotion"])
    if len(df) == 0:
        df = pd.DataFrame(columns=["pair", "label", "emotion"])
    return df, docs_pair_size, num_unpred


# ----------------------------------------------------------------------------------------------
# ECPEDataset (:86-146)
# ----------------------------------------------------------------------------------------------
class ECPEDataset(torch.utils.data.Dataset):
    """Same item contract as the reference (:136-144): dict with `input_ids`, `attention_masks`,
    `token_type_ids` (int64 [max_len]), `labels` f32 [1], `emo_labels` i64 [1], `cau_labels` f32 [1],
    `bow_reps` f32 [V].  Works with the stock DataLoader(batch_size, shuffle, num_workers=0).

    tokenizer: any object with the HF `encode_plus` interface (the reference's BertTokenizer /
    RobertaTokenizer).  bow: the vocabulary list from get_bow_zh / get_bow_en.
    pretokenize=True (default) tokenises every row once at construction instead of on every access, and the
    vocabulary lookup is a dict instead of the reference's O(V) list.index per word -- same outputs.
    """

    def __init__(self, df, tokenizer=None, bow=None, max_len=128, segmenter=None, pretokenize=True):
        self.tokenizer = tokenizer
        self.pairs = df["pair"].reset_index(drop=True)
        self.labels = df["label"].values
        self.emo_labels = df["emotion"].values
        self.cau_labels = df["label"].values               # (:92) cause label == pair label
        self.max_len = max_len
        self.bow_features = list(bow) if bow is not None else []
        self._bow_index = {}
        for i, w in enumerate(self.bow_features):          # first occurrence wins, like list.index
            self._bow_index.setdefault(w, i)
        self._segmenter = segmenter
        self.bow_representations = [self._get_bow_representations(p) for p in self.pairs]
        self._cache = None
        if pretokenize and tokenizer is not None and len(self.pairs):
            enc = [self._encode(str(p)) for p in self.pairs]
            self._cache = tuple(torch.stack([e[k] for e in enc]) for k in range(3))

    def __len__(self):
        return len(self.pairs)

    def _get_bow_representations(self, text_pair):
        """(:100-119) non-CJK characters are stripped even for English input."""
        rep = np.zeros(shape=len(self.bow_features), dtype=np.float32)
        text = _NON_ZH.sub(r"", str(text_pair))
        if text:
            seg = self._segmenter or _default_segmenter()
            for word in seg(text):
                j = self._bow_index.get(word)
                if j is not None:
                    rep[j] += 1
        rep /= np.max([np.sum(rep), 1])
        return rep

    def _encode(self, pair):
        inputs = self.tokenizer.encode_plus(pair, None, add_special_tokens=True, max_length=self.max_len,
                                            padding="max_length", return_token_type_ids=True, truncation=True,
                                            return_attention_mask=True, return_tensors="pt")
        return (inputs["input_ids"].flatten().to(torch.long), inputs["attention_mask"].flatten().to(torch.long),
                inputs["token_type_ids"].flatten().to(torch.long))

    def __getitem__(self, index):
        if self._cache is not None:
            ids, att, tt = (c[index] for c in self._cache)
        else:
            ids, att, tt = self._encode(str(self.pairs[index]))
        return {
            "input_ids": ids,
            "attention_masks": att,
            "token_type_ids": tt,
            "labels": torch.FloatTensor([self.labels[index]]),
            "emo_labels": torch.LongTensor([self.emo_labels[index]]),
            "cau_labels": torch.FloatTensor([self.cau_labels[index]]),
            "bow_reps": torch.from_numpy(self.bow_representations[index]).clone(),
        }


# ----------------------------------------------------------------------------------------------
# BatchLoader: drop-in for `DataLoader(dataset, batch_size, shuffle, num_workers=0)` (ref :952-961)
# ----------------------------------------------------------------------------------------------
class _Batch(dict):
    """A collated batch whose expensive fields may be gathered on first access (BatchLoader: the bag-of-words rows of a whole
    evaluation set).  Behaves like the plain dict a DataLoader yields: keys(), items(), `in` and [] all see the lazy fields."""

    def __init__(self):
        super().__init__()
        self._lazy = {}

    def _materialise(self, k):
        t, idx, pin = self._lazy.pop(k)
        nt = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            out = t.index_select(0, idx)
        finally:
            torch.set_num_threads(nt)
        super().__setitem__(k, out.pin_memory() if pin else out)

    def __missing__(self, k):
        if k in self._lazy:
            self._materialise(k)
            return super().__getitem__(k)
        raise KeyError(k)

    def __contains__(self, k):
        return super().__contains__(k) or k in self._lazy

    def get(self, k, default=None):
        return self[k] if k in self else default

    def _all(self):
        for k in list(self._lazy):
            self._materialise(k)

    def keys(self):
        self._all()
        return super().keys()

    def items(self):
        self._all()
        return super().items()

    def values(self):
        self._all()
        return super().values()

    def __iter__(self):
        self._all()
        return super().__iter__()

    def __len__(self):
        return super().__len__() + len(self._lazy)


class BatchLoader:
    """Yields the same dict-of-tensors batches as `torch.utils.data.DataLoader(ECPEDataset, batch_size, shuffle,
    num_workers=0)` -- same keys, dtypes, shapes and, under the same `torch.manual_seed`, the same sample order (the index
    stream of torch's RandomSampler is reproduced) -- but each batch is ONE row gather per field from tensors stacked at
    construction instead of `batch_size` `__getitem__` calls and a collate (64 x 95 KB bag-of-words rows cloned and
    re-stacked per batch at V = 23 771).  pin_memory=True allocates each batch in page-locked memory (as the stock
    loader's flag does); that allocation is slow, so it only pays when the copies are overlapped with compute.  Adds `seq_lengths` (host list of attended lengths) so the model can skip padding without reading
    the mask back from the device.  Needs a dataset built with a tokenizer (pretokenize=True)."""

    def __init__(self, dataset, batch_size=64, shuffle=False, drop_last=False, pin_memory=False, generator=None):
        if dataset._cache is None and len(dataset):
            raise ValueError("BatchLoader needs ECPEDataset(..., tokenizer=..., pretokenize=True)")
        self.dataset, self.batch_size, self.shuffle, self.drop_last, self.generator = dataset, int(batch_size), shuffle, drop_last, generator
        n = len(dataset)
        ids, att, tt = dataset._cache if n else (torch.zeros((0, dataset.max_len), dtype=torch.long),) * 3
        V = len(dataset.bow_features)
        bow = torch.from_numpy(np.stack(dataset.bow_representations)) if n else torch.zeros((0, V))
        self.fields = {
            "input_ids": ids, "attention_masks": att, "token_type_ids": tt,
            "labels": torch.as_tensor(np.asarray(dataset.labels, dtype=np.float32)).view(-1, 1),
            "emo_labels": torch.as_tensor(np.asarray(dataset.emo_labels, dtype=np.int64)).view(-1, 1),
            "cau_labels": torch.as_tensor(np.asarray(dataset.cau_labels, dtype=np.float32)).view(-1, 1),
            "bow_reps": bow.to(torch.float32),
        }
        self.lengths = att.sum(1).tolist()
        self.pin = bool(pin_memory) and torch.cuda.is_available()
        self._sparse = None

    def enable_sparse_bow(self):
        """Yield the bag-of-words targets as padded per-sample entry lists -- `bow_cols` int32 [B, M] (-1 = padding) and
        `bow_vals` f32 [B, M], M = the largest entry count of the dataset -- INSTEAD of the dense `bow_reps` rows (95 KB per
        sample at V = 23 771).  For PrefetchLoader, which ships the entries and expands them on the device."""
        if self._sparse is None:
            bow = self.fields["bow_reps"]
            n, V = bow.shape
            counts = (bow != 0).sum(1)
            M = max(1, int(counts.max()) if n else 1)
            M = (M + 7) & ~7
            cols = torch.full((n, M), -1, dtype=torch.int32)
            vals = torch.zeros((n, M), dtype=torch.float32)
            nz = bow.nonzero(as_tuple=False)                      # row-major: entries of a row are consecutive
            if nz.numel():
                start = torch.cumsum(counts, 0) - counts
                pos = torch.arange(nz.shape[0]) - start[nz[:, 0]]
                cols[nz[:, 0], pos] = nz[:, 1].to(torch.int32)
                vals[nz[:, 0], pos] = bow[nz[:, 0], nz[:, 1]]
            self._sparse = (cols, vals, V)
        return self

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _order(self):
        n = len(self.dataset)
        if not self.shuffle:
            return torch.arange(n)
        g = self.generator
        if g is None:
            # what DataLoader + RandomSampler draw from the global stream (torch 2.x): the iterator's base seed first
            # (unused with num_workers=0), then the sampler's seed for a private generator
            torch.empty((), dtype=torch.int64).random_()
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            g = torch.Generator()
            g.manual_seed(seed)
        return torch.randperm(n, generator=g)

    def __iter__(self):
        order = self._order()
        n = len(order)
        for s in range(0, n, self.batch_size):
            idx = order[s:s + self.batch_size]
            if len(idx) < self.batch_size and self.drop_last:
                return
            batch = _Batch()
            # single-threaded on purpose: a parallel gather leaves the OpenMP pool spin-waiting on every core, which starves
            # the HIP runtime's threads -- measured: a whole training epoch 5x slower (tools/bench_train_epoch.py)
            nt = torch.get_num_threads()
            torch.set_num_threads(1)
            try:
                for k, t in self.fields.items():
                    if k == "bow_reps" and self._sparse is not None:
                        batch["bow_cols"] = self._sparse[0].index_select(0, idx)
                        batch["bow_vals"] = self._sparse[1].index_select(0, idx)
                        batch["bow_dim"] = self._sparse[2]
                        continue
                    if k == "bow_reps" and len(idx) > 256:
                        # a whole evaluation set as ONE batch (ref :958): nothing on that path reads the bag-of-words rows
                        # (get_pair_preds takes ids / masks / types), and gathering them is 95 KB per pair -- 42 ms of a 270-ms
                        # epoch for the 1 938-pair test split.  Built on first access instead.
                        batch._lazy["bow_reps"] = (t, idx, self.pin)
                        continue
                    out = t.index_select(0, idx)      # (the out= form of index_select is ~60x slower on CPU)
                    batch[k] = out.pin_memory() if self.pin else out
            finally:
                torch.set_num_threads(nt)
            batch["seq_lengths"] = [self.lengths[i] for i in idx.tolist()]
            yield batch


# ----------------------------------------------------------------------------------------------
# Synthetic ECPE-shaped batches: the tokenizer vocabulary and the corpus' BoW vocabulary
# cannot be built offline, so benchmarks and smoke runs draw batches with the measured statistics of
# data/all_data_pair_zh.txt -- pair-length distribution (mean 26.9 tokens, clipped to [4, S]), label balance 0.5 with at
# least one positive, the emotion histogram of the train split, 3-12 bag-of-words entries per pair.
# ----------------------------------------------------------------------------------------------
EMOTION_HISTOGRAM = np.array([578, 705, 234, 98, 535, 437], dtype=np.float64)      # society_num train split, classes 0..5


def synthetic_ecpe_batch(B, S, vocab_size, V, seed=1, shape="A", pad_id=0, first_id=1, binary_emotion=False):
    """One batch with the keys / dtypes of `ECPEDataset.__getitem__` after collation (ref :136-144).
    shape "A": every position attended (the dense roofline workload); "B": ECPE-like lengths (~77 % padding).
    first_id: smallest token id drawn (2 for RoBERTa, whose id 1 is <pad>); binary_emotion: the float 0/1 emotion label of
    drl_classifier_en.py (:132) instead of the six-way class index."""
    rs = np.random.RandomState(seed)
    ids = rs.randint(first_id, vocab_size, size=(B, S)).astype(np.int64)
    att = np.ones((B, S), dtype=np.int64)
    if shape == "B":
        ln = np.clip(np.round(rs.gamma(shape=6.0, scale=26.9 / 6.0, size=B)), 4, S).astype(np.int64)
        for b in range(B):
            ids[b, ln[b]:] = pad_id
            att[b, ln[b]:] = 0
    tt = np.zeros((B, S), dtype=np.int64)
    y = (rs.uniform(size=(B, 1)) < 0.5).astype(np.float32)
    if y.sum() == 0:
        y[0, 0] = 1.0
    emo = rs.choice(6, size=(B, 1), p=EMOTION_HISTOGRAM / EMOTION_HISTOGRAM.sum()).astype(np.int64)
    bow = np.zeros((B, V), dtype=np.float32)
    for b in range(B):
        k = rs.randint(3, 13)
        cols = rs.randint(0, V, size=k)
        np.add.at(bow[b], cols, 1.0)
        bow[b] /= max(bow[b].sum(), 1.0)
    t = torch.from_numpy
    out = dict(input_ids=t(ids), attention_masks=t(att), token_type_ids=t(tt), labels=t(y), emo_labels=t(emo),
               cau_labels=t(y.copy()), bow_reps=t(bow))
    if binary_emotion:
        rs2 = np.random.RandomState(seed + 7919)
        out["emo_labels"] = t((rs2.uniform(size=(B, 1)) < 0.5).astype(np.float32))
    return out


# ----------------------------------------------------------------------------------------------
# PrefetchLoader: the seven `.to(device)` copies of every step (ref :823-830) taken off the step's critical path
# ----------------------------------------------------------------------------------------------
class SyntheticECPEDataset(ECPEDataset):
    """An ECPEDataset whose caches are filled from `synthetic_ecpe_batch` (no tokenizer / corpus vocabulary offline): n pairs of
    ECPE-shaped lengths, bag-of-words width V.  Same `__getitem__` contract; works with DataLoader / BatchLoader / PrefetchLoader."""

    def __init__(self, n, V, seed, max_len=128, vocab_size=21128, shape="B"):
        b = synthetic_ecpe_batch(n, max_len, vocab_size, V, seed=seed, shape=shape)
        self.pairs = pd.Series(["x"] * n)
        self.labels = b["labels"].view(-1).numpy()
        self.emo_labels = b["emo_labels"].view(-1).numpy()
        self.cau_labels = self.labels
        self.max_len, self.bow_features, self.tokenizer = max_len, [None] * V, object()
        self.bow_representations = list(b["bow_reps"].numpy())
        self._cache = (b["input_ids"], b["attention_masks"], b["token_type_ids"])


class PackedLengths(list):
    """The host list of attended lengths (what `seq_lengths=` takes) that ALSO carries the batch's token-packing arrays already
    on the device (`cu`, `tok_row`, `n_tokens`, `t_eff`), so that the model has nothing to copy per step."""
    cu = tok_row = None
    n_tokens = t_eff = 0


class PrefetchLoader:
    """Wraps a loader of dict batches (BatchLoader or a stock DataLoader over ECPEDataset) and yields the same batches with
    every tensor ALREADY ON THE DEVICE:

      * a background thread pulls batches ahead of the training loop (`depth` batches, default 3) and packs each into ONE
        page-locked staging block -- ids / mask / token types / labels, and the bag-of-words targets as (row, column, value)
        triples (a pair has 3-30 vocabulary words of V = 23 771: ~10 KB instead of 6 MB dense);
      * one async H2D copy per batch on a dedicated copy stream into a preallocated device slot; the training stream only
        waits for that copy's event and expands the triples on the device (`carel_bow_expand`);
      * a slot is refilled only when the GPU is done with it (two event queries on the host; with `depth` slots the host may
        run `depth` batches ahead of the GPU before that ever waits).

    The reference's loop body is unchanged: `batch[k].to(device, ...)` of a tensor that is already there is a no-op.
    A batch's tensors are views of a recycled device slot: they are valid until the NEXT batch is requested (what a training
    loop needs; clone what must live longer).
    `seq_lengths` (host list) is added like BatchLoader does, so the model can skip padding without a device read-back.
    Values are bit-identical to the wrapped loader's (tests/test_gpu_training.py)."""

    _INT_KEYS = ("input_ids", "attention_masks", "token_type_ids")

    def __init__(self, loader, device="cuda", depth=3, max_nnz_per_row=64):
        import threading
        self.loader, self.device, self.depth = loader, torch.device(device), max(2, int(depth))
        self.max_nnz_per_row = int(max_nnz_per_row)
        self._threading = threading
        self._slots = None
        if isinstance(loader, BatchLoader):
            loader.enable_sparse_bow()             # entries straight from the dataset: no dense 6 MB gather, no non-zero scan

    def __len__(self):
        return len(self.loader)

    # ---- packing (host, producer thread) ------------------------------------------------------
    @staticmethod
    def pack_layout(B, S, emo_is_float, nnz_cap):
        """word offsets (4-byte units) of the fields inside one staging block"""
        o, lay = 0, {}

        def take(name, n):
            nonlocal o
            lay[name] = (o, n)
            o = (o + n + 1) & ~1                              # every field starts 8-byte aligned (int64 views)
        for k in PrefetchLoader._INT_KEYS:
            take(k, B * S * 2)                                # int64 as two words
        take("labels", B)
        take("cau_labels", B)
        take("emo_labels", B if emo_is_float else 2 * B)
        take("trip", 3 * nnz_cap)
        Bp = B
        while (Bp * S) % 128:
            Bp += 1
        take("cu", Bp + 1)                                    # token packing (filled by the native packer only)
        take("tok", (B * S + 127) // 128 * 128)
        return lay, o

    @staticmethod
    def sparsify(bow):
        """dense [B, V] f32 -> (rows i32, cols i32, vals f32) of its non-zeros, row-major order (slow path: a scan of the dense
        block; loaders that know their entries -- BatchLoader -- hand them over instead)"""
        nz = bow.nonzero(as_tuple=False)
        rows, cols = nz[:, 0].to(torch.int32), nz[:, 1].to(torch.int32)
        return rows, cols, bow[nz[:, 0], nz[:, 1]].contiguous()

    @staticmethod
    def _bow_dim(batch):
        return int(batch["bow_dim"]) if "bow_cols" in batch else batch["bow_reps"].shape[1]

    def _make_slots(self, batch):
        B, S = batch["input_ids"].shape
        V = self._bow_dim(batch)
        emo_f = batch["emo_labels"].dtype.is_floating_point
        cap = B * (batch["bow_cols"].shape[1] if "bow_cols" in batch else self.max_nnz_per_row)
        lay, words = self.pack_layout(B, S, emo_f, cap)
        slots = []
        for _ in range(self.depth):
            slots.append(dict(host=torch.empty(words, dtype=torch.int32).pin_memory(), dev=torch.empty(words, dtype=torch.int32, device=self.device),
                              bow=torch.empty((B, V), dtype=torch.float32, device=self.device),
                              copied=torch.cuda.Event(), released=None))
        self._geom = (B, S, V, emo_f, cap, lay, words)
        self._slots = slots
        self._copy_stream = torch.cuda.Stream(device=self.device)

    def _pack(self, batch, slot):
        B, S, V, emo_f, cap, lay, words = self._geom
        h = slot["host"]
        for k in self._INT_KEYS:
            o, n = lay[k]
            h[o:o + n].view(torch.int64).view(B, S).copy_(batch[k])
        for k in ("labels", "cau_labels"):
            o, n = lay[k]
            h[o:o + n].view(torch.float32).copy_(batch[k].reshape(-1).to(torch.float32))
        o, n = lay["emo_labels"]
        if emo_f:
            h[o:o + n].view(torch.float32).copy_(batch["emo_labels"].reshape(-1).to(torch.float32))
        else:
            h[o:o + n].view(torch.int64).copy_(batch["emo_labels"].reshape(-1))
        o, _ = lay["trip"]
        if "bow_cols" in batch:                    # padded entry lists: rows are implied, padding columns are -1 (skipped on the device)
            M = batch["bow_cols"].shape[1]
            nnz = B * M
            h[o:o + nnz].view(B, M).copy_(torch.arange(B, dtype=torch.int32).view(B, 1).expand(B, M))
            h[o + nnz:o + 2 * nnz].view(B, M).copy_(batch["bow_cols"])
            h[o + 2 * nnz:o + 3 * nnz].view(torch.float32).view(B, M).copy_(batch["bow_vals"])
            return nnz
        rows, cols, vals = self.sparsify(batch["bow_reps"])
        nnz = int(rows.numel())
        if nnz > cap:
            raise ValueError("PrefetchLoader: %d bag-of-words entries in one batch exceed the staging capacity %d (raise max_nnz_per_row)" % (nnz, cap))
        h[o:o + nnz].copy_(rows)
        h[o + nnz:o + 2 * nnz].copy_(cols)
        h[o + 2 * nnz:o + 3 * nnz].view(torch.float32).copy_(vals)
        return nnz

    # ---- iteration ----------------------------------------------------------------------------
    def __iter__(self):
        """Who does what (measured, tools/bench_train_epoch.py, one ECPE-sized epoch: 0.29 s with a plain BatchLoader):
          * BatchLoader underneath: NO thread.  The batch is assembled by one C call (~30 us) in the consuming thread, one
            batch ahead of use, and copied by the copy stream while the current step runs: 0.28 s.
          * any other loader (a stock DataLoader's Python collate takes ~1.2 ms per batch): a background thread runs the
            wrapped loader and packs; it does CPU work only.  EVERY HIP call (the async copy, the events) is still issued by
            the consuming thread: a second thread calling into the HIP runtime while the training thread launches ~450
            kernels per step made the epoch 1.8x slower (0.53 s), with a short or the default GIL switch interval, with
            hipEventSynchronize or with polling; a CPU-only thread costs ~10 % (0.33 s) in interpreter-lock hand-overs."""
        import collections
        import queue
        q = queue.Queue()
        free = self._threading.Semaphore(self.depth)          # staging blocks the producer may fill
        stop = self._threading.Event()

        def acquire_slot():
            while not free.acquire(timeout=0.05):
                if stop.is_set():
                    return False
            return not stop.is_set()

        def native_items():
            """BatchLoader underneath: the batch is gathered from the dataset's stacked arrays by ONE C call."""
            from . import _lib as L
            import ctypes as C
            import time
            bl = self.loader
            f, (cols, vals, V) = bl.fields, bl._sparse
            n_all, S = f["input_ids"].shape
            B, M = bl.batch_size, cols.shape[1]
            emo = f["emo_labels"].reshape(-1).contiguous()
            lab, cau = f["labels"].reshape(-1).contiguous(), f["cau_labels"].reshape(-1).contiguous()
            order = bl._order().to(torch.int64).contiguous()
            lib = L.load()
            lay = self._geom[5] if self._slots is not None else None
            a = L.HostPackArgs()
            a.input_ids, a.attention_masks, a.token_type_ids = f["input_ids"].data_ptr(), f["attention_masks"].data_ptr(), f["token_type_ids"].data_ptr()
            a.labels, a.cau_labels, a.emo_labels = lab.data_ptr(), cau.data_ptr(), emo.data_ptr()
            a.bow_cols, a.bow_vals = cols.data_ptr(), vals.data_ptr()
            a.n_samples, a.batch, a.seq_len, a.bow_entries, a.emo_is_float = n_all, B, S, M, int(emo.dtype.is_floating_point)
            lens32 = torch.as_tensor(bl.lengths, dtype=torch.int32)
            Bp = B
            while (Bp * S) % 128:
                Bp += 1
            a.lengths, a.batch_padded = lens32.data_ptr(), Bp
            if lay is not None:
                a.off_cu, a.off_tok = lay["cu"][0], lay["tok"][0]
                a.off_input_ids, a.off_attention_masks, a.off_token_type_ids = lay["input_ids"][0], lay["attention_masks"][0], lay["token_type_ids"][0]
                a.off_labels, a.off_cau_labels, a.off_emo_labels, a.off_trip = lay["labels"][0], lay["cau_labels"][0], lay["emo_labels"][0], lay["trip"][0]
            n = 0
            for s0 in range(0, n_all, B):
                idx = order[s0:s0 + B].contiguous()
                if idx.numel() < B:
                    if bl.drop_last:
                        return
                    b = {k: t.index_select(0, idx) for k, t in f.items() if k != "bow_reps"}
                    b.update(bow_cols=cols.index_select(0, idx), bow_vals=vals.index_select(0, idx), bow_dim=V,
                             seq_lengths=[bl.lengths[i] for i in idx.tolist()])
                    yield dict(kind="plain", batch=b)
                    continue
                i = n % self.depth
                n += 1
                slot = self._slots[i]
                while not self._slot_idle(slot):               # the host is `depth` batches ahead of the GPU: let it catch up
                    time.sleep(1e-4)
                a.idx, a.dst = idx.data_ptr(), slot["host"].data_ptr()
                L.check(lib.carel_host_pack_batch(C.byref(a)), "carel_host_pack_batch")
                lens = PackedLengths(bl.lengths[j] for j in idx.tolist())
                lens.t_eff, lens.n_tokens = int(a.t_eff), int(a.t_pad)
                yield dict(kind="slot", slot=i, nnz=B * M, lengths=lens, att_host=None)

        def producer_generic():
            n = 0
            nt = torch.get_num_threads()
            for batch in self.loader:
                if not acquire_slot():
                    return
                Bb = batch["input_ids"].shape[0]
                if Bb != self._geom[0]:                    # short last batch (once per epoch): plain copies by the consumer
                    q.put(dict(kind="plain", batch=batch))
                    continue
                i = n % self.depth
                n += 1
                torch.set_num_threads(1)                   # see BatchLoader: a spinning OpenMP pool starves the HIP runtime threads
                nnz = self._pack(batch, self._slots[i])
                torch.set_num_threads(nt)
                q.put(dict(kind="slot", slot=i, nnz=nnz, lengths=batch.get("seq_lengths"),
                           att_host=None if "seq_lengths" in batch else batch["attention_masks"]))

        native = isinstance(self.loader, BatchLoader) and self.loader._sparse is not None

        def producer():
            try:
                producer_generic()
            except BaseException as e:                          # surface loader errors in the consumer
                q.put(e)
                return
            q.put(None)

        # the slots exist before the thread starts (device allocations and page-locking belong to the consuming thread)
        if native:
            bl = self.loader
            f, (cols, vals, V) = bl.fields, bl._sparse
            B, S, M = bl.batch_size, f["input_ids"].shape[1], cols.shape[1]
            if len(bl.dataset) >= B and (self._slots is None or self._geom[:3] != (B, S, V) or self._geom[4] != B * M):
                self._make_slots(dict(input_ids=f["input_ids"][:B], emo_labels=f["emo_labels"][:B], bow_cols=cols[:B], bow_dim=V))
        else:
            it = iter(self.loader)
            try:
                first = next(it)
            except StopIteration:
                return
            if self._slots is None or (first["input_ids"].shape[1], self._bow_dim(first)) != (self._geom[1], self._geom[2]) or first["input_ids"].shape[0] > self._geom[0]:
                self._make_slots(first)
            del it, first                                        # (re-iterated by the producer: loaders here are re-iterable)
        th = None
        if native:
            src = native_items()
        else:
            th = self._threading.Thread(target=producer, daemon=True)
            th.start()
        issued = collections.deque()        # items whose H2D copy has been enqueued (consumed in order)
        parked = []                         # staging blocks handed back by the training loop that their copy may still be reading
        held, done = None, False
        main = torch.cuda.current_stream(self.device)

        def poll_parked():
            freed = [x for x in parked if self._slot_idle(self._slots[x])]
            for sl in freed:
                parked.remove(sl)
                free.release()
            return bool(freed)

        def next_item(block):
            """-> an item, None at the end of the epoch, or queue.Empty (class) when nothing is ready and block is False"""
            if native:
                return next(src, None)
            while True:
                try:
                    return q.get(timeout=1e-3) if block else q.get_nowait()
                except queue.Empty:
                    if not block:
                        return queue.Empty
                    poll_parked()                               # the producer may be waiting for a staging block

        def issue(item):
            if isinstance(item, dict) and item["kind"] == "slot":
                slot = self._slots[item["slot"]]
                with torch.cuda.stream(self._copy_stream):        # (the slot is idle: _slot_idle was true before it was packed)
                    slot["dev"].copy_(slot["host"], non_blocking=True)
                    slot["copied"].record(self._copy_stream)
            issued.append(item)

        try:
            while True:
                if held is not None:                            # all work on the previous batch has been enqueued by now
                    if held >= 0:
                        ev = torch.cuda.Event()
                        ev.record(main)
                        self._slots[held]["released"] = ev
                        if not native:
                            parked.append(held)
                    elif not native:
                        free.release()
                    held = None
                poll_parked()
                while not done and len(issued) < 2:             # keep one batch's copy in flight behind the one being consumed
                    item = next_item(block=not issued)
                    if item is queue.Empty:
                        break
                    if item is None:
                        done = True
                    else:
                        issue(item)
                if not issued:
                    return
                item = issued.popleft()
                if isinstance(item, BaseException):
                    raise item
                if item["kind"] == "plain":
                    b = item["batch"]
                    out = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in b.items()}
                    if "bow_cols" in out:                       # entry lists -> dense targets (not slot-backed)
                        from . import ops
                        Bb, M = b["bow_cols"].shape
                        trip = torch.cat((torch.arange(Bb, dtype=torch.int32).view(Bb, 1).expand(Bb, M).reshape(-1), b["bow_cols"].reshape(-1),
                                          b["bow_vals"].reshape(-1).view(torch.int32))).to(self.device)
                        out["bow_reps"] = ops.bow_expand(trip, Bb * M, torch.empty((Bb, int(b["bow_dim"])), dtype=torch.float32, device=self.device))
                        for k in ("bow_cols", "bow_vals", "bow_dim"):
                            del out[k]
                    if "seq_lengths" not in out:
                        out["seq_lengths"] = b["attention_masks"].sum(1).tolist()
                    held = -1
                    yield out
                    continue
                held = item["slot"]
                yield self._unpack(item)
        finally:
            stop.set()
            if th is not None:
                th.join(timeout=2.0)

    @staticmethod
    def _slot_idle(slot):
        """True when the GPU is done with both halves of a slot: the copy out of its staging block and every kernel that read
        its device block (`released`, recorded when the training loop asked for the following batch).  Checked on the HOST
        before the slot is refilled.  A stream-side wait instead (copy stream waits for `released`) was measured to make
        the epoch 1.9x slower once the host runs ahead of the GPU: HIP multiplexes streams onto 4 hardware queues, and the
        barrier packet parks whatever shares the copy stream's queue -- here the weight-gradient side stream -- until a step
        from several steps ago has finished."""
        return slot["copied"].query() and (slot["released"] is None or slot["released"].query())

    def _unpack(self, item):
        from . import ops
        B, S, V, emo_f, cap, lay, words = self._geom
        slot = self._slots[item["slot"]]
        torch.cuda.current_stream(self.device).wait_event(slot["copied"])
        d = slot["dev"]
        out = {}
        for k in self._INT_KEYS:
            o, n = lay[k]
            out[k] = d[o:o + n].view(torch.int64).view(B, S)
        for k in ("labels", "cau_labels"):
            o, n = lay[k]
            out[k] = d[o:o + n].view(torch.float32).view(B, 1)
        o, n = lay["emo_labels"]
        out["emo_labels"] = d[o:o + n].view(torch.float32).view(B, 1) if emo_f else d[o:o + n].view(torch.int64).view(B, 1)
        o, _ = lay["trip"]
        nnz = item["nnz"]
        out["bow_reps"] = ops.bow_expand(d[o:o + 3 * nnz] if nnz else d[o:o], nnz, slot["bow"])
        out["seq_lengths"] = item["lengths"] if item["lengths"] is not None else item["att_host"].sum(1).tolist()
        if isinstance(out["seq_lengths"], PackedLengths) and out["seq_lengths"].n_tokens:
            pl = out["seq_lengths"]
            o, n = lay["cu"]
            pl.cu = d[o:o + n]
            o, _ = lay["tok"]
            pl.tok_row = d[o:o + pl.n_tokens]
        return out
