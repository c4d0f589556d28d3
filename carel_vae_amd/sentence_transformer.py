"""MI355X implementation of the sentence-embedding fine-tune of chi_ec_sentence_transformer.py (the zh script: a
BERT-architecture SimCSE checkpoint) and en_ec_sentence_transformer.py (the English script: all-mpnet-base-v2 = MPNet encoder +
mean pooling + Normalize) -- the names the scripts use from the third-party `sentence_transformers` package:

    from carel_vae_amd.sentence_transformer import SentenceTransformer, InputExample, losses
    model = SentenceTransformer(...)                                              (:22)
    train_loss = losses.BatchSemiHardTripletLoss(model=model, margin=4.45)        (:78)
    model.fit(train_objectives=[(train_dataloader, train_loss)], epochs=..., warmup_steps=..., output_path=...)   (:84-87)

What runs where: the encoder is the same bf16-MFMA BERT stack as the VAE path (carel_encoder_forward / _backward_layer, every
token of the last layer kept: cls_only_last off); mean pooling, the batch-semi-hard triplet loss (forward + gradient), the
global gradient norm and AdamW (decoupled weight decay on the weight matrices, clip coefficient read from device memory) are
HIP kernels (csrc/triplet.hip, csrc/adam.hip).  The loop of `fit` (AdamW lr 2e-5 / weight_decay 0.01, WarmupLinear schedule,
max_grad_norm 1) follows the package's published defaults.  PARITY UNPINNED: the package is absent from this container and from
the reference tree; oracle/carel_oracle_st.py restates its published algorithm and tests/test_gpu_triplet.py holds this module to
that restatement.  Pretrained checkpoints cannot be fetched offline: the constructor takes an encoder configuration (random
init) or a state dict with HF BertModel / MPNetModel key names.  MPNet (encoder_config("mpnet")): the relative-position attention
bias is added inside the attention kernels (csrc/attention.hip, REL instantiations; its table gradient is accumulated by distance and
folded into the 32 buckets), the embeddings have no token types, and the sentence embedding is L2-normalised (models.Normalize);
the MPNet encoder restatement in oracle/carel_oracle.py is pinned to the installed transformers MPNetModel
(tests/test_oracle_triplet.py).

Two known deviations from what the reference's scripts most likely ran (both PARITY UNPINNED, stated here because they cannot be checked):
  * optimiser: fit() defaults to torch.optim.AdamW semantics -- eps 1e-8, decoupled decay applied BEFORE the update.  The 2.x releases of
    sentence-transformers of the reference's era (09/2021) defaulted to transformers.AdamW: eps 1e-6 and decay applied AFTER the update.
    `optimizer_params={"eps": 1e-6}` selects that epsilon; the decay order differs by a factor (1 - lr * wd) on the update of one step
    (2e-7 relative at the default lr 2e-5, wd 0.01).
  * sequence length: max_seq_length defaults to 128 because the attention kernels hold one (sample, head) per workgroup with S <= 128;
    all-mpnet-base-v2 ships max_seq_length 384.  ECPE clauses are far shorter (p99 70-89 tokens); a sentence that IS longer
    is truncated by the tokenizer, and tokenize() warns once when that happens instead of truncating silently.
"""
import ctypes as C
import math
import os
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import drl_classifier as M

H = 768
TT_KEY = "encoder.embeddings.token_type_embeddings.weight"


class InputExample:
    """sentence_transformers.InputExample: texts (list of str), label."""

    def __init__(self, guid="", texts=None, label=0):
        self.guid, self.texts, self.label = guid, texts, label


class _EmbedFn(torch.autograd.Function):
    """tokens -> sentence embeddings [B, 768]; backward fills the encoder's parameter gradients."""

    @staticmethod
    def forward(ctx, anchor, st, call):
        ctx.st, ctx.call = st, call
        return st._forward(call, training=True)

    @staticmethod
    def backward(ctx, g):
        ctx.st._backward(ctx.call, g.contiguous().float())
        return None, None, None


class _TripletFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, labels, margin):
        emb = emb.contiguous().float()
        loss = torch.empty(1, device=emb.device, dtype=torch.float32)
        demb = torch.empty_like(emb)
        L.check(L.load().carel_triplet_semihard(emb.data_ptr(), labels.data_ptr(), emb.shape[0], emb.shape[1], float(margin), loss.data_ptr(),
                                                demb.data_ptr(), L.current_stream()), "carel_triplet_semihard")
        ctx.save_for_backward(demb)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        demb, = ctx.saved_tensors
        return demb * g, None, None


class _Losses:
    class BatchSemiHardTripletLoss(nn.Module):
        """losses.BatchSemiHardTripletLoss(model, margin=5) with the package's default Euclidean distance.
        forward(sentence_features, labels) as the package calls it from fit(); labels: integer class per sentence."""

        def __init__(self, model, margin=5.0):
            super().__init__()
            self.sentence_embedder, self.triplet_margin = model, float(margin)

        def forward(self, sentence_features, labels):
            rep = self.sentence_embedder(sentence_features[0])["sentence_embedding"]
            return self.batch_semi_hard_triplet_loss(labels, rep)

        def batch_semi_hard_triplet_loss(self, labels, embeddings):
            if embeddings.shape[0] > 64:
                raise L.CarelError("BatchSemiHardTripletLoss: at most 64 sentences per batch (the reference uses 16)")
            lab = labels.to(embeddings.device, torch.int32).reshape(-1).contiguous()
            return _TripletFn.apply(embeddings, lab, self.triplet_margin)


losses = _Losses


class SentenceTransformer(nn.Module):
    """Transformer (BERT-base architecture) + mean pooling.  encoder_cfg: carel_vae_amd.encoder_config(...) (default zh
    BERT-base); tokenizer: any object with the HF `encode_plus` interface; state_dict: HF BertModel weights
    (`embeddings.*`, `encoder.layer.*`; a `pooler.*` entry is accepted and ignored like models.Transformer ignores it)."""

    def __init__(self, encoder_cfg=None, tokenizer=None, max_seq_length=128, seed=None, state_dict=None, normalize=None):
        super().__init__()
        if isinstance(encoder_cfg, str):                 # the checkpoint names the reference scripts pass (weights are NOT fetched)
            encoder_cfg = M.encoder_config("mpnet" if "mpnet" in encoder_cfg.lower() else "zh")
        cfg = encoder_cfg if encoder_cfg is not None else M.encoder_config("zh")
        self.mpnet = bool(getattr(cfg, "rel_pos", False))
        # modules.json of all-mpnet-base-v2: Transformer, Pooling(mean), Normalize; the zh SimCSE checkpoint: Transformer, Pooling
        self.normalize = self.mpnet if normalize is None else bool(normalize)
        # the engine: a DrlClassifier whose encoder (flat fp32 parameters + bf16 shadow, workspaces, kernels) is what runs; its
        # VAE heads are never touched
        self._m = M.DrlClassifier(M.make_opt(pair_bow_dim=8), cfg, seed=seed)
        self._m.cls_only_last = False            # mean pooling reads every attended token of the last layer
        self.tokenizer, self.max_seq_length = tokenizer, int(max_seq_length)
        self._fwd = 0
        if self.mpnet:                                   # MPNetEmbeddings has no token types: the engine's row stays zero
            with torch.no_grad():
                self._m._named[TT_KEY].zero_()
        if state_dict is not None:
            self.load_state_dict(state_dict)

    # ---- parameters: the BertModel without its pooler ---------------------------------------------------------------
    def _enc_keys(self):
        return [k for k in self._m._order if k.startswith("encoder.") and not k.startswith("encoder.pooler.") and not (self.mpnet and k == TT_KEY)]

    # MPNetModel names its attention sub-modules attn.q / k / v / o and attention.LayerNorm; the engine stores BERT names
    _MPNET = ((".attention.attn.q.", ".attention.self.query."), (".attention.attn.k.", ".attention.self.key."),
              (".attention.attn.v.", ".attention.self.value."), (".attention.attn.o.", ".attention.output.dense."),
              (".attention.LayerNorm.", ".attention.output.LayerNorm."))

    def _to_internal(self, k):
        if self.mpnet:
            for hf, ours in self._MPNET:
                k = k.replace(hf, ours)
        return k

    def _to_public(self, k):
        if self.mpnet:
            for hf, ours in self._MPNET:
                k = k.replace(ours, hf)
        return k

    def parameters(self, recurse=True):
        return iter([self._m._named[k] for k in self._enc_keys()])

    def named_parameters(self, prefix="", recurse=True):
        return iter([(self._to_public(k[len("encoder."):]), self._m._named[k]) for k in self._enc_keys()])

    def state_dict(self, *a, **k):
        return {self._to_public(k_[len("encoder."):]): self._m._named[k_].detach().clone() for k_ in self._m._order
                if k_.startswith("encoder.") and not (self.mpnet and k_ == TT_KEY)}

    def load_state_dict(self, sd, strict=True):
        full = self._m.state_dict()
        for k, v in sd.items():
            if k.endswith("position_ids"):
                continue
            kk = "encoder." + self._to_internal(k)
            if kk not in full or (self.mpnet and kk == TT_KEY):
                if strict and kk != TT_KEY:
                    raise KeyError(k)
                continue
            full[kk] = v
        self._m.load_state_dict(full)
        return self

    def to(self, device):
        self._m.to(device)
        return self

    def train(self, mode=True):
        self._m.train(mode)
        return super().train(mode)

    # ---- tokenisation -----------------------------------------------------------------------------------------------
    def tokenize(self, texts):
        if self.tokenizer is None:
            raise L.CarelError("SentenceTransformer was built without a tokenizer: pass token tensors instead of strings")
        rows = [self.tokenizer.encode_plus(t, None, add_special_tokens=True, max_length=self.max_seq_length, padding="max_length",
                                           return_token_type_ids=True, truncation=True, return_attention_mask=True, return_tensors="pt")
                for t in texts]
        if not getattr(self, "_warned_truncation", False) and any(int(r["attention_mask"].sum()) >= self.max_seq_length for r in rows):
            import warnings
            warnings.warn("SentenceTransformer.tokenize: a sentence fills all %d positions and was probably truncated (the attention kernels "
                          "cap the sequence at 128 tokens; all-mpnet-base-v2 itself allows 384)" % self.max_seq_length)
            self._warned_truncation = True
        cat = lambda k: torch.cat([r[k].reshape(1, -1).to(torch.long) for r in rows], 0)
        return {"input_ids": cat("input_ids"), "attention_mask": cat("attention_mask"), "token_type_ids": cat("token_type_ids")}

    def smart_batching_collate(self, batch):
        """fit()'s collate: a list of InputExample -> ([features], labels)."""
        texts = [e.texts[0] for e in batch]
        labels = torch.tensor([e.label for e in batch])
        return [self.tokenize(texts)], labels

    # ---- forward / backward -----------------------------------------------------------------------------------------
    def _make_call(self, features):
        m = self._m
        m._require_cuda()
        dev = m._flat.device
        ids = features["input_ids"].to(dev)
        att = features.get("attention_mask", features.get("attention_masks")).to(dev)
        tt = features.get("token_type_ids")
        B, S = ids.shape
        Bp = m._padded_batch(B, S)
        c = SimpleNamespace(B=B, S=S, Bp=Bp)
        c.ids, c.att = m._prep_ids(ids, Bp), m._prep_ids(att, Bp)
        c.tt = None if tt is None else m._prep_ids(tt.to(dev), Bp)
        lens = features.get("seq_lengths")
        if lens is None:
            lens = att.sum(1).tolist()                        # host list (one read-back; loaders that know the lengths pass them)
        c.lens = [int(v) for v in lens]
        c.pack = m._pack_info(c.att, B, Bp, S, c.lens)
        self._fwd += 1
        c.seed = (m.dropout_base_seed * 1000003 + 7919 * self._fwd) & 0xFFFFFFFF
        rows = c.pack.n_tokens if c.pack is not None else Bp * S
        c.rows = rows
        if c.pack is not None:
            row0 = c.pack.cu[:B]
        else:
            row0 = torch.arange(B, dtype=torch.int32, device=dev) * S
        c.row0 = row0.to(torch.int32).contiguous()
        c.len_dev = torch.tensor(c.lens, dtype=torch.int32).to(dev)
        # sample of every encoder row (-1 = padding / filler): for the pooling backward
        rs = np.full(rows, -1, dtype=np.int32)
        o = 0
        for b, n in enumerate(c.lens):
            if c.pack is not None:
                rs[o:o + n] = b
                o += n
            else:
                rs[b * S:b * S + n] = b
        c.row_sample = torch.from_numpy(rs).to(dev)
        return c

    def _forward(self, c, training):
        m = self._m
        if m._adam_hook is not None:
            m._adam_hook._join()
        m._refresh_shadow()
        ws = m._workspace(c.Bp, c.S, inference=not training)
        ea = m._encoder_args(c.ids, c.att, c.tt, ws, c.Bp, c.S, not training, m.training, c.seed, 0, c.pack, None)
        lib = L.load()
        L.check(lib.carel_encoder_forward(C.byref(ea), L.current_stream()), "carel_encoder_forward")
        x_last = lib.carel_encoder_x_last(C.byref(ea))
        emb = torch.empty((c.B, H), device=m._flat.device, dtype=torch.float32)
        L.check(lib.carel_mean_pool_fwd(x_last, c.row0.data_ptr(), c.len_dev.data_ptr(), c.B, H, emb.data_ptr(), L.current_stream()), "carel_mean_pool_fwd")
        c.ea, c.ws = ea, ws
        if self.normalize:                                   # models.Normalize
            c.y, c.norm = torch.empty_like(emb), torch.empty(c.B, device=emb.device, dtype=torch.float32)
            L.check(lib.carel_l2_normalize_fwd(emb.data_ptr(), c.B, H, c.y.data_ptr(), c.norm.data_ptr(), L.current_stream()), "carel_l2_normalize_fwd")
            return c.y
        return emb

    def _backward(self, c, g):
        m = self._m
        lib = L.load()
        first = m._named[m._order[0]]
        accumulate = first.grad is not None
        prev = m._flat_grad.clone() if accumulate else None
        key = ("st_dx", c.rows)
        dx = m._ws.get(key)
        if dx is None:
            dx = m._ws[key] = torch.empty((c.rows, H), device=g.device, dtype=torch.float32)
        if self.normalize:
            gp = torch.empty_like(g)
            L.check(lib.carel_l2_normalize_bwd(g.data_ptr(), c.y.data_ptr(), c.norm.data_ptr(), c.B, H, gp.data_ptr(), L.current_stream()), "carel_l2_normalize_bwd")
            g = gp
        L.check(lib.carel_mean_pool_bwd(g.data_ptr(), c.row_sample.data_ptr(), c.len_dev.data_ptr(), c.rows, H, dx.data_ptr(), L.current_stream()),
                "carel_mean_pool_bwd")
        c.ea.dx = dx.data_ptr()
        m._backward_encoder(c.ea, accumulate)
        if self.mpnet:                                       # no token types in MPNet: the placeholder row takes no gradient (and so never moves)
            m._grad_view(TT_KEY).zero_()
        if accumulate:
            m._flat_grad.add_(prev)
        m._bind_grads()

    def forward(self, features):
        """features: {"input_ids", "attention_mask", "token_type_ids"} -> the same dict plus "sentence_embedding" [B, 768]."""
        c = self._make_call(features)
        if torch.is_grad_enabled():
            anchor = self._m._flat.new_zeros((), requires_grad=True)
            emb = _EmbedFn.apply(anchor, self, c)
        else:
            emb = self._forward(c, training=False)
        out = dict(features)
        out["sentence_embedding"] = emb
        return out

    def encode(self, sentences, batch_size=32, convert_to_numpy=True):
        """SentenceTransformer.encode: eval mode, no gradients; strings (needs the tokenizer) or a features dict."""
        was = self.training
        self.train(False)
        outs = []
        with torch.no_grad():
            if isinstance(sentences, dict):
                n = sentences["input_ids"].shape[0]
                for s in range(0, n, batch_size):
                    outs.append(self.forward({k: v[s:s + batch_size] for k, v in sentences.items() if torch.is_tensor(v)})["sentence_embedding"])
            else:
                for s in range(0, len(sentences), batch_size):
                    outs.append(self.forward(self.tokenize(sentences[s:s + batch_size]))["sentence_embedding"])
        self.train(was)
        emb = torch.cat(outs, 0)
        return emb.cpu().numpy() if convert_to_numpy else emb

    # ---- training loop ----------------------------------------------------------------------------------------------
    def fit(self, train_objectives, epochs=1, steps_per_epoch=None, scheduler="WarmupLinear", warmup_steps=10000,
            optimizer_params=None, weight_decay=0.01, max_grad_norm=1.0, output_path=None, show_progress_bar=False, callback=None):
        """SentenceTransformer.fit for ONE (dataloader, loss) objective -- what both reference scripts pass (:84-87)."""
        if len(train_objectives) != 1:
            raise L.CarelError("fit: one (dataloader, loss) objective is supported (the reference scripts use one)")
        if scheduler != "WarmupLinear":
            raise L.CarelError("fit: only the WarmupLinear schedule (the package default, used by the reference) is built")
        loader, loss_model = train_objectives[0]
        if hasattr(loader, "collate_fn") and loader.collate_fn is not None and self.tokenizer is not None:
            try:
                loader.collate_fn = self.smart_batching_collate
            except Exception:
                pass
        # optimizer_params: lr (2e-5) and eps (1e-8 = torch.optim.AdamW; the sentence-transformers 2.x default optimiser, transformers.AdamW,
        # used 1e-6 -- module docstring)
        lr = float((optimizer_params or {}).get("lr", 2e-5))
        if steps_per_epoch is None:
            steps_per_epoch = len(loader)
        total = int(steps_per_epoch * epochs)
        optim = FusedAdamW(self, lr=lr, weight_decay=weight_decay, max_grad_norm=max_grad_norm,
                           eps=float((optimizer_params or {}).get("eps", 1e-8)))
        sched = WarmupLinear(optim, warmup_steps, total)
        self.train(True)
        losses_seen = []
        step = 0
        for epoch in range(epochs):
            it = iter(loader)
            for _ in range(steps_per_epoch):
                try:
                    features, labels = next(it)
                except StopIteration:
                    it = iter(loader)
                    features, labels = next(it)
                loss = loss_model(features, labels)
                loss.backward()
                optim.step()                                   # clip (device-side coefficient) + AdamW in one pass over the flat buffer
                optim.zero_grad()
                sched.step()
                losses_seen.append(loss.detach())
                step += 1
            if callback is not None:
                callback(None, epoch, step)
        if output_path is not None:
            self.save(output_path)
        self.last_fit = SimpleNamespace(losses=[float(x) for x in losses_seen], optimizer=optim)
        return self

    def save(self, path):
        os.makedirs(path, exist_ok=True)
        torch.save(self.state_dict(), os.path.join(path, "pytorch_model.bin"))
        cfg = self._m.cfg
        with open(os.path.join(path, "carel_sentence_transformer.txt"), "w") as f:
            f.write("architecture: %s (12 x 768), pooling: mean%s, max_seq_length: %d, vocab_size: %d\n" % (
                "MPNet-base" if self.mpnet else "BERT-base", ", normalize" if self.normalize else "", self.max_seq_length, cfg.vocab_size))


class WarmupLinear:
    """transformers.get_linear_schedule_with_warmup as the package's "WarmupLinear": lr = base * step / warmup while warming up,
    then base * (total - step) / (total - warmup).  LambdaLR semantics: the FIRST optimiser step runs at lambda(0)."""

    def __init__(self, optim, warmup_steps, total_steps):
        self.optim, self.warmup, self.total, self.n = optim, int(warmup_steps), int(total_steps), 0
        self.base = optim.param_groups[0]["lr"]
        self._set()

    def factor(self, n):
        if n < self.warmup:
            return float(n) / float(max(1, self.warmup))
        return max(0.0, float(self.total - n) / float(max(1, self.total - self.warmup)))

    def _set(self):
        self.optim.param_groups[0]["lr"] = self.base * self.factor(self.n)

    def step(self):
        self.n += 1
        self._set()


class FusedAdamW(M.FusedAdam):
    """torch.optim.AdamW over the encoder (weight_decay on the weight matrices and embedding tables, none on biases and
    LayerNorm parameters -- fit()'s two parameter groups) with torch.nn.utils.clip_grad_norm_ folded in: one reduction
    kernel leaves the clip coefficient in device memory, the fused Adam kernel multiplies it into every gradient."""

    NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")

    def __init__(self, st_model, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_grad_norm=1.0):
        m = st_model._m
        m._require_cuda()
        keys = st_model._enc_keys()
        hi = m._offs["encoder.pooler.dense.weight"]
        super().__init__(m, lr=lr, betas=betas, eps=eps, fuse_into_backward=False, param_range=(0, hi), params=[m._named[k] for k in keys])
        self.weight_decay, self.max_grad_norm = float(weight_decay), (None if max_grad_norm is None else float(max_grad_norm))
        segs = []
        for k in keys:
            if any(nd in k for nd in self.NO_DECAY):
                continue
            lo = m._offs[k]
            end = lo + m._named[k].numel()
            end4 = (end + 3) & ~3                             # (segments are 256-byte aligned in the flat buffer)
            if segs and segs[-1][1] >= lo:
                segs[-1][1] = end4
            else:
                segs.append([lo, end4])
        self._segs = torch.tensor(segs, dtype=torch.int64).to(m._flat.device).contiguous()
        self._norm_scratch = torch.empty(1024, device=m._flat.device, dtype=torch.float32)
        self._norm_out = torch.zeros(2, device=m._flat.device, dtype=torch.float32)

    def last_grad_norm(self):
        """||g||_2 before clipping of the most recent step (device tensor, no sync)."""
        return self._norm_out[0]

    def step(self):
        m = self.model
        lib = L.load()
        lo, hi = self._lo, self._hi
        a = L.AdamArgs()
        a.param, a.grad = m._flat.data_ptr() + 4 * lo, m._flat_grad.data_ptr() + 4 * lo
        a.exp_avg, a.exp_avg_sq = self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr()
        a.shadow_bf16 = m._shadow.data_ptr() + 2 * lo
        a.n, a.step = hi - lo, self.step_count + 1
        a.lr, a.beta1, a.beta2, a.eps = self.param_groups[0]["lr"], self.betas[0], self.betas[1], self.eps
        a.grad_scale = 1.0
        if self.max_grad_norm is not None:
            L.check(lib.carel_grad_norm_clip(a.grad, hi - lo, self.max_grad_norm, self._norm_scratch.data_ptr(), self._norm_out.data_ptr(),
                                             L.current_stream()), "carel_grad_norm_clip")
            a.grad_scale_dev = self._norm_out.data_ptr() + 4
        a.weight_decay = self.weight_decay
        if self.weight_decay != 0.0:
            a.decay_segments, a.n_decay_segments = self._segs.data_ptr(), self._segs.shape[0]
        L.check(lib.carel_adam_step(C.byref(a), L.current_stream()), "carel_adam_step")
        self.step_count += 1
        m.mark_shadow_fresh()
