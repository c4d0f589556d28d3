"""`DrlClassifier` with the module surface of the reference's drl_classifier_en.py (:140-624): the three-space
(content / emotion / cause) adversarial model of config 4, on the RoBERTa-base encoder kernels of libcarel_hip.so.

Same constructor argument, attribute names, state_dict keys, `forward` (returns the seven-loss tuple, :334),
`get_pair_preds` (raw pair logits, :336-353) and `get_params` (six optimiser groups, :357-376), so the reference's step

    content_disc_opt.zero_grad(); (content_disc_loss_emo + content_disc_loss_cau).backward(retain_graph=True)
    ... four more discriminator backward calls ...; vae_and_cls_opt.zero_grad(); vae_and_cls_loss.backward()
    six optimiser steps                                                                         (:919-947)

runs unchanged.  All seven losses and every gradient are produced by the forward kernels (the losses are roots of the
graph); a discriminator's backward call only adds its share into `.grad`, the last one runs the encoder backward.
No CPU fallback: CPU tensors raise.
"""
import ast
import ctypes as C
import math
import random
from types import SimpleNamespace

import numpy as np
import pandas as pd
import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from . import data as _data
from .drl_classifier import DrlClassifier as _Base, FusedAdam, H, _Call, _Holder, encoder_config

DEFAULT_OPT = dict(language="en", max_len=128, ec_num_class=1, pair_num_class=1, ec_dim=24, con_dim=384, pair_bow_dim=23771, bert_dim=768,
                   kl_ann_iterations=20000, epochs=10, batch_size=64, ec_kl_lambda=0.03, con_kl_lambda=0.03, label_smoothing=0.1,
                   con_adv_loss_weight=0.03, ec_adv_loss_weight=1.0, ecce_adv_loss_weight=3.0, con_mul_loss_weight=3.0,
                   ec_mul_loss_weight=10.0, pair_mul_loss_weight=30.0, dropout=0.5, epsilon=1e-8, adv_lr=0.001, vae_lr=1e-5,
                   self_iteration=30, self_epochs=10, self_strategy="random", best_model_path="ECPE_model/best_drl_model", model_id="carel-en")

LOSS_NAMES = ("content_disc_loss_emo", "content_disc_loss_cau", "emotion_disc_loss", "ec_disc_loss", "cause_disc_loss", "ce_disc_loss",
              "vae_and_classifier_loss")
TERM_NAMES = LOSS_NAMES + ("content_entropy_emo", "content_entropy_cau", "emotion_entropy", "cause_entropy", "ec_entropy", "ce_entropy",
                           "emo_mul", "cau_mul", "content_mul", "pair_mul", "kl_e", "kl_c", "kl_content", "rec")
LATENT_HEADS = ("content_mu", "content_log_var", "emotion_mu", "emotion_log_var", "cause_mu", "cause_log_var")
DISC_GROUPS = ("content_disc", "emotion_disc", "cause_disc", "ec_disc", "ce_disc")           # get_params order (:365-369)
SMALL_DISCS = ("emotion_disc", "cause_disc", "ec_disc", "ce_disc")                            # carel_en_tail_args.sdisc_* order
OTHER_HEADS = ("decoder", "emotion_classifier", "cause_classifier", "pair_classifier", "content_classifier")   # :370-375
# which returned loss feeds which discriminator group: loss index -> (group, gradient image)
_LOSS_TO_GROUP = {0: ("content_disc", 0), 1: ("content_disc", 1), 2: ("emotion_disc", 0), 3: ("ec_disc", 0), 4: ("cause_disc", 0),
                  5: ("ce_disc", 0)}


def make_opt(**kw):
    """The reference's argparse namespace (drl_classifier_en.py:29-61) with its defaults."""
    d = dict(DEFAULT_OPT)
    d.update(kw)
    return SimpleNamespace(**d)


def read_ECPE_data(file_path, test=False, rng=random):
    """drl_classifier_en.py:748-813: (df[pair, label], docs_pair_size).  Same row order and the same use of random.sample for
    the training negatives, so `random.seed(42)` (:26) reproduces the reference's rows; test=True keeps every negative."""
    rows, docs_pair_size = [], []
    with open(file_path, encoding="utf8") as f:
        while True:
            line = f.readline()
            if not line:
                break
            if not _data._DOC_HEADER.search(line):
                continue
            doc_len = int(line.strip().split(" ")[1])
            pos_pairs = [(e, c) for e, c in ast.literal_eval("[" + f.readline().strip() + "]")]      # (:763-765)
            emotions = list(dict.fromkeys(e for e, _ in pos_pairs))
            causes = [c for _, c in pos_pairs]
            non_causes = [i + 1 for i in range(doc_len) if (i + 1) not in causes]
            neg_pairs = [(e, nc) for e in emotions for nc in non_causes]
            if not test:                                                                           # (:779-785)
                neg_pairs = rng.sample(neg_pairs, min(len(pos_pairs), len(neg_pairs)))
            sentence_list = [f.readline() for _ in range(doc_len)]

            def text(i):
                return sentence_list[i - 1].strip().split(",")[3].replace(" ", "")
            for e, c in pos_pairs:
                rows.append((text(e) + "[SEP]" + text(c), 1))
            for e, c in neg_pairs:
                rows.append((text(e) + "[SEP]" + text(c), 0))
            docs_pair_size.append(len(pos_pairs) + len(neg_pairs))
    return pd.DataFrame(rows, columns=["pair", "label"]), docs_pair_size


class ECPEDataset(_data.ECPEDataset):
    """drl_classifier_en.py:77-138: the emotion label of every pair is 1 (:82) and is a FloatTensor (:132); everything else as
    the main script's dataset."""

    def __init__(self, df, tokenizer=None, bow=None, max_len=128, segmenter=None, pretokenize=True):
        if "emotion" not in df.columns:
            df = df.assign(emotion=1)
        super().__init__(df, tokenizer=tokenizer, bow=bow, max_len=max_len, segmenter=segmenter, pretokenize=pretokenize)
        self.emo_labels = np.ones(len(df), dtype=np.int64)

    def __getitem__(self, index):
        item = super().__getitem__(index)
        item["emo_labels"] = torch.FloatTensor([float(self.emo_labels[index])])
        return item


class _EnLosses(torch.autograd.Function):
    """Whole-step forward returning the seven losses; each backward call receives the upstream gradients of the losses
    it was started from (None for the others)."""

    @staticmethod
    def forward(ctx, anchor, model, call):
        ctx.model, ctx.call = model, call
        ctx.set_materialize_grads(False)
        model._run_forward(call, training=True)
        return tuple(call.buf.terms[i].clone() for i in range(7))

    @staticmethod
    def backward(ctx, *grads):
        ctx.model._run_backward_en(ctx.call, grads)
        return None, None, None


class FusedRMSprop:
    """`torch.optim.RMSprop(params, lr)` (torch defaults) over one contiguous range of the model's flat buffer; used through
    zero_grad() / step() like the reference's discriminator optimisers (:919-947, :1056-1060)."""

    def __init__(self, model, lr, param_range, params, alpha=0.99, eps=1e-8):
        model._require_cuda()
        self.model, self.lr, self.alpha, self.eps = model, lr, alpha, eps
        self._lo, self._hi = param_range
        self._params = list(params)
        self.square_avg = torch.zeros(self._hi - self._lo, device=model._flat.device, dtype=torch.float32)
        self.param_groups = [dict(params=self._params, lr=lr, alpha=alpha, eps=eps)]

    def zero_grad(self, set_to_none=True):
        for p in self._params:
            p.grad = None

    def step(self):
        m = self.model
        if any(p.grad is None for p in self._params):
            return                                   # torch skips parameters without a gradient
        L.check(L.load().carel_rmsprop_step(m._flat.data_ptr() + 4 * self._lo, m._flat_grad.data_ptr() + 4 * self._lo, self.square_avg.data_ptr(),
                                            self._hi - self._lo, self.param_groups[0]["lr"], self.alpha, self.eps, L.current_stream()),
                "carel_rmsprop_step")

    def state_dict(self):
        return dict(square_avg=self.square_avg, lr=self.lr)

    def load_state_dict(self, sd):
        self.square_avg.copy_(sd["square_avg"])


class DrlClassifier(_Base):
    """Reference `DrlClassifier` of drl_classifier_en.py (:140-624)."""

    TERM_NAMES = TERM_NAMES

    def __init__(self, opt, encoder_cfg=None, seed=None):
        if encoder_cfg is None:
            encoder_cfg = encoder_config("en")
        super().__init__(opt, encoder_cfg=encoder_cfg, seed=seed)
        self._has_pair_skip = False          # this script has no "pair loss replaced by 0" branch (:587-603)

    def _build_heads(self, opt):
        D, Cd, V, E = opt.ec_dim, opt.con_dim, opt.pair_bow_dim, opt.ec_num_class
        if E != 1 or opt.pair_num_class != 1:
            raise L.CarelError("ec_num_class and pair_num_class must be 1 (one-logit heads, drl_classifier_en.py:31-32)")
        self.content_mu, self.content_log_var = _Holder((Cd, H)), _Holder((Cd, H))
        self.emotion_mu, self.emotion_log_var = _Holder((D, H)), _Holder((D, H))
        self.cause_mu, self.cause_log_var = _Holder((D, H)), _Holder((D, H))
        self.emotion_disc, self.content_disc, self.cause_disc = _Holder((E, Cd)), _Holder((V, D)), _Holder((E, Cd))
        self.ec_disc, self.ce_disc = _Holder((E, D)), _Holder((E, D))
        self.content_classifier = _Holder((V, Cd))
        self.emotion_classifier, self.cause_classifier = _Holder((E, D)), _Holder((E, D))
        self.pair_classifier = _Holder((opt.pair_num_class, 2 * D))
        self.decoder = _Holder((V, 2 * D + Cd))
        self.dropout = nn.Dropout(opt.dropout)

    # ------------------------------------------------------------------ flat parameter storage
    def _param_order(self):
        """[vae_and_cls group | content_disc | emotion_disc | cause_disc | ec_disc | ce_disc | six latent heads]: every
        optimiser group of get_params() is one contiguous range; the latent heads are in no group (:357-376)."""
        order, _, named = _Base._param_order_encoder(self)
        order += ["encoder.pooler.dense.weight", "encoder.pooler.dense.bias"]
        for h in OTHER_HEADS:
            order += [h + ".weight", h + ".bias"]
        n_opt_names = len(order)
        for g in DISC_GROUPS:
            order += [g + ".weight", g + ".bias"]
        for h in LATENT_HEADS:
            order += [h + ".weight", h + ".bias"]
        self._pair_range_names = ["pair_classifier.weight", "pair_classifier.bias"]
        assert set(order) == set(named), "parameter inventory mismatch"
        return order, n_opt_names, named

    def _flatten(self):
        super()._flatten()
        offs, named = self._offs, self._named
        end = lambda k: offs[k] + ((named[k].numel() + 63) & ~63)        # noqa: E731
        self._group_ranges = {"vae": (0, self._n_opt)}
        for g in DISC_GROUPS:
            self._group_ranges[g] = (offs[g + ".weight"], end(g + ".bias"))
        self._disc_lo, self._disc_hi = self._group_ranges[DISC_GROUPS[0]][0], self._group_ranges[DISC_GROUPS[-1]][1]
        dev = self._flat.device
        # gradient images of the discriminator range: [0] own loss (content_disc: the emotion-sample loss), [1] content_disc's
        # cause-sample loss, [2] the vae loss's entropy terms
        self._disc_img = [torch.zeros(self._disc_hi - self._disc_lo, device=dev) for _ in range(3)] if dev.type == "cuda" else None

    def get_params(self):
        """(content_disc, emotion_disc, cause_disc, ec_disc, ce_disc, other) parameter lists, reference order (:357-376)."""
        groups = [list(getattr(self, g).parameters()) for g in DISC_GROUPS]
        other = list(self.encoder.parameters())
        for h in OTHER_HEADS:
            other += list(getattr(self, h).parameters())
        return tuple(groups) + (other,)

    def make_fused_optimizers(self, adv_lr=None, vae_lr=None, fuse_into_backward=False):
        """The six optimisers of the script body (:1056-1062) as HIP kernels over the flat buffer, in get_params() order (the
        list the reference's train() unpacks, :884): RMSprop(adv_lr) for the five discriminators, Adam(vae_lr) for the rest."""
        adv_lr = self.opt.adv_lr if adv_lr is None else adv_lr
        vae_lr = self.opt.vae_lr if vae_lr is None else vae_lr
        groups = self.get_params()
        opts = [FusedRMSprop(self, lr=adv_lr, param_range=self._group_ranges[g], params=groups[i]) for i, g in enumerate(DISC_GROUPS)]
        opts.append(FusedAdam(self, lr=vae_lr, param_range=self._group_ranges["vae"], params=groups[5], fuse_into_backward=fuse_into_backward))
        return tuple(opts)

    # ------------------------------------------------------------------ forward / backward bodies
    def set_noise(self, eps_con, eps_e=None, eps_c=None):
        """Test hook: the next call's three noise vectors ([con_dim], [ec_dim], [ec_dim]; reference draw order :238-240)."""
        self._noise = None if eps_con is None else (eps_con, eps_e, eps_c)

    def _draw_noise_en(self, dev):
        if self._noise is not None:
            con, e, c = (t.to(dev, torch.float32).reshape(-1) for t in self._noise)
            self._noise = None
        elif self._dp is not None:                                  # same draws on every rank, no collective
            con, e, c = self._dp.draw_noise_sizes((self.opt.con_dim, self.opt.ec_dim, self.opt.ec_dim), dev)
        else:
            con = torch.randn(self.opt.con_dim, device=dev)        # content, emotion, cause (:238-240)
            e = torch.randn(self.opt.ec_dim, device=dev)
            c = torch.randn(self.opt.ec_dim, device=dev)
        return torch.cat((e, c, con)).contiguous()                  # layout of z / generative_emb (:243)

    def _make_call(self, input_ids, att_masks, token_type_ids, emotion_labels, cause_labels, pair_labels, content_bow, iteration, training,
                   seq_lengths=None):
        self._require_cuda()
        ops._chk_cuda(input_ids, att_masks, token_type_ids, content_bow)
        B, S = input_ids.shape
        Bp = self._padded_batch(B, S)
        c = _Call()
        c.B, c.S, c.Bp = B, S, Bp
        c.ids, c.att = self._prep_ids(input_ids, Bp), self._prep_ids(att_masks, Bp)
        c.tt = None if token_type_ids is None else self._prep_ids(token_type_ids, Bp)
        dev, f32 = input_ids.device, torch.float32
        c.labels = dict(emo=emotion_labels.to(dev, f32).reshape(-1).contiguous(), cau=cause_labels.to(dev, f32).reshape(-1).contiguous(),
                        pair=pair_labels.to(dev, f32).reshape(-1).contiguous(), bow=content_bow.to(dev, f32).contiguous())
        if c.labels["bow"].shape != (B, self.opt.pair_bow_dim):
            raise L.CarelError("content_bow must be [batch, pair_bow_dim]")
        c.iteration, c.training = int(iteration), training
        c.eps = self._draw_noise_en(dev)
        self._fwd_count += 1
        c.seed = (self.dropout_base_seed * 1000003 + self._fwd_count) & 0xFFFFFFFF
        c.row_offset = 0 if self._dp is None else self._dp.row_offset(B)
        c.label_sum = None
        if self._dp is not None:            # pos_weight (:599) of the GLOBAL batch: one float summed over ranks
            c.label_sum = self._dp.all_reduce_sum(c.labels["pair"].sum().reshape(1))
        c.pack = self._pack_info(c.att, B, Bp, S, seq_lengths)
        key = ("en_tail", B, S)
        buf = self._ws.get(key)
        if buf is None:
            o = self.opt
            lib = L.load()
            buf = SimpleNamespace(
                pooled=torch.empty(B, H, device=dev), lat=torch.empty(B, 2 * o.con_dim + 4 * o.ec_dim, device=dev),
                z=torch.empty(B, 2 * o.ec_dim + o.con_dim, device=dev), terms=torch.zeros(32, device=dev),
                work=torch.empty(lib.carel_en_tail_workspace_floats(B, o.ec_dim, o.con_dim, o.pair_bow_dim), device=dev),
                dx_last=torch.empty(Bp * S, H, device=dev))
            self._ws[key] = buf
        c.buf = buf
        return c

    def _kl_weights(self, iteration):
        o = self.opt
        if iteration < o.kl_ann_iterations:                          # :289-303
            return self.get_annealed_weight(iteration, o.ec_kl_lambda), self.get_annealed_weight(iteration, o.con_kl_lambda)
        return 1.0, 1.0

    def _en_tail_args(self, c, x_last_ptr, cls_rows, n_rows, train_drop):
        o, a = self.opt, L.EnTailArgs()
        a.batch, a.seq_len, a.hidden, a.ec_dim, a.con_dim, a.bow_dim = c.B, c.S, H, o.ec_dim, o.con_dim, o.pair_bow_dim
        a.x_last_f32, a.cls_rows, a.n_rows = x_last_ptr, (None if cls_rows is None else cls_rows.data_ptr()), n_rows
        w, g = self._w, self._g
        a.pooler_w, a.pooler_b = w("encoder.pooler.dense.weight"), w("encoder.pooler.dense.bias")
        for i, h in enumerate(LATENT_HEADS):
            a.head_w[i], a.head_b[i] = w(h + ".weight"), w(h + ".bias")
        a.cdisc_w, a.cdisc_b = w("content_disc.weight"), w("content_disc.bias")
        for i, h in enumerate(SMALL_DISCS):
            a.sdisc_w[i], a.sdisc_b[i] = w(h + ".weight"), w(h + ".bias")
        a.ccls_w, a.ccls_b = w("content_classifier.weight"), w("content_classifier.bias")
        a.emo_w, a.emo_b = w("emotion_classifier.weight"), w("emotion_classifier.bias")
        a.cau_w, a.cau_b = w("cause_classifier.weight"), w("cause_classifier.bias")
        a.pair_w, a.pair_b = w("pair_classifier.weight"), w("pair_classifier.bias")
        a.dec_w, a.dec_b = w("decoder.weight"), w("decoder.bias")
        if c.labels is not None:
            lb = c.labels
            a.emo_labels, a.cau_labels, a.pair_labels, a.bow = lb["emo"].data_ptr(), lb["cau"].data_ptr(), lb["pair"].data_ptr(), lb["bow"].data_ptr()
            a.eps = c.eps.data_ptr()
            a.kl_w_ec, a.kl_w_con = self._kl_weights(c.iteration)
        a.w_con_adv, a.w_ec_adv, a.w_ecce_adv = o.con_adv_loss_weight, o.ec_adv_loss_weight, o.ecce_adv_loss_weight
        a.w_ec_mul, a.w_con_mul, a.w_pair = o.ec_mul_loss_weight, o.con_mul_loss_weight, o.pair_mul_loss_weight
        a.label_smoothing, a.epsilon = o.label_smoothing, o.epsilon
        a.drop_p, a.drop_seed = (o.dropout if train_drop else 0.0), c.seed
        a.drop_row_offset = getattr(c, "row_offset", 0)
        if getattr(c, "label_sum", None) is not None:
            a.global_label_sum, a.global_n = c.label_sum.data_ptr(), self._dp.world * c.B
        b = c.buf
        a.pooled, a.lat, a.z, a.terms, a.work = b.pooled.data_ptr(), b.lat.data_ptr(), b.z.data_ptr(), b.terms.data_ptr(), b.work.data_ptr()
        img = lambda i, k: self._disc_img[i].data_ptr() + 4 * (self._offs[k] - self._disc_lo)      # noqa: E731
        for i in range(3):
            a.g_cdisc_w[i], a.g_cdisc_b[i] = img(i, "content_disc.weight"), img(i, "content_disc.bias")
        for i, h in enumerate(SMALL_DISCS):
            a.g_sdisc_w[i], a.g_sdisc_b[i] = img(0, h + ".weight"), img(0, h + ".bias")
            a.g_sdisc_ent_w[i], a.g_sdisc_ent_b[i] = img(2, h + ".weight"), img(2, h + ".bias")
        a.d_ccls_w, a.d_ccls_b = g("content_classifier.weight"), g("content_classifier.bias")
        a.d_emo_w, a.d_emo_b = g("emotion_classifier.weight"), g("emotion_classifier.bias")
        a.d_cau_w, a.d_cau_b = g("cause_classifier.weight"), g("cause_classifier.bias")
        a.d_pair_w, a.d_pair_b = g("pair_classifier.weight"), g("pair_classifier.bias")
        a.d_dec_w, a.d_dec_b = g("decoder.weight"), g("decoder.bias")
        a.d_pooler_w, a.d_pooler_b = g("encoder.pooler.dense.weight"), g("encoder.pooler.dense.bias")
        a.dx_last_f32 = b.dx_last.data_ptr()
        return a

    def _run_forward(self, c, training):
        if self._adam_hook is not None:
            self._adam_hook._join()
        self._refresh_shadow()
        train_drop = self.training
        ws = self._workspace(c.Bp, c.S, inference=not training)
        c.cls = self._cls_info(c.B, c.Bp, c.S, c.pack, self._flat.device)
        ea = self._encoder_args(c.ids, c.att, c.tt, ws, c.Bp, c.S, not training, train_drop, c.seed, c.row_offset, c.pack, c.cls)
        lib, st = L.load(), L.current_stream()
        L.check(lib.carel_encoder_forward(C.byref(ea), st), "carel_encoder_forward")
        x_last_ptr = lib.carel_encoder_x_last(C.byref(ea))
        if c.cls is not None:
            cls_rows, n_rows = c.cls.compact, c.cls.n_cls
        else:
            cls_rows, n_rows = (None, c.Bp * c.S) if c.pack is None else (c.pack.cu, c.pack.n_tokens)
        ta = self._en_tail_args(c, x_last_ptr, cls_rows, n_rows, train_drop)
        L.check(lib.carel_en_tail_latents(C.byref(ta), st), "carel_en_tail_latents")
        L.check(lib.carel_en_tail_losses(C.byref(ta), st), "carel_en_tail_losses")
        c.ea, c.ta, c.ws = ea, ta, ws
        c.keep = (cls_rows,)

    def _group_has_grad(self, g):
        return self._named[(g if g != "vae" else "encoder.embeddings.word_embeddings") + ".weight"].grad is not None

    def _bind_group(self, g):
        if self._grad_views is None:
            self._grad_views = {k: self._grad_view(k) for k in self._order}
        lo, hi = self._group_ranges[g]
        for k in self._order:
            if lo <= self._offs[k] < hi:
                self._named[k].grad = self._grad_views[k]

    def _run_backward_en(self, c, grads):
        lib, st = L.load(), L.current_stream()
        f32 = torch.float32

        def share(group, image, g, acc):
            lo, hi = self._group_ranges[group]
            n = hi - lo
            dst = self._flat_grad.data_ptr() + 4 * lo
            src = self._disc_img[image].data_ptr() + 4 * (lo - self._disc_lo)
            gd = g.to(f32).reshape(1).contiguous()
            L.check(lib.carel_axpy_f32(dst, src, n, gd.data_ptr(), int(acc), st), "carel_axpy_f32")

        touched = {}
        for i in range(6):
            if grads[i] is None:
                continue
            group, image = _LOSS_TO_GROUP[i]
            acc = touched.get(group, self._group_has_grad(group))
            share(group, image, grads[i], acc)
            touched[group] = True
        if grads[6] is not None:
            go = grads[6].to(f32).reshape(1).contiguous()
            for group in DISC_GROUPS:                  # entropy terms of the vae loss (they reach the discriminators only)
                acc = touched.get(group, self._group_has_grad(group))
                share(group, 2, go, acc)
                touched[group] = True
            lo, hi = self._group_ranges["vae"]
            accumulate = self._group_has_grad("vae")
            prev = self._flat_grad[lo:hi].clone() if accumulate else None
            ea = c.ea
            ea.dx = c.buf.dx_last.data_ptr()
            L.check(lib.carel_en_tail_backward(C.byref(c.ta), go.data_ptr(), st), "carel_en_tail_backward")
            h_lo = self._offs[OTHER_HEADS[0] + ".weight"]          # classifier / decoder gradients were produced for grad_output = 1
            ops.scale_(self._flat_grad[h_lo:hi], go)
            if self._dp is not None:          # pooler, heads and (now complete) discriminator gradients: one bucket.  Under data
                self._dp.tail_done()          # parallelism the discriminator backward calls must precede this one (the reference's order)
            self._backward_encoder(ea, accumulate)
            if self._dp is not None:
                self._dp.backward_done()
            if accumulate:
                self._flat_grad[lo:hi].add_(prev)
            touched["vae"] = True
        for g in touched:
            self._bind_group(g)

    # ------------------------------------------------------------------ public API (reference surface)
    def forward(self, input_ids, att_masks, token_type_ids, emotion_labels, cause_labels, pair_labels, content_bow, iteration,
                seq_lengths=None):
        """Reference `forward` (:205-334): (content_disc_loss_emo, content_disc_loss_cau, emotion_disc_loss, ec_disc_loss,
        cause_disc_loss, ce_disc_loss, vae_and_classifier_loss)."""
        c = self._make_call(input_ids, att_masks, token_type_ids, emotion_labels, cause_labels, pair_labels, content_bow, iteration,
                            training=torch.is_grad_enabled(), seq_lengths=seq_lengths)
        self._last_call = c
        if torch.is_grad_enabled():
            anchor = self._flat.new_zeros((), requires_grad=True)
            return _EnLosses.apply(anchor, self, c)
        self._run_forward(c, training=False)
        return tuple(c.buf.terms[i].clone() for i in range(7))

    def forward_terms(self, *args, **kw):
        """Added introspection entry point: every loss term plus the latent means / log-variances (no autograd)."""
        with torch.no_grad():
            self.forward(*args, **kw)
        c = self._last_call
        t = c.buf.terms.clone()
        o = self.opt
        D, Cd = o.ec_dim, o.con_dim
        lat = c.buf.lat.clone()
        out = {n: t[i] for i, n in enumerate(TERM_NAMES)}
        out.update(mu_con=lat[:, :Cd], lv_con=lat[:, Cd:2 * Cd], mu_e=lat[:, 2 * Cd:2 * Cd + D], lv_e=lat[:, 2 * Cd + D:2 * Cd + 2 * D],
                   mu_c=lat[:, 2 * Cd + 2 * D:2 * Cd + 3 * D], lv_c=lat[:, 2 * Cd + 3 * D:], pooled=c.buf.pooled.clone(), z=c.buf.z.clone())
        return out

    def last_terms(self):
        return {n: self._last_call.buf.terms[i] for i, n in enumerate(TERM_NAMES)}

    def sampled_embeddings(self):
        raise L.CarelError("sampled_embeddings() belongs to the two-space model")

    def pair_logits(self, input_ids, att_masks, token_type_ids, chunk=1024):
        """pair_classifier([z_e, z_c]) with fresh emotion / cause noise (:336-353), chunked over the batch."""
        self._require_cuda()
        ops._chk_cuda(input_ids, att_masks, token_type_ids)
        dev, o = input_ids.device, self.opt
        if self._noise is not None:
            _, eps_e, eps_c = (t.to(dev, torch.float32).reshape(-1).contiguous() for t in self._noise)
            self._noise = None
        else:
            eps_e, eps_c = torch.randn(o.ec_dim, device=dev), torch.randn(o.ec_dim, device=dev)     # emotion first (:347-348)
        N, S = input_ids.shape
        out = torch.empty(N, device=dev, dtype=torch.float32)
        self._refresh_shadow()
        lib = L.load()
        LW = 2 * o.con_dim + 4 * o.ec_dim
        for s in range(0, N, chunk):
            ids = input_ids[s:s + chunk]
            B = ids.shape[0]
            Bp = self._padded_batch(B, S)
            c = _Call()
            c.B, c.S, c.Bp, c.labels, c.seed = B, S, Bp, None, 0
            ids, att = self._prep_ids(ids, Bp), self._prep_ids(att_masks[s:s + chunk], Bp)
            tt = None if token_type_ids is None else self._prep_ids(token_type_ids[s:s + chunk], Bp)
            ws = self._workspace(Bp, S, inference=True)
            pack = self._pack_info(att, B, Bp, S)
            cls = self._cls_info(B, Bp, S, pack, dev)
            ea = self._encoder_args(ids, att, tt, ws, Bp, S, True, False, 0, 0, pack, cls)
            L.check(lib.carel_encoder_forward(C.byref(ea), L.current_stream()), "carel_encoder_forward")
            key = ("en_lat", B)
            buf = self._ws.get(key)
            if buf is None:       # latents only: the loss-side buffers of carel_en_tail_args stay NULL
                null = SimpleNamespace(data_ptr=lambda: None)
                buf = SimpleNamespace(pooled=torch.empty(B, H, device=dev), lat=torch.empty(B, LW, device=dev), z=null, terms=null, work=null,
                                      dx_last=null)
                self._ws[key] = buf
            c.buf = buf
            cls_rows = cls.compact if cls is not None else (None if pack is None else pack.cu)
            ta = self._en_tail_args(c, lib.carel_encoder_x_last(C.byref(ea)), cls_rows, 0, False)
            L.check(lib.carel_en_tail_latents(C.byref(ta), L.current_stream()), "carel_en_tail_latents")
            L.check(lib.carel_en_pair_logits(buf.lat.data_ptr(), LW, 2 * o.con_dim, 2 * o.con_dim + 2 * o.ec_dim, eps_e.data_ptr(), eps_c.data_ptr(),
                                             self._w("pair_classifier.weight"), self._w("pair_classifier.bias"), B, o.ec_dim,
                                             out[s:s + B].data_ptr(), L.current_stream()), "carel_en_pair_logits")
        return out

    def pair_probabilities(self, input_ids, att_masks, token_type_ids, chunk=1024):
        return torch.sigmoid(self.pair_logits(input_ids, att_masks, token_type_ids, chunk))

    def get_pair_preds(self, input_ids, att_masks, token_type_ids):
        """Reference `get_pair_preds` (:336-353): the raw logits, a [N, 1] tensor."""
        return self.pair_logits(input_ids, att_masks, token_type_ids).reshape(-1, 1)

    def get_annealed_weight(self, iteration, lambda_weight):
        return (math.tanh((iteration - self.opt.kl_ann_iterations * 1.5) / (self.opt.kl_ann_iterations / 3)) + 1) * lambda_weight

    # two-space-only entry points of the base class
    def get_ec_aprx_loss(self, *a, **k):
        raise L.CarelError("the approximation network belongs to drl_classifier_ec_vi")

    get_ec_upper_loss = get_ec_aprx_loss
