"""The reference's training driver around the hot path: `train`, `generate_self_train_data`, `save_ckp`,
`load_ckp` (drl_classifier_ec_mmd_final_mul.py :603-628, :734-799, :802-922), same signatures and control
flow; `opt` is an explicit keyword instead of a module global.  With two optimisers it runs the VI ablation's
two-phase step (drl_classifier_ec_vi.py :723-790); with six, the adversarial step of drl_classifier_en.py (:904-947: five
discriminator backward calls, the vae backward, six optimiser steps; evaluation on sigmoid(logits), :975-976).  The step body (:823-845) is unchanged: the
model/optimiser objects it calls are carel_vae_amd.DrlClassifier and FusedAdam (or any torch optimiser).
"""
import os
from random import randint

import pandas as pd
import torch


def _prf1(labels, preds):
    """binary precision / recall / F1 with sklearn's zero-division convention (0.0), ref :868-870."""
    y = [int(round(v[0] if isinstance(v, (list, tuple)) else v)) for v in labels]
    p = [int(round(v[0] if isinstance(v, (list, tuple)) else v)) for v in preds]
    tp = sum(1 for a, b in zip(y, p) if a == 1 and b == 1)
    fp = sum(1 for a, b in zip(y, p) if a == 0 and b == 1)
    fn = sum(1 for a, b in zip(y, p) if a == 1 and b == 0)
    prec = tp / (tp + fp) if tp + fp else 0.0
    rec = tp / (tp + fn) if tp + fn else 0.0
    f1 = 2 * prec * rec / (prec + rec) if prec + rec else 0.0
    return prec, rec, f1


class RunningLoss:
    """The reference's `running_loss += loss.item()` ... print every 10 iterations (:845-851) without ever draining the GPU queue.
    The sum is kept on the device; every `every`-th step it is copied into a page-locked slot behind an event, and the line is
    handed to `log` as soon as that event has completed (checked with a query when the next step is enqueued, waited for only by
    flush(wait=True) at the end of the epoch's training loop) -- same numbers, same order, printed a few steps later.  A
    `float(loss)` on the host instead makes the host wait for everything enqueued so far, and the GPU then idles until the host has
    run ahead again (bench.py measured that step 26 % slower than the same step without the read-back on one box)."""

    def __init__(self, device, every=10, log=print, fmt="[%d, %5d] training loss: %.4f", slots=64):
        self.every, self.log, self.fmt = int(every), log, fmt
        self._pin = torch.zeros(slots, dtype=torch.float32).pin_memory() if torch.device(device).type == "cuda" else torch.zeros(slots)
        self._cuda = torch.device(device).type == "cuda"
        self._acc, self._pending, self._n, self.values = None, [], 0, []

    def add(self, step_loss, epoch, iteration):
        self._acc = step_loss if self._acc is None else self._acc + step_loss
        if iteration % self.every == self.every - 1:
            slot = self._n % self._pin.numel()
            if len(self._pending) >= self._pin.numel():
                self.flush(wait=True)
            self._n += 1
            v = self._acc.reshape(1).to(torch.float32)
            ev = None
            if self._cuda:
                self._pin[slot:slot + 1].copy_(v, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            else:
                self._pin[slot] = float(v)
            self._pending.append((ev, slot, epoch, iteration + 1))
            self._acc = None
        self.flush()

    def flush(self, wait=False):
        while self._pending:
            ev, slot, epoch, it = self._pending[0]
            if ev is not None:
                if wait:
                    ev.synchronize()
                elif not ev.query():
                    return
            self._pending.pop(0)
            val = float(self._pin[slot]) / self.every
            self.values.append(val)
            self.log(self.fmt % (epoch, it, val))


def load_ckp(checkpoint_path, model):
    """ref :603-613.  Loads tensors only (no pickled code is executed)."""
    checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    model.load_state_dict(checkpoint)
    return model


def save_ckp(state, ckpt_path, model_id="carel"):
    """ref :616-628: <ckpt_path>/<model_id>.pt holding the state_dict (reference key names)."""
    if not os.path.exists(ckpt_path):
        os.makedirs(ckpt_path)
    torch.save({k: v.detach().cpu().clone() for k, v in state.items()}, os.path.join(ckpt_path, model_id + ".pt"))


def generate_self_train_data(test_docs_pair_size, test_df, test_loader, model, strategy, device="cuda"):
    """ref :734-799: pseudo-label the test documents (top pair positive, random / extreme / thresholded negative)."""
    predicted_df = test_df.copy()
    model.eval()
    with torch.no_grad():
        for data in test_loader:
            ids = data["input_ids"].to(device, dtype=torch.long)
            att = data["attention_masks"].to(device, dtype=torch.long)
            tt = data["token_type_ids"].to(device, dtype=torch.long)
            outs = model.get_pair_preds(ids, att, tt)
            if torch.is_tensor(outs):        # drl_classifier_en.py:826-828: raw logits -> probabilities
                outs = torch.sigmoid(outs).cpu().detach().numpy().tolist()
            predicted_df["label"] = [x[0] for x in outs]
    rows, curr = [], 0
    for doc_pair_size in test_docs_pair_size:
        max_pos, max_neg = float("-inf"), float("-inf")
        pos_pair = pos_emotion = neg_pair = neg_emotion = None
        prob_dict = {}
        for i in range(doc_pair_size):
            index = i + curr
            row = predicted_df.iloc[index]
            prob = row["label"]
            if strategy == "threshold":
                if prob > 0.5 and prob > max_pos:
                    pos_pair, max_pos = row["pair"], prob
                elif 0.5 >= prob > max_neg:
                    neg_pair, max_neg = row["pair"], prob
            elif strategy in ("random", "extreme"):
                prob_dict[index] = prob
                srt = sorted(prob_dict.items(), key=lambda x: x[1], reverse=True)
                top = predicted_df.iloc[srt[0][0]]
                pos_pair = top["pair"]
                if strategy == "random":
                    pos_emotion = top["emotion"]
                    if len(srt) == 1:
                        continue
                    other = predicted_df.iloc[srt[randint(1, len(srt) - 1)][0]]
                    neg_pair, neg_emotion = other["pair"], other["emotion"]
                else:               # "extreme" carries no emotion column values (ref :789-793 leaves both None), and a one-pair
                    neg_pair = predicted_df.iloc[srt[-1][0]]["pair"]     # document yields that pair as positive AND negative
        curr += doc_pair_size
        if pos_pair is not None and neg_pair is not None:
            rows.append((pos_pair, 1, pos_emotion))
            rows.append((neg_pair, 0, neg_emotion))
    return pd.DataFrame(rows, columns=["pair", "label", "emotion"])


def train(train_loader, test_loader, model, optimizers, device, num_unpred_pairs, self_metrics=None, self_train=False,
          opt=None, log=print):
    """ref :802-922.  One epoch = every batch through forward / zero_grad / backward / step (:823-845), then one
    evaluation pass with `get_pair_preds`, checkpointing the best F1."""
    opt = opt if opt is not None else model.opt
    vi = len(optimizers) == 2       # VI ablation (drl_classifier_ec_vi.py:723-725): [ec_aprx_opt, vae_and_cls_opt]
    en = len(optimizers) == 6       # drl_classifier_en.py:884: five discriminator optimisers + vae_and_cls_opt
    ec_aprx_opt, vae_and_cls_opt = optimizers if vi else (None, optimizers[-1])
    max_p = max_r = max_f1 = 0.0
    self_p = self_r = self_f1 = 0.0
    if self_train:
        self_p, self_r, self_f1 = self_metrics
        epochs = opt.self_epochs
    else:
        epochs = opt.epochs
    for epoch in range(1, epochs + 1):
        running = RunningLoss(device, every=10, log=log)
        model.train()
        log("\n############ Epoch {}: Training Start ############\n".format(epoch))
        for iteration, batch in enumerate(train_loader):
            # attended lengths from the HOST copy of the mask: lets the model skip padding without reading the mask back
            # from the device (a sync that would stop the host from running ahead of the GPU)
            kw = {}
            if getattr(model, "varlen", False):
                if "seq_lengths" in batch:                               # carel_vae_amd.data.BatchLoader provides them
                    kw["seq_lengths"] = batch["seq_lengths"]
                elif not batch["attention_masks"].is_cuda:
                    kw["seq_lengths"] = batch["attention_masks"].sum(1).tolist()
            ids = batch["input_ids"].to(device, dtype=torch.long, non_blocking=True)
            att = batch["attention_masks"].to(device, dtype=torch.long, non_blocking=True)
            tt = batch["token_type_ids"].to(device, dtype=torch.long, non_blocking=True)
            labels = batch["labels"].to(device, dtype=torch.float, non_blocking=True)
            emo = batch["emo_labels"].to(device, dtype=torch.float if en else torch.long, non_blocking=True)
            cau = batch["cau_labels"].to(device, dtype=torch.float, non_blocking=True)
            bow = batch["bow_reps"].to(device, dtype=torch.float, non_blocking=True)
            if en:                  # drl_classifier_en.py:913-947
                content_disc_opt, emotion_disc_opt, cause_disc_opt, ec_disc_opt, ce_disc_opt, _ = optimizers
                losses = model(ids, att, tt, emo, cau, labels, bow, iteration, **kw)
                cd_emo, cd_cau, emotion_disc_loss, ec_disc_loss, cause_disc_loss, ce_disc_loss, loss = losses
                content_disc_opt.zero_grad()
                (cd_emo + cd_cau).backward(retain_graph=True)
                emotion_disc_opt.zero_grad()
                emotion_disc_loss.backward(retain_graph=True)
                ec_disc_opt.zero_grad()
                ec_disc_loss.backward(retain_graph=True)
                cause_disc_opt.zero_grad()
                cause_disc_loss.backward(retain_graph=True)
                ce_disc_opt.zero_grad()
                ce_disc_loss.backward(retain_graph=True)
                vae_and_cls_opt.zero_grad()
                loss.backward()
                for o in optimizers:
                    o.step()
                running.add(sum(l.detach() for l in losses), epoch, iteration)
                continue
            if vi:                  # two-phase step, drl_classifier_ec_vi.py:754-774
                e_embedding, c_embedding, ec_aprx_loss, loss = model(ids, att, tt, emo, cau, labels, bow, iteration, **kw)
                ec_aprx_opt.zero_grad()
                ec_aprx_loss.backward(retain_graph=True)
                ec_aprx_opt.step()
                Rj_loss = model.get_ec_upper_loss(e_embedding, c_embedding)
                beta = min(1, (epoch - 1) * 0.1)
                loss += beta * Rj_loss
            else:
                loss = model(ids, att, tt, emo, cau, labels, bow, iteration, **kw)
            vae_and_cls_opt.zero_grad()
            loss.backward()
            vae_and_cls_opt.step()
            # same numbers as the reference's `running_loss += loss.item()` (:845-851), but accumulated on the device and read back
            # through a page-locked slot + event: neither a per-step .item() nor the print every 10 steps stops the host (RunningLoss)
            running.add(loss.detach() + ec_aprx_loss.detach() if vi else loss.detach(), epoch, iteration)
        running.flush(wait=True)
        model.eval()
        with torch.no_grad():
            for batch in test_loader:
                ids = batch["input_ids"].to(device, dtype=torch.long)
                att = batch["attention_masks"].to(device, dtype=torch.long)
                tt = batch["token_type_ids"].to(device, dtype=torch.long)
                labels = batch["labels"].cpu().numpy().tolist()
                preds = model.get_pair_preds(ids, att, tt)
                if torch.is_tensor(preds):       # drl_classifier_en.py:975-976
                    preds = torch.sigmoid(preds).cpu().detach().numpy().round().tolist()
                labels += [[1]] * num_unpred_pairs            # unpredicted emotions count as misses (:864-865)
                preds += [[0]] * num_unpred_pairs
                p, r, f1 = _prf1(labels, preds)
                log("current test pair precision: {:.4f}, recall: {:.4f}, f1 socre: {:.4f}\n".format(p, r, f1))
                checkpoint = model.state_dict()
                if f1 > max_f1 and not self_train:
                    save_ckp(checkpoint, opt.best_model_path, opt.model_id)
                    max_p, max_r, max_f1 = p, r, f1
                elif f1 > self_f1 and self_train:
                    save_ckp(checkpoint, opt.best_model_path, opt.model_id)
                    self_p, self_r, self_f1 = p, r, f1
    best = os.path.join(opt.best_model_path, opt.model_id + ".pt")
    best_model = load_ckp(best, model) if os.path.exists(best) else model
    if not self_train:
        return best_model
    return best_model, self_p, self_r, self_f1
