#!/usr/bin/env python3
"""CAREL-VAE training-step benchmark on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path (drl_classifier_ec_mmd_final_mul.py:823-845) over one resident batch:
forward (BERT-base encoder -> VAE tail -> loss) + backward + Adam (+ RCCL gradient all-reduce when N > 1).
Workload = BASELINE.json configs[1]: zh ECPE-shaped batch, 64 clause pairs per GPU, S = 128, vocabulary
21 128, bag-of-words width 23 771, bf16 MFMA / fp32 accumulate, dropout active (model.train()).
Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

def log(msg):
    sys.stderr.write("[bench %.1fs] %s\n" % (time.time() - T_START, msg))
    sys.stderr.flush()


T_START = time.time()
PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PMC_TRAFFIC_CSV = "r04_pmc_hbm_traffic.csv"      # regenerated every round for the final code (tools/pmc_traffic.sh)
PMC_MFMA_CSV = "r04_pmc_mfma_util.csv"           # (tools/pmc_mfma.sh)
FLOP_PER_PAIR = 67.05e9       # fwd+bwd algorithmic FLOPs per clause pair


ATTN_FLOP_PER_PAIR = 3 * 12 * 50.33e6          # QK^T + PV of 12 layers, fwd + bwd, at S = 128; scales with (l / 128)^2
LINEAR_FLOP_PER_PAIR = FLOP_PER_PAIR - ATTN_FLOP_PER_PAIR      # everything else scales with l / 128


def effective_flops(lengths):
    """SURVEY 8(d), shape-B: algorithmic FLOPs of one step counted on ATTENDED tokens only, so that skipping padding cannot inflate a
    utilisation figure: linear layers scale with l / 128 per pair, attention with (l / 128)^2."""
    return sum(LINEAR_FLOP_PER_PAIR * (l / 128.0) + ATTN_FLOP_PER_PAIR * (l / 128.0) ** 2 for l in lengths)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="clause pairs per GPU")
    ap.add_argument("--shape", default="A", choices=["A", "B"], help="A dense (roofline headline), B ECPE-shaped lengths")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ecpe", action="store_true", help="skip the secondary ECPE-shaped leg")
    ap.add_argument("--no-varlen", action="store_true", help="run padded positions through the encoder like the reference does")
    ap.add_argument("--torch-adam", action="store_true", help="use torch.optim.Adam instead of the fused HIP Adam")
    ap.add_argument("--adam-in-backward", action="store_true",
                    help="N > 1 only (the default at N = 1): each layer's fused Adam update inside backward(), behind its bucket's all-reduce, on the "
                         "auxiliary stream; off by default under DataParallel (stream budget: carel_vae_amd/dp.py)")
    ap.add_argument("--adam-in-step", action="store_true",
                    help="run the fused Adam as one pass in optim.step() instead of layer by layer inside backward() on the weight-gradient "
                         "stream (FusedAdam(fuse_into_backward=True), the default here: tools/ab_adam_stream.sh measured -2 %% dense, -2.5 %% ECPE-shaped)")
    ap.add_argument("--forward-chains", action="store_true",
                    help="forward pass as two half-batch chains on two streams (results identical); measured ~1 % slower with the round-2 GEMMs")
    ap.add_argument("--no-adam-in-backward", action="store_true", help=argparse.SUPPRESS)      # (accepted; same as --adam-in-step)
    ap.add_argument("--no-forward-chains", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--gemm-variant", type=int, action="append", default=[],
                    help="A/B runs: tuning hook passed to carel_gemm_set_variant before the run (repeatable; switches the run to the EXPERIMENTS "
                         "library libcarel_hip_exp.so -- the product library has no hooks)")
    ap.add_argument("--wire-dtype", default="fp32", choices=["fp32", "bf16"],
                    help="N > 1: format of the gradient buckets on the links (bf16 halves the all-reduce's bytes and time; DataParallel(wire_dtype=...), "
                         "tests/test_gpu_dp2.py states its tolerance).  Default fp32: the single-GPU numerics")
    ap.add_argument("--no-overlap", action="store_true",
                    help="weight-gradient GEMMs on the main stream (serial kernels: the run to put under rocprofv3 --kernel-trace)")
    return ap.parse_args()


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg_kw, opt_kw, hip_loss_fn):
    """The CPU restatement of the same step (oracle/, kind "port": pure-PyTorch fp32, the op sequence of ref :823-845) on
    the host cores of this box, SURVEY 8(d)'s protocol bounded to ~40 s: B = 8 with set_detect_anomaly as the reference
    runs it (:837) and without, B = 64 (the GPU workload's batch) without; warm-ups, then timed steps, MEDIAN step time.
    `value` is the best of the legs (the most favourable to the CPU).  Also returns the ELBO agreement of the HIP path."""
    import statistics
    from oracle import carel_oracle as O
    # The GPU box shows a one-GPU job every CPU of a much larger host (sched_getaffinity: 256 on the boxes seen so far), but the pool's
    # rule for it is a 16-CPU share per GPU ("size worker pools to the box's CPU share: 16 for one GPU"): the baseline uses
    # min(affinity, 16) threads and records both numbers (CAREL_CPU_BASELINE_THREADS overrides the share).
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    share = int(os.environ.get("CAREL_CPU_BASELINE_THREADS", "16"))
    cores = max(1, min(affinity, share))
    torch.set_num_threads(cores)
    log("cpu_baseline: %d threads (affinity %d CPUs, os.cpu_count() = %s), %s" % (cores, affinity, os.cpu_count(), _cpu_model()))
    cfg, opt = O.EncoderConfig(**cfg_kw), O.Opt(**opt_kw)
    g = torch.Generator().manual_seed(3)
    eps_e, eps_c = torch.randn(opt.ec_dim, generator=g), torch.randn(opt.ec_dim, generator=g)
    legs, first, P0 = [], {}, None
    for (B, anomaly, warm, timed_n, budget) in ((8, True, 2, 5, 12.0), (8, False, 1, 5, 10.0), (64, False, 1, 3, 30.0)):
        P = O.init_params(cfg, opt, seed=0)
        batch = O.synthetic_batch(B, 128, cfg, opt.pair_bow_dim, seed=1, shape="A")
        if P0 is None:
            P0 = {k: v.clone() for k, v in P.items()}
        st = O.AdamState()
        ts, t_begin = [], time.time()
        for n in range(warm + timed_n):
            if n >= warm + 2 and time.time() - t_begin > budget:
                break
            t0 = time.time()
            with torch.autograd.set_detect_anomaly(anomaly):
                P, out, _ = O.train_step(P, batch, 3, cfg, opt, st, eps_e, eps_c)
            dt = time.time() - t0
            if n >= warm:
                ts.append(dt)
            if B not in first:                                   # step 0 of this batch size: the CPU fp32 loss terms from the initial weights
                first[B] = (batch, {k: float(v) for k, v in out.items() if v.numel() == 1})
        med = statistics.median(ts)
        legs.append({"batch": B, "set_detect_anomaly": anomaly, "warmup_steps": warm, "timed_steps": len(ts), "median_s_per_step": med,
                     "clause_pairs_per_s": B / med})
        log("cpu_baseline: B=%d anomaly=%s: median %.2f s/step over %d steps -> %.1f pairs/s" % (B, anomaly, med, len(ts), B / med))
        del P, st
    best = max(legs, key=lambda l: l["clause_pairs_per_s"])
    terms = ("mmd", "emo", "cau", "pair", "kl_e", "kl_c", "rec")
    weights = dict(mmd=30.0, emo=10.0, cau=10.0, pair=30.0, kl_e=1.0, kl_c=1.0, rec=1.0)

    def compare(B):
        batch, out0 = first[B]
        hip = hip_loss_fn(P0, batch, eps_e, eps_c, cfg, opt)
        hip32 = hip_loss_fn(P0, batch, eps_e, eps_c, cfg, opt, fp32=True)      # the library's fp32 debug forward (carel_encoder_forward_f32)
        scale = sum(abs(weights[k] * out0[k]) for k in terms)
        return {"batch": B, "loss_cpu_fp32": out0["loss"], "loss_hip_bf16": hip["loss"],
                "loss_rel_err": abs(hip["loss"] - out0["loss"]) / abs(out0["loss"]),
                "loss_hip_fp32_debug": hip32["loss"], "loss_rel_err_fp32_debug": abs(hip32["loss"] - out0["loss"]) / abs(out0["loss"]),
                "term_rel_err_fp32_debug": {k: abs(hip32[k] - out0[k]) / max(abs(out0[k]), 1e-12) for k in terms},
                "loss_err_over_term_scale": abs(hip["loss"] - out0["loss"]) / scale,
                "loss_share_of_term_scale": abs(out0["loss"]) / scale,
                "term_rel_err": {k: abs(hip[k] - out0[k]) / max(abs(out0[k]), 1e-12) for k in terms}}
    p64, p8 = compare(64), compare(8)
    parity = dict(p64)
    parity["note"] = ("HIP (bf16 MFMA, fp32 accumulate) vs the CPU fp32 path on the SAME weights, batch and noise, dropout off.  Headline = the bench "
                      "configuration (B=64, S=128, 12 layers, V=23771); north_star tolerance 1e-3 relative, asserted by "
                      "tests/test_gpu_model.py::test_bench_configuration_elbo_within_1e_3_of_cpu_fp32.  The 8-sample batch below has a total that is "
                      "a near-cancellation of its weighted terms (loss_share_of_term_scale): every term still agrees to ~3e-4.  *_fp32_debug: the same comparison "
                      "with the encoder in the library's fp32 debug mode (model.debug_fp32): what is left there is kernel error, not bf16 rounding.")
    parity["small_batch_B8"] = p8
    return {"value": best["clause_pairs_per_s"], "unit": "clause-pairs/s", "cores": cores, "affinity_cpus": affinity, "torch_threads": torch.get_num_threads(),
            "cores_note": "threads actually used = min(sched_getaffinity, the pool's 16-CPU share per GPU); the B = 64 leg is slower PER PAIR than the B = 8 "
                          "legs on these hosts: its ~6 GB of saved fp32 activations stream from DRAM through the memory-bound elementwise / softmax / "
                          "LayerNorm passes of eager PyTorch, the B = 8 leg's ~0.7 GB largely stay in the last-level cache; `value` is the best leg",
            "cpu_model": _cpu_model(), "kind": "port",
            "sample": "oracle train_step (fwd+bwd+Adam, fp32, 12 layers, S=128): best of the legs below (B=%d, set_detect_anomaly %s), median "
                      "step time" % (best["batch"], "on as ref :837" if best["set_detect_anomaly"] else "off"),
            "legs": legs}, parity


def english_leg(dev, batch_size, steps):
    """Config 4 (BASELINE.json configs[3]): one training step of the three-space adversarial model of drl_classifier_en.py --
    RoBERTa-base encoder (vocab 50 265), content / emotion / cause latents, five discriminators, the six backward calls and
    six Adam steps of its loop (:919-947).  V = 22 463 (data/ecpe_and_reccon_all_data_pair_en.txt)."""
    from carel_vae_amd import drl_classifier_en as ME
    from carel_vae_amd.data import synthetic_ecpe_batch
    V = 22463
    opt = ME.make_opt(pair_bow_dim=V)
    model = ME.DrlClassifier(opt, seed=0).to(dev)
    model.train()
    opts = model.make_fused_optimizers(fuse_into_backward=False)      # (per-layer Adam inside backward: 0.5 % slower here, tools/ab_en.py)
    out = {}
    for shape in ("A", "B"):
        bb, ll = [], []
        for i in range(4):
            b = synthetic_ecpe_batch(batch_size, 128, 50265, V, seed=301 + i, shape=shape, pad_id=1, first_id=2, binary_emotion=True)
            ll.append(b["attention_masks"].sum(1).tolist())
            bb.append({k: v.to(dev) for k, v in b.items()})

        def step(i):
            b = bb[i % 4]
            cd_e, cd_c, ed, ecd, cad, ced, vae = model(b["input_ids"], b["attention_masks"], b["token_type_ids"], b["emo_labels"],
                                                       b["cau_labels"], b["labels"], b["bow_reps"], i % 41, seq_lengths=ll[i % 4])
            opts[0].zero_grad(); (cd_e + cd_c).backward(retain_graph=True)        # noqa: E702   the reference's sequence (:919-939)
            opts[1].zero_grad(); ed.backward(retain_graph=True)                  # noqa: E702
            opts[3].zero_grad(); ecd.backward(retain_graph=True)                 # noqa: E702
            opts[2].zero_grad(); cad.backward(retain_graph=True)                 # noqa: E702
            opts[4].zero_grad(); ced.backward(retain_graph=True)                 # noqa: E702
            opts[5].zero_grad(); vae.backward()                                  # noqa: E702
            for o in opts:
                o.step()
            return vae
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            vae = step(3 + i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["dense" if shape == "A" else "ecpe_shaped"] = {"value": batch_size * steps / dt, "unit": "clause-pairs/s", "ms_per_step": 1e3 * dt / steps,
                                                           "steps": steps, "final_vae_loss": float(vae.detach())}
        log("english adversarial leg, shape %s: %.3f ms/step" % (shape, 1e3 * dt / steps))
    out["note"] = ("drl_classifier_en.py step: RoBERTa-base (vocab 50265), con_dim 384, bow V=22463, B=%d, dropout on, fused HIP Adam x6; "
                   "one GPU" % batch_size)
    del model, opts
    torch.cuda.empty_cache()
    return out


def ablation_leg(dev, batch_size, steps, cfg_vocab):
    """Config 5 (BASELINE.json configs[4]): the same training step with the HSIC head (drl_classifier_ec_hsic.py:253: HSIC added
    unweighted, one-logit BCE emotion head) and with the VI / CLUB head (drl_classifier_ec_vi.py:759-774: approximation-net step with
    its own Adam(lr 3e-3) between forward and the main backward, then the CLUB upper bound x beta) -- dense shape-A batches, the
    headline's encoder kernels.  The MMD head is the headline itself."""
    from carel_vae_amd import drl_classifier as M
    from carel_vae_amd.data import synthetic_ecpe_batch
    out = {}
    for name, kw in (("hsic", dict(disentangle="hsic", emotion_head="bce", e_num_class=1)), ("vi", dict(disentangle="vi", emotion_head="bce", e_num_class=1))):
        opt = M.make_opt(**kw)
        model = M.DrlClassifier(opt, M.encoder_config("zh"), seed=0).to(dev)
        model.train()
        bb, ll = [], []
        for i in range(4):
            b = synthetic_ecpe_batch(batch_size, 128, cfg_vocab, opt.pair_bow_dim, seed=401 + i, shape="A", binary_emotion=True)
            ll.append(b["attention_masks"].sum(1).tolist())
            bb.append({k: v.to(dev) for k, v in b.items()})
        if name == "vi":
            aprx_params, _ = model.get_params()
            aprx_opt = torch.optim.Adam(aprx_params, lr=opt.aprx_lr)
        main_opt = M.FusedAdam(model, lr=opt.vae_lr)

        def step(i):
            b = bb[i % 4]
            args = (b["input_ids"], b["attention_masks"], b["token_type_ids"], b["emo_labels"], b["cau_labels"], b["labels"], b["bow_reps"], i % 41)
            if name == "vi":                                    # the reference's two-phase step, line for line (:759-774)
                e_emb, c_emb, aprx_loss, loss = model(*args, seq_lengths=ll[i % 4])
                aprx_opt.zero_grad()
                aprx_loss.backward(retain_graph=True)
                aprx_opt.step()
                loss = loss + 0.5 * model.get_ec_upper_loss(e_emb, c_emb)
            else:
                loss = model(*args, seq_lengths=ll[i % 4])
            main_opt.zero_grad()
            loss.backward()
            main_opt.step()
            return loss
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            loss = step(3 + i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[name] = {"value": batch_size * steps / dt, "unit": "clause-pairs/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
                     "final_loss": float(loss.detach())}
        log("ablation leg %s: %.3f ms/step" % (name, 1e3 * dt / steps))
        del model, main_opt
        torch.cuda.empty_cache()
    out["note"] = ("config 5: dense shape-A step (B=%d, S=128, dropout on, fused HIP Adam) with opt.disentangle = hsic / vi (+ 1-logit BCE emotion head); "
                   "vi = two-phase step with torch.optim.Adam(lr 3e-3) on the approximation net and beta = 0.5; mmd = the headline" % batch_size)
    return out


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: start N rank processes (torch.distributed.run, one per GPU) as a CHILD of
    this process -- which never touches the GPU -- relay rank 0's JSON line, and fail unless exactly N ranks reported."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("no WORLD_SIZE in the environment: launching %d ranks: %s" % (a.gpus, " ".join(cmd)))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if r.returncode != 0 or line is None:
        sys.stderr.write(r.stdout)
        raise SystemExit("bench.py --gpus %d: the rank processes failed (rc %d)" % (a.gpus, r.returncode))
    n = json.loads(line).get("n_gpus")
    if n != a.gpus:
        raise SystemExit("bench.py --gpus %d: %s ranks reported" % (a.gpus, n))
    print(line, flush=True)


def input_pipeline_leg(dev, model, optim, batch_size):
    """SURVEY 8(d)'s full step -- host batch assembly + H2D copy + forward + backward + Adam + the reference's loss logging --
    through the reference's own driver loop (carel_vae_amd.training.train = ref :802-922): one epoch over a society_num-sized
    synthetic dataset (2 587 ECPE-shaped pairs, 41 steps) + the evaluation pass over 1 938 pairs, fed by PrefetchLoader."""
    from carel_vae_amd import data as D
    from carel_vae_amd import training as T
    V = model.opt.pair_bow_dim
    tr_ds, te_ds = D.SyntheticECPEDataset(2587, V, 1), D.SyntheticECPEDataset(1938, V, 2)
    saved = T.save_ckp
    T.save_ckp = lambda *a_, **k_: None            # the 400 MB checkpoint write on an F1 improvement is not what is measured
    opt = model.opt
    ep, opt.epochs = opt.epochs, 1
    out = {}
    try:
        for name in ("BatchLoader", "PrefetchLoader"):
            tr = D.BatchLoader(tr_ds, batch_size=batch_size, shuffle=True)
            if name == "PrefetchLoader":
                tr = D.PrefetchLoader(tr, dev, depth=3)
            te = D.BatchLoader(te_ds, batch_size=len(te_ds), shuffle=False)
            T.train(tr, te, model, [optim], dev, num_unpred_pairs=22, opt=opt, log=lambda *_: None)       # warm-up epoch
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            T.train(tr, te, model, [optim], dev, num_unpred_pairs=22, opt=opt, log=lambda *_: None)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out[name] = {"epoch_s": dt, "training_pairs_per_s": 2587 / dt}
            log("input pipeline leg, %s: %.3f s per epoch" % (name, dt))
    finally:
        T.save_ckp = saved
        opt.epochs = ep
    out["note"] = ("training.train (ref :802-922) for one epoch: 41 steps of %d ECPE-shaped pairs incl. batch assembly and H2D, running-loss read-back "
                   "every 10 steps, then get_pair_preds over 1938 test pairs; checkpoint write excluded" % batch_size)
    return out


def sentence_transformer_leg(dev, steps):
    """The fit() step (:84-87) of chi_ec_sentence_transformer.py (BERT-base, vocab 21128, margin 4.45) and of
    en_ec_sentence_transformer.py (all-mpnet-base-v2 = MPNet-base, vocab 30527, relative-position attention bias, L2-normalised
    embeddings; margin 5): 16 sentences, mean pooling, batch-semi-hard triplet loss, gradient-norm clip + AdamW; S = 128 dense and
    ECPE-like sentence lengths."""
    from carel_vae_amd import drl_classifier as M
    from carel_vae_amd import sentence_transformer as S
    from carel_vae_amd.data import synthetic_ecpe_batch
    out = {}
    for lang, cfg_name, vocab, margin in (("zh", "zh", 21128, 4.45), ("en_mpnet", "mpnet", 30527, 5.0)):
        model = S.SentenceTransformer(M.encoder_config(cfg_name), seed=0).to(dev)
        model.train(True)
        loss_mod = S.losses.BatchSemiHardTripletLoss(model=model, margin=margin)
        optim = S.FusedAdamW(model, lr=2e-5)
        res = {}
        for shape in ("A", "B"):
            feats = []
            for i in range(4):
                b = synthetic_ecpe_batch(16, 128, vocab, 8, seed=500 + i, shape=shape)
                ids = b["input_ids"]
                if cfg_name == "mpnet":                    # <pad> = 1 for MPNet's position ids (the generator pads with 0)
                    ids = torch.where(b["attention_masks"] == 0, torch.ones_like(ids), ids.clamp(min=3))
                feats.append(({"input_ids": ids.to(dev), "attention_mask": b["attention_masks"].to(dev), "token_type_ids": b["token_type_ids"].to(dev),
                               "seq_lengths": b["attention_masks"].sum(1).tolist()}, (b["emo_labels"].view(-1) % 4).to(dev)))

            def step(i):
                f, lab = feats[i % 4]
                loss = loss_mod([f], lab)
                loss.backward()
                optim.step()
                optim.zero_grad()
                return loss
            for i in range(3):
                step(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                loss = step(i)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res["dense" if shape == "A" else "ecpe_shaped"] = {"ms_per_step": 1e3 * dt / steps, "sentences_per_s": 16 * steps / dt, "final_loss": float(loss.detach())}
            log("sentence-transformer leg %s, shape %s: %.3f ms/step" % (lang, shape, 1e3 * dt / steps))
        if lang == "zh":
            out.update(res)
        else:
            out[lang] = res
        del model, optim, loss_mod
        torch.cuda.empty_cache()
    out["note"] = ("fit() step: B=16, S=128, dropout on, clip 1.0 + AdamW, one GPU.  Top level: chi_ec_sentence_transformer.py (BERT-base vocab 21128, "
                   "margin 4.45); en_mpnet: en_ec_sentence_transformer.py (MPNet-base vocab 30527 + relative-position bias + Normalize, margin 5).  "
                   "sentence_transformers parity unpinned; the MPNet encoder restatement is pinned to transformers.MPNetModel (see DESIGN.md)")
    return out


def main():
    a = parse()
    a.adam_in_step = a.adam_in_step or a.no_adam_in_backward
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(a)
    # Libraries chat on stdout (RCCL prints a five-line version banner at the first collective, gloo its peer counts): the
    # contract is ONE JSON line on stdout, so file descriptor 1 points at stderr until that line is printed.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if os.environ.get("CAREL_REHEARSE_ONE_GPU") == "1":      # rehearsal of the N > 1 code path on a one-GPU box: every rank on
        local_rank = 0                                       # cuda:0, gloo carrying the collectives (RCCL refuses that)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from carel_vae_amd import _lib as L
    from carel_vae_amd import drl_classifier as M
    if a.gemm_variant:           # tuning hooks exist in the experiments build only: the whole run then goes through libcarel_hip_exp.so
        L.experiments().__enter__()
    lib = L.load()
    L.check(lib.carel_init(local_rank), "carel_init")
    for v in a.gemm_variant:
        L.check(lib.carel_gemm_set_variant(v), "carel_gemm_set_variant")
    force_dp = os.environ.get("CAREL_FORCE_DP") == "1"      # exercise the RCCL path with a single rank (self-test)
    if world > 1 or force_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if os.environ.get("CAREL_REHEARSE_ONE_GPU") == "1":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from carel_vae_amd.data import synthetic_ecpe_batch
    opt = M.make_opt()
    cfg = M.encoder_config("zh")
    torch.manual_seed(0)
    model = M.DrlClassifier(opt, cfg, seed=0).to(dev)
    model.train()
    dp = None
    if world > 1 or force_dp:
        from carel_vae_amd.dp import DataParallel
        dp = DataParallel(model, wire_dtype=torch.bfloat16 if a.wire_dtype == "bf16" else None)
    # N > 1: each layer's Adam update starts as soon as that layer's gradient bucket has been all-reduced (dp.py), so the optimiser pass
    # hides behind the remaining backward kernels and collectives instead of trailing the last (embedding) bucket
    optim = torch.optim.Adam(model.get_params(), lr=opt.vae_lr) if a.torch_adam else M.FusedAdam(model, lr=opt.vae_lr, fuse_into_backward=(a.adam_in_backward if dp is not None else not a.adam_in_step) and not a.no_overlap)

    batches, lengths = [], []
    for i in range(4):
        b = synthetic_ecpe_batch(a.batch, 128, cfg.vocab_size, opt.pair_bow_dim, seed=1 + 10 * rank + i, shape=a.shape)
        lengths.append(b["attention_masks"].sum(1).tolist())     # known on the host before the H2D copy (as in a DataLoader)
        batches.append({k: v.to(dev) for k, v in b.items()})
    model.varlen = not a.no_varlen
    if dp is None:          # (DataParallel keeps the weight gradients on the main stream: its stream budget, carel_vae_amd/dp.py)
        model.overlap_wgrad = not a.no_overlap and os.environ.get("CAREL_BENCH_SERIAL_WGRAD") != "1"      # (the variable: A/B tool only)
    model.forward_chains = a.forward_chains and dp is None        # DataParallel switches the second chain off (dp.py)

    def step(i):
        b = batches[i % len(batches)]
        loss = model(b["input_ids"], b["attention_masks"], b["token_type_ids"], b["emo_labels"], b["cau_labels"], b["labels"],
                     b["bow_reps"], i % 41, seq_lengths=lengths[i % len(batches)])
        optim.zero_grad()
        loss.backward()
        optim.step()
        return loss

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def read_gemm_events():
        ms, fl, n = C.c_double(), C.c_double(), C.c_int64()
        L.check(lib.carel_profile_gemm_read(C.byref(ms), C.byref(fl), C.byref(n)))
        return ms.value, fl.value, n.value

    # ---- the timed region: SURVEY 8(d)'s step -- "H2D of a ready batch -> fwd -> bwd -> (all-reduce) -> Adam" with the reference's
    # running-loss read-back (:845-851) -- fed by the product's own input path: carel_vae_amd.data.PrefetchLoader over a BatchLoader (one
    # page-locked block per batch -- ids / mask / types / labels + the bag-of-words entries, expanded on the device inside the step -- ONE
    # async copy per batch on a copy stream, one batch ahead) and training.RunningLoss (device-side sum, page-locked slot + event every
    # 10 steps: the host never waits for the GPU).  `resident_inputs` below is the same step on batches that already sit in HBM.
    from carel_vae_amd import data as D
    from carel_vae_amd.training import RunningLoss
    n_feed = max(4, min(a.warmup + a.steps + 3, 96))              # batches in the synthetic dataset (cycled if the run is longer)
    feed_ds = D.SyntheticECPEDataset(n_feed * a.batch, opt.pair_bow_dim, 1000 + rank, vocab_size=cfg.vocab_size, shape=a.shape)
    feed_loader = D.PrefetchLoader(D.BatchLoader(feed_ds, batch_size=a.batch, shuffle=False), dev, depth=3)

    def feed():
        while True:
            for b in feed_loader:
                yield b
    feeder = feed()
    KEYS = ("input_ids", "attention_masks", "token_type_ids", "emo_labels", "cau_labels", "labels", "bow_reps")

    def step_fed(i):
        b = next(feeder)
        loss = model(*(b[k] for k in KEYS), i % 41, seq_lengths=b["seq_lengths"])
        optim.zero_grad()
        loss.backward()
        optim.step()
        return loss

    # (everything the timed region needs is allocated BEFORE the warm-up: a page-locked allocation right in front of the timed steps --
    # RunningLoss's read-back slots -- re-maps host memory into the GPU's page tables and made the second timed step 3 ms slower)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]     # step boundaries on the main stream (no host sync)
    loss_lines = []
    running = RunningLoss(dev, every=10, log=loss_lines.append)
    warm_running = RunningLoss(dev, every=10, log=lambda *_: None)
    log("model ready on %s; warm-up" % dev)
    for i in range(a.warmup):
        warm_running.add(step_fed(i).detach(), 0, i)
    warm_running.flush(wait=True)
    sync()
    log("timing %d steps (H2D of every batch + loss read-back every 10 steps inside)" % a.steps)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(a.steps):
        loss = step_fed(a.warmup + i)
        running.add(loss.detach(), 1, i)
        marks[i + 1].record()
    sync()
    dt = time.perf_counter() - t0
    running.flush(wait=True)
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
    median_ms = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.detach())
    log("timed region done: %.3f ms/step" % (1e3 * dt / a.steps))
    # ---- the same step on resident inputs (batches already in HBM, no read-back): what the H2D + read-back cost on this box ----
    for i in range(3):
        step(i)
    sync()
    marks_r = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    t_r = time.perf_counter()
    marks_r[0].record()
    for i in range(a.steps):
        step(3 + i)
        marks_r[i + 1].record()
    sync()
    dtr = time.perf_counter() - t_r
    if world > 1:
        t = torch.tensor([dtr], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dtr = float(t.item())
    resident = {"value": world * a.batch * a.steps / dtr, "unit": "clause-pairs/s", "ms_per_step": 1e3 * dtr / a.steps, "steps": a.steps,
                "headline_over_resident": dt / dtr, "ms_per_step_in_order": [round(marks_r[i].elapsed_time(marks_r[i + 1]), 3) for i in range(a.steps)],
                "note": "same step on batches that already sit in HBM, no loss read-back (rounds 1-3 quoted this as `value`); "
                        "headline_over_resident = what the H2D copy + read-back cost on this box"}
    log("resident inputs: %.3f ms/step (headline / resident = %.3f)" % (1e3 * dtr / a.steps, dt / dtr))
    # roofline leg: the SAME step, 3 more times, with HIP events bracketing every GEMM launch on its stream.  It is kept
    # out of the timed region because the event markers between kernels cost ~5 % of step time (no kernel overlap at the
    # boundaries), and it runs the kernels SERIALLY (weight gradients back on the main stream): with two streams the
    # brackets of concurrent kernels overlap and a per-kernel duration stops meaning anything.  The per-launch average
    # agrees with `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-overlap --no-ecpe` (profiles/).
    nprof = 3

    def gemm_event_replay(first_step):
        keep_overlap = model.overlap_wgrad
        model.overlap_wgrad = False
        hook, model._adam_hook = model._adam_hook, None
        L.check(lib.carel_profile_gemm(1, 260 * nprof))
        for i in range(nprof):
            step(first_step + i)
        torch.cuda.synchronize()
        ev = read_gemm_events()
        ov_e, ov_p = C.c_double(), C.c_double()
        L.check(lib.carel_profile_gemm_overheads(C.byref(ov_e), C.byref(ov_p)))
        L.check(lib.carel_profile_gemm(0, 0))
        model.overlap_wgrad = keep_overlap
        model._adam_hook = hook
        return ev, ov_e.value, ov_p.value
    def gemm_roofline(ev, ov_e, ov_p):
        """achieved / frac from the CONSERVATIVE launch time: event bracket around the launch minus a bare event pair, i.e. the whole
        interval the stream spends on the launch, dispatch gap included -- the figure that agrees with the rocprofv3 kernel_stats
        average committed under profiles/ (within ~3 %).  *_kernel_only: bracket minus the bracket around an EMPTY
        kernel (kernel time over an empty kernel's), the optimistic end of the same measurement."""
        ms, fl, n = ev
        us_lo = 1e3 * ms / n                               # bracket - empty-kernel bracket (subtracted inside the library)
        us_hi = us_lo + ov_e - ov_p                        # bracket - event pair
        return {"achieved": fl / n / (us_hi * 1e-6) / 1e12, "frac": fl / n / (us_hi * 1e-6) / 1e12 / PEAK_BF16_TFLOPS,
                "avg_launch_us": us_hi, "avg_launch_us_kernel_only": us_lo, "frac_kernel_only": fl / n / (us_lo * 1e-6) / 1e12 / PEAK_BF16_TFLOPS,
                "launches_per_step": n / nprof, "alg_gflop_per_launch": fl / n / 1e9, "gemm_ms_per_step": us_hi * n / nprof * 1e-3,
                "event_calibration_us": {"empty_kernel_bracket": ov_e, "event_pair": ov_p}}
    ev_timed, ov_empty_v, ov_pair_v = gemm_event_replay(a.warmup + a.steps)
    pairs_per_s = world * a.batch * a.steps / dt

    # ---- secondary line: the same step on ECPE-shaped batches (SURVEY 8(d) shape-B: ~77 % padding), padding skipped ----
    ecpe = None
    if a.shape == "A" and not a.no_varlen and not a.no_ecpe:
        bb, ll = [], []
        for i in range(4):
            b = synthetic_ecpe_batch(a.batch, 128, cfg.vocab_size, opt.pair_bow_dim, seed=101 + 10 * rank + i, shape="B")
            ll.append(b["attention_masks"].sum(1).tolist())
            bb.append({k: v.to(dev) for k, v in b.items()})
        keep = (batches, lengths)
        batches, lengths = bb, ll
        for i in range(3):
            step(i)
        sync()
        t1 = time.perf_counter()
        nb = max(5, a.steps // 2)
        for i in range(nb):
            step(i)
        sync()
        dtb = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dtb], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtb = float(t.item())
        ecpe = {"value": world * a.batch * nb / dtb, "unit": "clause-pairs/s", "ms_per_step": 1e3 * dtb / nb, "steps": nb,
                "attended_tokens_per_pair": sum(sum(l) for l in ll) / (len(ll) * a.batch),
                "note": "same step, ECPE-shaped lengths; padded positions are not run through the encoder (results identical)"}
        # the roofline fraction of this leg is computed from EFFECTIVE-token FLOPs (attended tokens only)
        eff = sum(effective_flops(ll[i % 4]) for i in range(nb))            # this rank's nb steps
        ev_b, ov_e_b, ov_p_b = gemm_event_replay(0)
        ecpe["roofline"] = {"bound": "mfma", "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "achieved": world * eff / dtb / 1e12,
                            "frac": world * eff / dtb / 1e12 / (world * PEAK_BF16_TFLOPS),
                            "effective_gflop_per_pair": eff / (nb * a.batch) / 1e9,
                            "note": "whole step: effective-token algorithmic FLOPs (linear x l/128, attention x (l/128)^2 per pair; dense = "
                                    "67.05 GFLOP/pair) / step time.  gemm_kernels: the GEMM launches of the same step replayed serially with HIP "
                                    "events; their FLOPs count the launched shapes (packed rows rounded up to 128)",
                            "gemm_kernels": gemm_roofline(ev_b, ov_e_b, ov_p_b) if ev_b[2] else None}
        batches, lengths = keep
        log("ECPE-shaped leg: %.3f ms/step" % (1e3 * dtb / nb))

    # ---- evaluation twin: get_pair_preds on a test-set-sized batch of ECPE-shaped pairs (:265-282, :958) ----
    infer = None
    if not a.no_ecpe and rank == 0:
        tb = synthetic_ecpe_batch(2048, 128, cfg.vocab_size, opt.pair_bow_dim, seed=777, shape="B")
        ti, ta_, tt_ = (tb[k].to(dev) for k in ("input_ids", "attention_masks", "token_type_ids"))
        model.eval()
        with torch.no_grad():
            model.pair_probabilities(ti, ta_, tt_)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(3):
                prob = model.pair_probabilities(ti, ta_, tt_)
            torch.cuda.synchronize()
            dti = (time.perf_counter() - t2) / 3
        model.train()
        infer = {"value": 2048 / dti, "unit": "clause-pairs/s", "ms_per_2048_pairs": 1e3 * dti,
                 "note": "pair_probabilities (the kernel path of get_pair_preds): eval-mode encoder + fresh noise, 2048 ECPE-shaped pairs, "
                         "chunks of 1024, padding skipped; one GPU"}
        log("inference leg: %.1f ms per 2048 pairs" % (1e3 * dti))

    # ---- config 4: the English three-space adversarial model (drl_classifier_en.py), one GPU ----
    english = None
    pipeline = sentence = ablation = None
    if not a.no_ecpe and rank == 0 and world == 1:
        pipeline = input_pipeline_leg(dev, model, optim, a.batch)
        english = english_leg(dev, a.batch, max(5, a.steps // 2))
        sentence = sentence_transformer_leg(dev, max(5, a.steps // 2))
        ablation = ablation_leg(dev, a.batch, max(5, a.steps // 2), cfg.vocab_size)

    # ---- roofline of the dominant kernel family ----
    roof = None
    ms_t, fl_t, n_t = ev_timed

    if n_t:
        traffic = None
        try:        # HBM traffic per GEMM launch from the committed PMC summary (rocprofv3 --pmc passes)
            for line in open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_CSV)):
                if line.startswith('"ALL carel::gemm'):
                    f = line.rsplit(",", 3)
                    traffic = (float(f[2]) + float(f[3])) * 1e6        # bytes per GEMM launch (fetch x2-corrected + write)
        except OSError:
            pass
        mfma_util = None
        try:        # matrix-core occupancy of the same kernels from the committed PMC pass (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles))
            for line in open(os.path.join(ROOT, "profiles", PMC_MFMA_CSV)):
                if line.startswith('"ALL carel::gemm'):
                    mfma_util = float(line.split(",")[4])
        except (OSError, ValueError, IndexError):
            pass
        roof = {"bound": "mfma", "kernel": "carel::gemm_pp_kernel + carel::gemm_kernel (every bf16 MFMA GEMM launch of the step: fwd NT, dgrad NN, wgrad TN)",
                "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "traffic": traffic,
                "traffic_unit": "bytes per launch, HBM/fabric side",
                "traffic_source": "STATIC: profiles/%s, committed with this code (rocprofv3 --pmc FETCH_SIZE x2 and --pmc WRITE_SIZE, separate passes of the serial dense run, tools/pmc_traffic.sh; algorithmic operand+output bytes average ~40e6); achieved / frac / avg_launch_us are measured live" % PMC_TRAFFIC_CSV,
                "mfma_util_pmc": mfma_util, "mfma_util_source": "STATIC: profiles/%s (tools/pmc_mfma.sh), committed with this code" % PMC_MFMA_CSV,
                "note": "per-kernel durations from a serial replay of the step (wgrad_side_stream off); the timed region overlaps them",
                "whole_step_frac_of_peak": (world * a.batch * FLOP_PER_PAIR * a.steps / dt) / (world * PEAK_BF16_TFLOPS * 1e12)}
        roof.update(gemm_roofline(ev_timed, ov_empty_v, ov_pair_v))

    out = {"metric": "clause-pairs/sec (training step)", "value": pairs_per_s, "unit": "clause-pairs/s", "n_gpus": world,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "ms_per_step_median": median_ms,
           "ms_per_step_min_max": [per_step[0], per_step[-1]], "ms_per_step_in_order": [round(marks[i].elapsed_time(marks[i + 1]), 3) for i in range(a.steps)], "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": "zh ECPE training step (H2D of a ready batch + fwd + bwd + Adam, loss read back every 10 steps), BERT-base "
                                  "vocab 21128, S=128 shape-%s, B=%d/GPU, bow V=23771, dropout on, random-init weights" % (a.shape, a.batch),
                      "input_path": "carel_vae_amd.data.PrefetchLoader(BatchLoader): one page-locked block + one async H2D per batch on a copy stream",
                      "global_batch": world * a.batch, "seq_len": 128, "parallelism": "dp%d" % world,
                      "gradient_wire_dtype": (a.wire_dtype if dp is not None else None),
                      "attended_tokens_per_pair": sum(sum(l) for l in lengths) / (len(lengths) * a.batch),
                      "padding_skipped": bool(model.varlen and a.shape == "B"),
                      "optimizer": "torch.optim.Adam" if a.torch_adam else "fused HIP Adam",
                      "wgrad_side_stream": bool(model.overlap_wgrad), "forward_chains": bool(model.overlap_wgrad and model.forward_chains),
                      "adam_in_backward": bool(getattr(optim, "_aux", None) is not None)},
           "roofline": roof, "resident_inputs": resident, "running_loss_lines": loss_lines[-2:], "ecpe_shaped": ecpe, "ablation_heads": ablation, "inference": infer, "english_adversarial": english, "input_pipeline": pipeline,
           "sentence_transformer": sentence, "final_loss": final_loss}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        def hip_loss(P0, batch, eps_e, eps_c, ocfg2, oopt, fp32=False):
            m2 = M.DrlClassifier(M.make_opt(**vars(oopt)), M.encoder_config("zh", hidden_dropout=0.0, attn_dropout=0.0), seed=0)
            m2.load_state_dict(P0)
            m2.to(dev).train()
            m2.opt.dropout = 0.0
            m2.debug_fp32 = fp32
            m2.set_noise(eps_e, eps_c)
            b = {k: v.to(dev) for k, v in batch.items()}
            t = m2.forward_terms(b["input_ids"], b["attention_masks"], b["token_type_ids"], b["emo_labels"], b["cau_labels"],
                                 b["labels"], b["bow_reps"], 3)
            return {k: float(v) for k, v in t.items() if v.numel() == 1}
        base, parity = cpu_baseline({}, {"dropout": 0.0}, hip_loss)
        out["cpu_baseline"] = base
        out["elbo_parity"] = parity
        out["speedup_vs_cpu_baseline"] = pairs_per_s / base["value"]
    out["ranks_seen"] = world            # (self-check for a scaling run: must equal n_gpus and the --gpus argument; launch_ranks() asserts it)
    if rank == 0:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
    if world > 1 or force_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
