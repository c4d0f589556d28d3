"""End-to-end parity of carel_vae_amd.DrlClassifier (HIP) against
  (a) the golden vectors produced by the reference's own classes (fp32 CPU)      -> bf16-level tolerance
  (b) the CPU oracle run with bf16 rounding at the kernels' storage points       -> tight tolerance
for forward terms, latent means, gradients and a 3-step Adam trajectory."""
import os

import numpy as np
import pytest
import torch

from carel_vae_amd import _lib as L
from carel_vae_amd import drl_classifier as M
from oracle import carel_oracle as O

pytestmark = pytest.mark.gpu

CASES = {
    "zh_small": (O.EncoderConfig(layers=2, vocab_size=1000), O.Opt(pair_bow_dim=257, dropout=0.0)),
    "zh_ragged": (O.EncoderConfig(layers=2, vocab_size=1000), O.Opt(pair_bow_dim=513, dropout=0.0)),
    "zh_allneg": (O.EncoderConfig(layers=1, vocab_size=500), O.Opt(pair_bow_dim=130, dropout=0.0)),
    "zh_s64": (O.EncoderConfig(layers=2, vocab_size=800), O.Opt(pair_bow_dim=300, dropout=0.0)),
    "en_small": (O.EncoderConfig(layers=2, vocab_size=1200, max_pos=514, type_vocab=1, ln_eps=1e-5, variant="roberta", pad_id=1),
                 O.Opt(language="en", pair_bow_dim=257, dropout=0.0)),
    "zh_full12": (O.EncoderConfig(), O.Opt(pair_bow_dim=1000, dropout=0.0)),
}
TERMS = ("mmd", "emo", "cau", "pair", "kl_e", "kl_c", "rec")
# bf16 encoder (fp32 accumulate, fp32 residual stream) vs the fp32 reference.  Measured on MI355X (tools/parity_report.py,
# round 2): main loss terms <= 3.6e-4 relative, the two KL terms (weight 7e-6, sums of squares of the latents) <= 1.1e-3,
# latents / pooler output 1.3e-3 .. 6.7e-3 of their norm (2 .. 12 layers), total loss 1.2e-4 .. 8.9e-4 relative wherever
# the total is not a near-cancellation of its weighted terms, and <= 1.5e-4 of the terms' scale everywhere.
# north_star: "ELBO matching to 1e-3 rel" -- asserted at the north-star configuration in
# test_bench_configuration_elbo_within_1e_3_of_cpu_fp32 (measured 1.2e-4 / 1.5e-4) and, with the margin the 8-sample
# golden needs (measured 8.9e-4), below.
TOL_GRAD_BF16_EMU = 5e-3        # per-tensor gradient vs the oracle that emulates the library's bf16 storage points in both directions (measured median 2e-3)
TOL_GRAD_BF16_EMU_QK = 1.5e-2   # ... for the query / key projections (measured 0.8 .. 1.2e-2: see the test)
TOL_TERM_BF16 = 1e-3            # mmd, emo, cau, pair, rec
TOL_KL_BF16 = 3e-3              # kl_e, kl_c
TOL_LATENT_BF16 = 1e-2
TOL_LOSS_BF16 = 1.5e-3          # |d loss| / |loss| when |loss| >= 10 % of sum |w_i t_i|
TOL_LOSS_OVER_SCALE = 3e-4      # |d loss| / sum |w_i t_i|, every case
WEIGHTS = dict(mmd=30.0, emo=10.0, cau=10.0, pair=30.0, kl_e=1.0, kl_c=1.0, rec=1.0)


def check_fp32_parity(out, ref, loss_tol=TOL_LOSS_BF16):
    """out: HIP forward_terms; ref: fp32 values (golden or CPU oracle) for TERMS, 'loss' and the latents."""
    for k in ("pooled", "mu_e", "lv_e", "mu_c", "lv_c"):
        assert relnorm(out[k], ref[k]) < TOL_LATENT_BF16, k
    for k in TERMS:
        r = float(ref[k])
        tol = TOL_KL_BF16 if k.startswith("kl") else TOL_TERM_BF16
        assert abs(float(out[k]) - r) <= tol * max(abs(r), 1e-3), (k, float(out[k]), r)
    scale = sum(abs(WEIGHTS[k] * float(ref[k])) for k in TERMS)
    dl, rl = abs(float(out["loss"]) - float(ref["loss"])), abs(float(ref["loss"]))
    assert dl <= TOL_LOSS_OVER_SCALE * scale, (dl, scale)
    if rl >= 0.1 * scale:
        assert dl <= loss_tol * rl, (float(out["loss"]), float(ref["loss"]))


def build(cfg, opt, wseed, train_dropout=False):
    mcfg = M.encoder_config("en" if cfg.variant == "roberta" else "zh", vocab_size=cfg.vocab_size, max_pos=cfg.max_pos,
                            type_vocab=cfg.type_vocab, ln_eps=cfg.ln_eps, layers=cfg.layers,
                            hidden_dropout=cfg.hidden_dropout if train_dropout else 0.0,
                            attn_dropout=cfg.attn_dropout if train_dropout else 0.0)
    mopt = M.make_opt(**vars(opt))
    model = M.DrlClassifier(mopt, mcfg)
    P = O.init_params(cfg, opt, seed=wseed)
    model.load_state_dict(P)
    model.to("cuda")
    return model, P


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)
    batch = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    return z, batch


def call(model, batch, it):
    b = {k: v.cuda() for k, v in batch.items()}
    return (b["input_ids"], b["attention_masks"], b["token_type_ids"], b["emo_labels"], b["cau_labels"], b["labels"], b["bow_reps"], it)


def relnorm(a, b):
    a, b = a.double().cpu().flatten(), b.double().cpu().flatten()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


@pytest.mark.parametrize("name", list(CASES))
def test_forward_terms_vs_golden_and_bf16_oracle(golden_dir, name):
    cfg, opt = CASES[name]
    z, batch = load(golden_dir, name)
    B, S, Lr, vocab, V, wseed, bseed, steps, it0 = (int(v) for v in z["meta"])
    model, P = build(cfg, opt, wseed)
    model.train()
    eps_e, eps_c = torch.from_numpy(z["eps_e_0"]), torch.from_numpy(z["eps_c_0"])
    model.set_noise(eps_e, eps_c)
    out = model.forward_terms(*call(model, batch, it0))
    # (a) golden, fp32 reference: every term, the latents and the TOTAL (the reference's weighted sum, :256-261)
    gold = {k: torch.from_numpy(z["t_" + k]) for k in TERMS}
    gold.update({k: torch.from_numpy(z[k]) for k in ("pooled", "mu_e", "lv_e", "mu_c", "lv_c")})
    gold["loss"] = sum(WEIGHTS[k] * float(z["t_" + k]) * (-1.0 if k == "mmd" else 1.0) for k in TERMS)
    assert abs(gold["loss"] - float(z["losses"][0])) <= 2e-5 * max(1.0, abs(gold["loss"]))   # the reference forward()'s own step-0 value
    check_fp32_parity(out, gold)
    # (b) oracle with bf16 rounding where the kernels store bf16: the kernels themselves must be ~exact
    ref = O.forward_terms(P, batch, it0, cfg, opt, eps_e, eps_c, quant=O.bf16_round)
    # the emulation rounds at the same places but not bit-identically (e.g. probabilities are rounded after
    # normalisation in the oracle, before it in the kernel); the residue grows ~sqrt(layers)
    tol = 3e-3 * max(1.0, (cfg.layers / 2) ** 0.5)
    assert relnorm(out["pooled"], ref["pooled"]) < tol
    for k in ("mu_e", "lv_e", "mu_c", "lv_c"):
        assert relnorm(out[k], ref[k]) < tol, k
    for k in TERMS:
        r = float(ref[k])
        assert abs(float(out[k]) - r) <= tol * max(abs(r), 1e-3) + 1e-6, (k, float(out[k]), r)
    # the total is a sum of opposing-sign weighted terms (|-30 mmd| ~ 48 vs total ~ 0.2): judge it on that scale
    scale = sum(abs(w * float(ref[k])) for w, k in ((opt.mmd_loss_weight, "mmd"), (opt.emo_mul_loss_weight, "emo"),
                                                   (opt.cau_mul_loss_weight, "cau"), (opt.pair_mul_loss_weight, "pair")))
    assert abs(float(out["loss"]) - float(ref["loss"])) <= tol * scale, (float(out["loss"]), float(ref["loss"]), scale)


TOL_FP32_DEBUG = 1e-5


@pytest.mark.parametrize("name", ["zh_small", "zh_ragged", "zh_s64", "en_small", "zh_full12"])
def test_fp32_debug_mode_matches_the_reference_fp32_outputs(golden_dir, name):
    """"<= 1e-5 (fp32 debug mode)".  model.debug_fp32 runs the encoder through carel_encoder_forward_f32 (fp32 MFMA
    linears, fp32 attention / GELU / LayerNorm; the tail is fp32 anyway): every term, the total and the latents must then match the
    golden vectors -- outputs of the reference's own classes in fp32 -- to 1e-5, i.e. what the bf16 path differs by is bf16 rounding."""
    cfg, opt = CASES[name]
    z, batch = load(golden_dir, name)
    it0 = int(z["meta"][8])
    model, _ = build(cfg, opt, int(z["meta"][5]))
    model.train()
    model.debug_fp32 = True
    model.set_noise(torch.from_numpy(z["eps_e_0"]), torch.from_numpy(z["eps_c_0"]))
    out = model.forward_terms(*call(model, batch, it0))
    worst = 0.0
    for k in ("pooled", "mu_e", "lv_e", "mu_c", "lv_c"):
        e = relnorm(out[k], torch.from_numpy(z[k]))
        worst = max(worst, e)
        assert e < TOL_FP32_DEBUG, (k, e)
    for k in TERMS:
        r = float(z["t_" + k])
        e = abs(float(out[k]) - r) / max(abs(r), 1e-3)
        worst = max(worst, e)
        assert e <= TOL_FP32_DEBUG, (k, float(out[k]), r)
    scale = sum(abs(WEIGHTS[k] * float(z["t_" + k])) for k in TERMS)
    assert abs(float(out["loss"]) - float(z["losses"][0])) <= TOL_FP32_DEBUG * scale
    print("fp32 debug mode, %s: worst relative difference %.2e" % (name, worst))
    # and it is a forward-only mode: a training forward refuses loudly
    with pytest.raises(Exception, match="forward-only"):
        model(*call(model, batch, it0))


def test_mpnet_encoder_under_the_model_bf16_and_fp32_debug_vs_cpu_fp32_oracle():
    """The MPNet encoder (relative-position attention bias, RoBERTa-style position ids, no token types:
    en_ec_sentence_transformer.py:22) under DrlClassifier, ECPE-like ragged batch: the bf16 path within the bf16 tolerances of the CPU
    fp32 oracle (whose MPNet branch is pinned to transformers.MPNetModel, tests/test_oracle_triplet.py) and the fp32 debug path within
    1e-5 -- the fp32 attention kernel's bias-by-distance branch included."""
    cfg = O.EncoderConfig(layers=2, vocab_size=1200, max_pos=514, type_vocab=1, ln_eps=1e-5, variant="mpnet", pad_id=1, rel_pos=True)
    opt = O.Opt(pair_bow_dim=257, dropout=0.0)
    mcfg = M.encoder_config("mpnet", vocab_size=cfg.vocab_size, layers=cfg.layers, hidden_dropout=0.0, attn_dropout=0.0)
    model = M.DrlClassifier(M.make_opt(**vars(opt)), mcfg)
    P = O.init_params(cfg, opt, seed=11)
    model.load_state_dict(P)
    model.to("cuda").train()
    batch = O.synthetic_batch(16, 64, cfg, opt.pair_bow_dim, seed=5, shape="B")
    ids, att = batch["input_ids"], batch["attention_masks"]
    ids[(ids == cfg.pad_id) & (att == 1)] = 2               # the pad id only where the mask says padding
    g = torch.Generator().manual_seed(7)
    eps_e, eps_c = torch.randn(opt.ec_dim, generator=g), torch.randn(opt.ec_dim, generator=g)
    ref = O.forward_terms(P, batch, 3, cfg, opt, eps_e, eps_c)
    model.set_noise(eps_e, eps_c)
    out = model.forward_terms(*call(model, batch, 3))
    check_fp32_parity(out, ref)
    model.debug_fp32 = True
    model.set_noise(eps_e, eps_c)
    out32 = model.forward_terms(*call(model, batch, 3))
    for k in ("pooled", "mu_e", "lv_e", "mu_c", "lv_c"):
        assert relnorm(out32[k], ref[k]) < TOL_FP32_DEBUG, k
    for k in TERMS:
        r = float(ref[k])
        assert abs(float(out32[k]) - r) <= TOL_FP32_DEBUG * max(abs(r), 1e-3), (k, float(out32[k]), r)


@pytest.mark.parametrize("shape", ["A", "B"])
def test_bench_configuration_elbo_within_1e_3_of_cpu_fp32(shape):
    """The north-star tolerance at the north-star configuration (BASELINE.json configs[1]): B = 64, S = 128, 12 layers,
    vocabulary 21 128, V = 23 771, dropout off, dense (A) and ECPE-shaped (B, padding skipped) batches, against the CPU
    fp32 oracle (held to the reference's own outputs by tests/test_oracle_golden.py): ELBO within 1e-3 relative."""
    cfg, opt = O.EncoderConfig(), O.Opt(dropout=0.0)
    assert (cfg.layers, cfg.vocab_size, opt.pair_bow_dim) == (12, 21128, 23771)
    model, P = build(cfg, opt, 0)
    model.train()
    batch = O.synthetic_batch(64, 128, cfg, opt.pair_bow_dim, seed=1, shape=shape)
    g = torch.Generator().manual_seed(3)
    eps_e, eps_c = torch.randn(opt.ec_dim, generator=g), torch.randn(opt.ec_dim, generator=g)
    model.set_noise(eps_e, eps_c)
    out = model.forward_terms(*call(model, batch, 3))
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    ref = O.forward_terms(P, batch, 3, cfg, opt, eps_e, eps_c)
    check_fp32_parity(out, ref, loss_tol=1e-3)
    assert abs(float(out["loss"]) - float(ref["loss"])) <= 1e-3 * abs(float(ref["loss"]))


@pytest.mark.parametrize("name", ["zh_small", "zh_ragged", "en_small", "zh_s64"])
def test_gradients_vs_oracle(golden_dir, name):
    cfg, opt = CASES[name]
    z, batch = load(golden_dir, name)
    B, S, Lr, vocab, V, wseed, bseed, steps, it0 = (int(v) for v in z["meta"])
    model, P = build(cfg, opt, wseed)
    model.train()
    eps_e, eps_c = torch.from_numpy(z["eps_e_0"]), torch.from_numpy(z["eps_c_0"])
    model.set_noise(eps_e, eps_c)
    loss = model(*call(model, batch, it0))
    loss.backward()
    torch.cuda.synchronize()
    out, grads = O.loss_and_grads(P, batch, it0, cfg, opt, eps_e, eps_c)
    named = dict(model.named_parameters())
    worst = {}
    for k, g in grads.items():
        got = named[k].grad
        assert got is not None, k
        if g is None or float(g.norm()) < 1e-7:      # key bias: analytically zero gradient (softmax shift invariance);
            qb = named[k.replace("key", "query")].grad  # what is left is bf16 rounding noise of dS, small next to dq's bias grad
            assert float(got.norm()) < 0.05 * float(qb.norm()) + 1e-4, (k, float(got.norm()), float(qb.norm()))
            continue
        worst[k] = relnorm(got, g)
    bad = {k: v for k, v in worst.items() if v > 4e-2}
    assert not bad, bad
    assert np.median(list(worst.values())) < 1.5e-2
    # golden slices (reference fp32): direction agreement
    for k in z.files:
        if k.startswith("g_") and float(z["gn_" + k[2:]]) > 1e-7:
            pk = k[2:]
            f = named[pk].grad.detach().cpu().reshape(-1)
            n = 64
            step = max(1, f.numel() // n)
            got = torch.cat((f[:n], f[-n:], f[::step][:n])).numpy()
            ref = z[k]
            den = np.linalg.norm(ref)
            if den > 1e-9:
                assert np.linalg.norm(got - ref) / den < 8e-2, pk


def test_three_step_adam_trajectory(golden_dir):
    cfg, opt = CASES["zh_small"]
    z, batch = load(golden_dir, "zh_small")
    B, S, Lr, vocab, V, wseed, bseed, steps, it0 = (int(v) for v in z["meta"])
    model, P = build(cfg, opt, wseed)
    model.train()
    optim = M.FusedAdam(model, lr=opt.vae_lr)
    losses = []
    for s in range(steps):
        model.set_noise(torch.from_numpy(z[f"eps_e_{s}"]), torch.from_numpy(z[f"eps_c_{s}"]))
        loss = model(*call(model, batch, it0 + s))
        optim.zero_grad()
        loss.backward()
        optim.step()
        losses.append(float(loss))
    ref = z["losses"]
    # the total is a sum of opposing-sign terms (|terms| ~ 50, total ~ 0.2): compare on the terms' scale
    scale = 30 * 1.6 + 10 * 2.0 + 10 * 0.75 + 30 * 0.7
    assert np.abs(np.array(losses) - ref).max() < 6e-3 * scale, (losses, ref)
    sd = model.state_dict()
    for k in z.files:
        if k.startswith("w_"):
            pk = k[2:]
            f = sd[pk].detach().cpu().reshape(-1)
            n = 64
            step = max(1, f.numel() // n)
            got = torch.cat((f[:n], f[-n:], f[::step][:n])).numpy()
            # every Adam step moves an element by ~+-lr; an element whose gradient is within bf16 noise of 0 can flip
            # sign, so allow up to 2*steps*lr on a few elements and demand lr-level agreement on the rest
            d = np.abs(got - z[k])
            assert d.max() <= 2 * steps * opt.vae_lr * 1.01, pk
            if not pk.endswith("key.bias"):
                assert (d <= 1.2e-5).mean() >= 0.97, (pk, float((d <= 1.2e-5).mean()))
    P0 = O.init_params(cfg, opt, seed=wseed)
    for n in ("emotion_mu.weight", "cause_log_var.bias"):          # quirk Q3: latent heads never move
        assert torch.equal(sd[n].cpu(), P0[n])


@pytest.mark.parametrize("fused", [True, False])
def test_adam_trajectory_with_an_all_negative_step_in_the_middle(golden_dir, fused):
    """tests/golden/zh_negmid.npz: the reference's class for four steps, step 1 with every pair label 0 -- its pair loss is
    the int 0 (:510-511), pair_classifier.grad is None and torch.optim.Adam skips that parameter WITHOUT advancing its own
    step counter, so from step 2 on the pair head's bias corrections lag the other parameters' by one.  FusedAdam keeps
    that count on the device (carel_adam_args.skip_count); with one global step count the pair head's two later updates
    come out ~14 % short (2.5e-6 of a 1e-5 move), which the 8e-7 bound below catches.  Stock torch.optim.Adam on the same
    model agrees too once model.strict_pair_skip makes the model leave .grad None for the frozen head (one host read per
    step, like the reference's own isinf().any())."""
    cfg, opt = O.EncoderConfig(layers=1, vocab_size=500), O.Opt(pair_bow_dim=130, dropout=0.0)
    z, batch = load(golden_dir, "zh_negmid")
    B, S, Lr, vocab, V, wseed, bseed, steps, it0 = (int(v) for v in z["meta"])
    neg = set(z["neg_steps"].tolist())
    assert steps == 4 and neg == {1}
    model, P = build(cfg, opt, wseed)
    model.train()
    optim = M.FusedAdam(model, lr=opt.vae_lr) if fused else torch.optim.Adam(model.get_params(), lr=opt.vae_lr)
    model.strict_pair_skip = not fused
    batch_neg = dict(batch, labels=torch.zeros_like(batch["labels"]), cau_labels=torch.zeros_like(batch["cau_labels"]))
    losses = []
    for s in range(steps):
        model.set_noise(torch.from_numpy(z[f"eps_e_{s}"]), torch.from_numpy(z[f"eps_c_{s}"]))
        loss = model(*call(model, batch_neg if s in neg else batch, it0 + s))
        optim.zero_grad()
        loss.backward()
        optim.step()
        losses.append(float(loss))
    scale = 30 * 1.6 + 10 * 2.0 + 10 * 0.75 + 30 * 1.05
    assert np.abs(np.array(losses) - z["losses"]).max() < 6e-3 * scale, (losses, z["losses"])
    sd = model.state_dict()
    for pk in ("pair_classifier.weight", "pair_classifier.bias"):
        f = sd[pk].detach().cpu().reshape(-1)
        n = 64
        step = max(1, f.numel() // n)
        got = torch.cat((f[:n], f[-n:], f[::step][:n])).numpy()
        d = np.abs(got - z["w_" + pk])
        assert np.median(d) <= 3e-7 and (d <= 8e-7).mean() >= 0.9, (pk, float(np.median(d)), float(d.max()))
    if fused:
        assert float(optim._skip_count) == 1.0


def test_torch_adam_drop_in_and_dropout_parity(golden_dir):
    """torch.optim.Adam(model.get_params()) works unchanged; with dropout ON the HIP step equals the oracle
    fed the same counter-based masks."""
    cfg, opt = CASES["zh_small"]
    opt = O.Opt(**{**vars(opt), "dropout": 0.5})
    z, batch = load(golden_dir, "zh_small")
    B, S, Lr, vocab, V, wseed, bseed, steps, it0 = (int(v) for v in z["meta"])
    model, P = build(cfg, opt, wseed, train_dropout=True)
    model.train()
    optim = torch.optim.Adam(model.get_params(), lr=1e-5)
    eps_e, eps_c = torch.from_numpy(z["eps_e_0"]), torch.from_numpy(z["eps_c_0"])
    model.set_noise(eps_e, eps_c)
    loss = model(*call(model, batch, it0))
    seed = model._last_call.seed
    terms = {k: float(v) for k, v in model.last_terms().items()}
    ref = O.forward_terms(P, batch, it0, cfg, opt, eps_e, eps_c, train=True, seed=seed, quant=O.bf16_round)
    for k in TERMS + ("loss",):
        r = float(ref[k])
        assert abs(terms[k] - r) <= 4e-3 * max(abs(r), 1e-3) + 1e-6, (k, terms[k], r)
    optim.zero_grad()
    loss.backward()
    w0 = model.encoder.encoder.layer[0].intermediate.dense.weight.detach().clone()
    optim.step()
    w1 = model.encoder.encoder.layer[0].intermediate.dense.weight.detach()
    assert float((w1 - w0).abs().max()) > 0          # parameters moved through the stock optimiser
    # next forward sees the new weights (bf16 shadow refreshed)
    model.set_noise(eps_e, eps_c)
    t2 = model.forward_terms(*call(model, batch, it0))
    assert float(t2["loss"]) != terms["loss"]


def test_get_pair_preds_and_cpu_refusal(golden_dir):
    cfg, opt = CASES["zh_ragged"]
    z, batch = load(golden_dir, "zh_ragged")
    B, S, Lr, vocab, V, wseed, bseed, steps, it0 = (int(v) for v in z["meta"])
    model, P = build(cfg, opt, wseed)
    model.eval()
    eps_e, eps_c = torch.randn(24), torch.randn(24)
    model.set_noise(eps_e, eps_c)
    preds = model.get_pair_preds(batch["input_ids"].cuda(), batch["attention_masks"].cuda(), batch["token_type_ids"].cuda())
    assert isinstance(preds, list) and len(preds) == B and preds[0][0] in (0.0, 1.0)
    prob = O.pair_preds(P, batch["input_ids"], batch["attention_masks"], batch["token_type_ids"], cfg, opt, eps_e, eps_c).squeeze(1)
    far = (prob - 0.5).abs() > 2e-2
    assert torch.equal(torch.tensor(preds).squeeze(1)[far], prob.round()[far])
    with pytest.raises(M.L.CarelError):
        model.get_pair_preds(batch["input_ids"], batch["attention_masks"], batch["token_type_ids"])   # CPU tensors


@pytest.mark.parametrize("name,dropout", [("zh_ragged", False), ("zh_ragged", True), ("en_small", True)])
def test_token_packing_equals_padded_computation(golden_dir, name, dropout):
    """Skipping the padded positions (varlen packing) must not change anything: same loss terms, latents and
    gradients as the padded computation, with dropout masks ON (they hash the original (sample, position) index)."""
    cfg, opt = CASES[name]
    if dropout:
        opt = O.Opt(**{**vars(opt), "dropout": 0.5})
    z, batch = load(golden_dir, name)
    B, S, Lr, vocab, V, wseed, bseed, steps, it0 = (int(v) for v in z["meta"])
    eps_e, eps_c = torch.from_numpy(z["eps_e_0"]), torch.from_numpy(z["eps_c_0"])
    res = {}
    for varlen in (False, True):
        model, P = build(cfg, opt, wseed, train_dropout=dropout)
        model.train()
        model.varlen = varlen
        model.set_noise(eps_e, eps_c)
        loss = model(*call(model, batch, it0))
        packed = model._last_call.pack is not None
        assert packed == varlen
        loss.backward()
        torch.cuda.synchronize()
        res[varlen] = (float(loss), {k: float(v) for k, v in model.last_terms().items()}, model._last_call.buf.lat.clone(),
                       {k: p.grad.detach().clone() for k, p in model.named_parameters()})
    (l0, t0, lat0, g0), (l1, t1, lat1, g1) = res[False], res[True]
    # not bitwise: the packed batch is small enough for the split-K GEMM path, whose fp32 summation order differs, and a
    # different last bit before a bf16 store is a 2^-8 relative change of that element
    for k in t0:
        assert abs(t0[k] - t1[k]) <= 1e-3 * max(abs(t0[k]), 1e-3), (k, t0[k], t1[k])
    assert relnorm(lat1, lat0) < 2e-3
    # (key biases excluded: their gradient is analytically zero, what is stored is rounding noise)
    worst = max(relnorm(g1[k], g0[k]) for k in g0 if float(g0[k].norm()) > 1e-6 and not k.endswith("key.bias"))
    assert worst < 2e-2, worst
    # a host-provided length list gives the same packing without the device->host read
    lens = batch["attention_masks"].sum(1).tolist()
    model.set_noise(eps_e, eps_c)
    model._fwd_count -= 1               # same dropout seed as the previous call
    args = call(model, batch, it0)
    with torch.no_grad():
        l2 = model(*args, seq_lengths=lens)
    assert abs(float(l2) - l1) <= 1e-5 * max(abs(l1), 1e-3)          # same path twice: identical


def test_packing_falls_back_to_dense_for_non_prefix_masks(golden_dir):
    cfg, opt = CASES["zh_small"]
    z, batch = load(golden_dir, "zh_small")
    model, P = build(cfg, opt, int(z["meta"][5]))
    model.eval()
    b = {k: v.clone() for k, v in batch.items()}
    b["attention_masks"][:, 5] = 0           # a hole in the middle: not right-padding
    model.set_noise(torch.zeros(24), torch.zeros(24))
    out = model.forward_terms(*call(model, b, 0))
    assert model._last_call.pack is None
    ref = O.forward_terms(P, b, 0, cfg, opt, torch.zeros(24), torch.zeros(24), quant=O.bf16_round)
    assert abs(float(out["emo"]) - float(ref["emo"])) < 5e-3 * abs(float(ref["emo"]))


@pytest.mark.parametrize("name", ["zh_small", "zh_ragged", "zh_allneg"])
def test_cls_only_last_layer_equals_full_computation(golden_dir, name):
    """Dead-row elimination: running the last layer's row-wise half on the [CLS] rows only changes nothing."""
    cfg, opt = CASES[name]
    opt = O.Opt(**{**vars(opt), "dropout": 0.5})
    z, batch = load(golden_dir, name)
    B, S, Lr, vocab, V, wseed, bseed, steps, it0 = (int(v) for v in z["meta"])
    eps_e, eps_c = torch.from_numpy(z["eps_e_0"]), torch.from_numpy(z["eps_c_0"])
    res = {}
    for flag in (False, True):
        model, P = build(cfg, opt, wseed, train_dropout=True)
        model.train()
        model.cls_only_last = flag
        model.set_noise(eps_e, eps_c)
        loss = model(*call(model, batch, it0))
        loss.backward()
        torch.cuda.synchronize()
        res[flag] = ({k: float(v) for k, v in model.last_terms().items()}, model._last_call.buf.lat.clone(),
                     {k: p.grad.detach().clone() for k, p in model.named_parameters()})
    (t0, lat0, g0), (t1, lat1, g1) = res[False], res[True]
    for k in t0:
        assert abs(t0[k] - t1[k]) <= 1e-3 * max(abs(t0[k]), 1e-3), (k, t0[k], t1[k])
    assert relnorm(lat1, lat0) < 2e-3
    worst = {k: relnorm(g1[k], g0[k]) for k in g0 if float(g0[k].norm()) > 1e-6 and not k.endswith("key.bias")}
    bad = {k: v for k, v in worst.items() if v > 2e-2}
    assert not bad, bad


@pytest.mark.parametrize("name,varlen,cls_only", [("zh_small", False, False), ("zh_ragged", True, True), ("zh_full12", True, True)])
def test_wgrad_side_stream_is_bitwise_identical(golden_dir, name, varlen, cls_only):
    """Weight-gradient GEMMs forked onto the second stream: same kernels, same summation order -> identical bits.
    Three forward/backward passes back to back (a missing event dependency would show up as a stale or torn operand).
    Only the embedding tables are exempt from torch.equal: their backward scatters with float atomics."""
    cfg, opt = CASES[name]
    opt = O.Opt(**{**vars(opt), "dropout": 0.5})
    z, batch = load(golden_dir, name)
    B, S, Lr, vocab, V, wseed, bseed, steps, it0 = (int(v) for v in z["meta"])
    res = {}
    for flag in (False, True):
        model, P = build(cfg, opt, wseed, train_dropout=True)
        model.train()
        model.overlap_wgrad, model.varlen, model.cls_only_last = flag, varlen, cls_only
        out = []
        for s in range(3):
            model.set_noise(torch.from_numpy(z["eps_e_0"]), torch.from_numpy(z["eps_c_0"]))
            loss = model(*call(model, batch, it0 + s))
            for p in model.parameters():
                p.grad = None
            loss.backward()
            out.append((float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters()}))
        torch.cuda.synchronize()
        res[flag] = out
    for (l0, g0), (l1, g1) in zip(res[False], res[True]):
        assert l0 == l1
        for k in g0:
            if "embeddings.word" in k or "embeddings.position" in k or "embeddings.token_type" in k:
                assert relnorm(g1[k], g0[k]) < 1e-5, k
            else:
                assert torch.equal(g0[k], g1[k]), k


@pytest.mark.parametrize("name", ["zh_small", "zh_allneg"])
def test_adam_fused_into_backward_equals_plain_step(golden_dir, name):
    """FusedAdam(fuse_into_backward=True) applies each encoder layer's update on the auxiliary stream while the backward
    pass is still running; the result must equal the plain zero_grad / backward / step one.  After the first step (no
    history) everything but the atomically scattered embedding tables must be bit-identical; after the second step
    (moments and step count carried over) the runs may differ by the rounding noise of that scatter only.  Longer
    trajectories are not comparable: maximising the MMD makes the dynamics expansive, and two runs of the SAME mode
    drift apart by 1e-4 in the loss within four steps."""
    cfg, opt = CASES[name]
    opt = O.Opt(**{**vars(opt), "dropout": 0.3})
    z, batch = load(golden_dir, name)
    B, S, Lr, vocab, V, wseed, bseed, steps, it0 = (int(v) for v in z["meta"])
    res = {}
    for fused in (False, True):
        model, P = build(cfg, opt, wseed, train_dropout=True)
        model.train()
        optim = M.FusedAdam(model, lr=1e-5, fuse_into_backward=fused)
        losses, snaps = [], []
        for s in range(2):
            model.set_noise(torch.from_numpy(z["eps_e_0"]), torch.from_numpy(z["eps_c_0"]))
            loss = model(*call(model, batch, it0 + s))
            optim.zero_grad()
            loss.backward()
            optim.step()
            losses.append(float(loss.detach()))
            torch.cuda.synchronize()
            snaps.append({k: p.detach().clone() for k, p in model.named_parameters()})
        assert optim._done == [] and optim.step_count == 2
        res[fused] = (losses, snaps, optim.exp_avg.clone(), optim.exp_avg_sq.clone())
    noisy = ("embeddings.word", "embeddings.position", "embeddings.token_type")
    moved = 0
    for k, w in res[False][1][0].items():
        if not any(n in k for n in noisy):
            assert torch.equal(w, res[True][1][0][k]), k
        moved += int(not torch.equal(w, P[k].cuda()))
    assert moved >= len(res[False][1][0]) - 12         # everything optimised did move (latent heads / dead pair head stay)
    for a, b in zip(res[False][0], res[True][0]):
        assert abs(a - b) <= 1e-5 * max(abs(a), 1.0), (res[False][0], res[True][0])
    for k in res[False][1][1]:
        assert relnorm(res[True][1][1][k], res[False][1][1][k]) <= 1e-6, k
    assert relnorm(res[True][2], res[False][2]) < 1e-4 and relnorm(res[True][3], res[False][3]) < 1e-4


@pytest.mark.parametrize("nb,varlen", [(2, False), (3, True), (5, False)])
def test_gradients_small_batches(golden_dir, nb, varlen):
    """Two to five pairs: the weight gradients run as ONE K slice written in place (no slabs), the batch is padded to a
    multiple of 128 rows, the forward stays a single chain (odd batch) -- against the oracle's autograd."""
    cfg, opt = CASES["zh_ragged"]
    z, batch = load(golden_dir, "zh_ragged")
    batch = {k: v[:nb].clone() for k, v in batch.items()}
    batch["labels"][0], batch["cau_labels"][0] = 1.0, 1.0          # at least one positive pair
    B, S, Lr, vocab, V, wseed, bseed, steps, it0 = (int(v) for v in z["meta"])
    model, P = build(cfg, opt, wseed)
    model.train()
    model.varlen = varlen
    eps_e, eps_c = torch.from_numpy(z["eps_e_0"]), torch.from_numpy(z["eps_c_0"])
    model.set_noise(eps_e, eps_c)
    loss = model(*call(model, batch, it0))
    loss.backward()
    torch.cuda.synchronize()
    out, grads = O.loss_and_grads(P, batch, it0, cfg, opt, eps_e, eps_c)
    terms = {k: float(v) for k, v in model.last_terms().items()}
    for k in TERMS:
        r = float(out[k])
        assert abs(terms[k] - r) <= TOL_TERM_BF16 * max(abs(r), 1e-3), (k, terms[k], r)
    named = dict(model.named_parameters())
    worst = {k: relnorm(named[k].grad, g) for k, g in grads.items() if g is not None and float(g.norm()) > 1e-7}
    bad = {k: v for k, v in worst.items() if v > 6e-2}
    assert not bad, bad
    assert np.median(list(worst.values())) < 2e-2


@pytest.mark.parametrize("shape", ["A", "B"])
def test_bench_shape_backward_and_adam_vs_oracle(shape):
    """model-level gradient and post-Adam parity at the bench's batch shape.  B = 64, S = 128, vocabulary 21 128,
    V = 23 771 (T = 8 192 rows dense / ~1.8 k packed rows ECPE-shaped); two layers are enough to put every production kernel inside
    the model -- 256-row ping-pong forward / data-gradient / weight-gradient tiles with their uneven split-K slabs and the slab
    reduction, attention backward at 768 workgroups, the [CLS]-only last layer, the side-stream weight gradients -- and keep the CPU
    oracle at ~20 s.  Every parameter gradient against the oracle's autograd (ref :841), then one fused Adam step against
    torch.optim.Adam's arithmetic (ref :842): same bounds as the small golden cases."""
    cfg, opt = O.EncoderConfig(layers=2), O.Opt(dropout=0.0)
    assert (cfg.vocab_size, opt.pair_bow_dim) == (21128, 23771)
    model, P = build(cfg, opt, 0)
    model.train()
    batch = O.synthetic_batch(64, 128, cfg, opt.pair_bow_dim, seed=1, shape=shape)
    g = torch.Generator().manual_seed(3)
    eps_e, eps_c = torch.randn(opt.ec_dim, generator=g), torch.randn(opt.ec_dim, generator=g)
    model.set_noise(eps_e, eps_c)
    optim = M.FusedAdam(model, lr=opt.vae_lr)
    loss = model(*call(model, batch, 3))
    optim.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    named = dict(model.named_parameters())
    got_grads = {k: p.grad.detach().clone() for k, p in named.items() if p.grad is not None}
    optim.step()
    torch.cuda.synchronize()
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    P1, out, grads = O.train_step({k: v.clone() for k, v in P.items()}, batch, 3, cfg, opt, O.AdamState(), eps_e, eps_c)
    # (with two layers the total is a near-cancellation of its weighted terms -- -0.65 against a scale of ~100 -- so it is held on the
    # terms' scale here, like every golden case; the 1e-3 on the total itself is asserted at 12 layers in the test above)
    scale = sum(abs(WEIGHTS[k] * float(out[k])) for k in TERMS)
    assert abs(float(loss) - float(out["loss"])) <= TOL_LOSS_OVER_SCALE * scale, (float(loss), float(out["loss"]), scale)
    worst = {}
    for k, gr in grads.items():
        assert k in got_grads, k
        if gr is None or float(gr.norm()) < 1e-7:
            continue
        worst[k] = relnorm(got_grads[k], gr)
    bad = {k: v for k, v in worst.items() if v > 4e-2}
    assert not bad, bad
    assert np.median(list(worst.values())) < 1.5e-2, np.median(list(worst.values()))
    # post-step weights: the first Adam step moves every element by lr * sign(g) (m / sqrt(v) = +-1 up to eps): elements whose
    # gradient is within bf16 noise of zero may move the other way, so demand lr-level agreement on >= 97 % and 2 lr everywhere
    sd = model.state_dict()
    opt_keys = set(O.optimised_keys(cfg, opt))
    for k, w1 in P1.items():
        d = (sd[k].detach().cpu() - w1).abs()
        if k not in opt_keys:                                   # quirk Q3: the four latent heads never move
            assert torch.equal(sd[k].detach().cpu(), P[k]), k
            continue
        assert float(d.max()) <= 2 * opt.vae_lr * 1.01, (k, float(d.max()))
        if not k.endswith("key.bias") and worst.get(k, 1.0) < 4e-2:
            # rows of the embedding tables that no token of the batch touches have an exactly zero gradient on both sides
            assert float((d <= 0.2 * opt.vae_lr).float().mean()) >= 0.90, (k, float((d <= 0.2 * opt.vae_lr).float().mean()))


@pytest.mark.parametrize("shape,variant", [("A", "zh"), ("B", "zh"), ("B", "roberta")])
def test_embedding_table_gradients_are_bit_reproducible(shape, variant):
    """Round 4: the word / position table gradients come from fixed-order segment sums over keys sorted in the forward pass
    (csrc/ln.hip: embed_sort_kernel, embed_segsum_kernel) instead of 6.3 M float atomics -- the library's last order-dependent sum.
    A small vocabulary (every id ~8 times per batch, [CLS]-like ids 64 times), dense and packed rows, BERT and RoBERTa position ids:
    three passes over the same batch give bit-identical gradients for EVERY parameter, with the side stream on and off; and the
    table gradients equal an fp64 index_add of the same rows to fp32 rounding (a wrong run boundary would drop or double a row)."""
    if variant == "roberta":
        cfg = O.EncoderConfig(layers=2, vocab_size=1000, max_pos=514, type_vocab=1, ln_eps=1e-5, variant="roberta", pad_id=1)
        opt = O.Opt(language="en", pair_bow_dim=257, dropout=0.0)
    else:
        cfg, opt = O.EncoderConfig(layers=2, vocab_size=1000), O.Opt(pair_bow_dim=257, dropout=0.0)
    model, P = build(cfg, opt, 7)
    model.train()
    batch = O.synthetic_batch(64, 128, cfg, opt.pair_bow_dim, seed=5, shape=shape)
    batch["input_ids"][:, 0] = 2                                      # one id in every sample: a 64-row run
    g = torch.Generator().manual_seed(3)
    eps_e, eps_c = torch.randn(opt.ec_dim, generator=g), torch.randn(opt.ec_dim, generator=g)
    runs = []
    for overlap in (True, True, False, True):
        model.overlap_wgrad = overlap
        model.set_noise(eps_e, eps_c)
        for p_ in model.parameters():
            p_.grad = None
        loss = model(*call(model, batch, 3))
        loss.backward()
        torch.cuda.synchronize()
        runs.append({k: p_.grad.detach().clone() for k, p_ in model.named_parameters() if p_.grad is not None})
    for r in runs[1:]:
        for k, v in runs[0].items():
            assert torch.equal(v, r[k]), k
    out, grads = O.loss_and_grads(P, batch, 3, cfg, opt, eps_e, eps_c)
    for k in ("encoder.embeddings.word_embeddings.weight", "encoder.embeddings.position_embeddings.weight"):
        assert relnorm(runs[0][k], grads[k]) < 4e-2, k
        touched = grads[k].abs().sum(1) > 0
        assert torch.equal(runs[0][k].cpu().abs().sum(1) > 0, touched), k          # exactly the rows the batch uses, no others


@pytest.mark.experiments
@pytest.mark.parametrize("cls_only", [True, False])
def test_split_k_epilogues_fused_into_the_layernorm_backward_same_bits(cls_only):
    """Round 4, packed ECPE batches: the slab epilogue of the split-K FFN1 / QKV data-gradient GEMMs is deferred into the LayerNorm backward
    that reads it, and the slab epilogue of the split-K out-projection / FFN2 forward GEMMs (bias + dropout + residual, stored or recomputed) into
    the LayerNorm FORWARD behind it (ln_fwd_slabs_kernel) -- hook 271, the default; 270 = their own launches as before.  Same expressions in the
    same order: the loss and EVERY gradient bit-identical,
    three layers (so that a middle layer both consumes the layer above's deferred rows and defers its own), with and without the [CLS]-only
    last layer, dropout on, side stream on and off."""
    lib = L.load()
    cfg, opt = O.EncoderConfig(layers=3, vocab_size=2000), O.Opt(pair_bow_dim=513, dropout=0.3)
    batch = O.synthetic_batch(64, 128, cfg, opt.pair_bow_dim, seed=9, shape="B")
    g = torch.Generator().manual_seed(3)
    eps_e, eps_c = torch.randn(opt.ec_dim, generator=g), torch.randn(opt.ec_dim, generator=g)
    res = {}
    for hook in (270, 271):
        for overlap in (False, True):
            L.check(lib.carel_gemm_set_variant(hook))
            try:
                model, P = build(cfg, opt, 5, train_dropout=True)
                model.train()
                model.overlap_wgrad, model.cls_only_last = overlap, cls_only
                model.set_noise(eps_e, eps_c)
                loss = model(*call(model, batch, 3))
                loss.backward()
                torch.cuda.synchronize()
                res[(hook, overlap)] = (loss.detach().clone(), {k: p_.grad.detach().clone() for k, p_ in model.named_parameters() if p_.grad is not None})
            finally:
                L.check(lib.carel_gemm_set_variant(271))
    base = res[(270, False)]
    for key, (loss, grads) in res.items():
        assert torch.equal(loss, base[0]), key
        for k, v in base[1].items():
            assert torch.equal(v, grads[k]), (key, k)


def _report(name, payload):
    """measured figures of the tight parity tests, for DESIGN.md (written beside the other GPU-box outputs when that directory exists)"""
    import json
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_%s.json" % name), "w") as f:
            json.dump(payload, f, indent=1, sort_keys=True)


@pytest.mark.parametrize("shape", ["A", "B"])
def test_bench_shape_gradients_vs_the_bf16_emulating_oracle(shape):
    """every parameter gradient at the bench's batch shape (B = 64, S = 128, vocabulary 21 128, V = 23 771, two
    layers: every production kernel incl. the grouped weight-gradient launch) against the oracle run with `quant=O.bf16_hip` -- operands
    rounded to bf16 where the kernels store bf16, straight-through, AND the gradient signals rounded where the backward pass stores them in
    bf16 (dyb / dyb2 / du / dctx / dS / dqkv, the saved bf16 gelu').  What is left is summation order and the points the emulation does
    not model (the attention kernels' internal bf16 probabilities in the backward pass), so the bound is ~5x tighter than against the fp32
    oracle (4e-2 above): a gradient scaled or shifted by a few per cent in ONE tensor fails here.  Ref :841."""
    cfg, opt = O.EncoderConfig(layers=2), O.Opt(dropout=0.0)
    model, P = build(cfg, opt, 0)
    model.train()
    batch = O.synthetic_batch(64, 128, cfg, opt.pair_bow_dim, seed=1, shape=shape)
    g = torch.Generator().manual_seed(3)
    eps_e, eps_c = torch.randn(opt.ec_dim, generator=g), torch.randn(opt.ec_dim, generator=g)
    model.set_noise(eps_e, eps_c)
    loss = model(*call(model, batch, 3))
    loss.backward()
    torch.cuda.synchronize()
    got = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None}
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    out, grads = O.loss_and_grads(P, batch, 3, cfg, opt, eps_e, eps_c, quant=O.bf16_hip)
    # (the key bias has an exactly zero true gradient -- softmax is invariant to a shift of every score of a query -- so what both sides hold is noise)
    worst = {k: relnorm(got[k], gr) for k, gr in grads.items() if gr is not None and float(gr.norm()) > 1e-7 and not k.endswith("key.bias")}
    med = float(np.median(list(worst.values())))
    # query / key projections: their gradient passes through dS = P * (dP - delta), a difference of nearly equal terms, with delta = rowsum(dO * O)
    # taken from the bf16-STORED context rows in the attention backward kernel (fp32 rows in the oracle): bf16 rounding, not modelled by the emulation
    qk = {k: v for k, v in worst.items() if ".attention.self.query." in k or ".attention.self.key." in k}
    rest = {k: v for k, v in worst.items() if k not in qk}
    _report("grad_bf16_emulating_oracle_" + shape, dict(worst_qk=max(qk.values()), worst_rest=max(rest.values()), worst_rest_key=max(rest, key=rest.get), median=med,
                                                       top_rest=sorted(rest.items(), key=lambda kv: -kv[1])[:8]))
    bad = {k: v for k, v in rest.items() if v > TOL_GRAD_BF16_EMU}
    bad.update({k: v for k, v in qk.items() if v > TOL_GRAD_BF16_EMU_QK})
    assert not bad, bad
    assert med < 0.6 * TOL_GRAD_BF16_EMU, med


def test_bench_shape_five_adam_steps_vs_oracle_trajectory():
    """five consecutive steps (forward, backward, fused Adam; a different batch and noise each) at the bench's
    batch shape against the fp32 oracle's trajectory (torch.optim.Adam's arithmetic, ref :840-842).  After the first step every update
    is lr * sign(g); from the second on m / sqrt(v) is no longer +-1, so a gradient whose SCALE is off moves the weights measurably:
    at step 5 at least 97 % of the elements of every tensor lie within 0.2 lr of the oracle's (measured: >= 99.2 %; total movement:
    up to 5 lr) and 99 % within 2 lr (measured: >= 99.4 %) -- an element whose gradient is within bf16 noise of zero may step +-lr the other way every
    time.  The key bias is excluded: its true gradient is exactly zero (softmax shift invariance), both sides step on rounding noise."""
    cfg, opt = O.EncoderConfig(layers=2), O.Opt(dropout=0.0)
    model, P = build(cfg, opt, 0)
    model.train()
    optim = M.FusedAdam(model, lr=opt.vae_lr)
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    Pc, st = {k: v.clone() for k, v in P.items()}, O.AdamState()
    g = torch.Generator().manual_seed(11)
    for step in range(5):
        batch = O.synthetic_batch(64, 128, cfg, opt.pair_bow_dim, seed=31 + step, shape="A" if step % 2 == 0 else "B")
        eps_e, eps_c = torch.randn(opt.ec_dim, generator=g), torch.randn(opt.ec_dim, generator=g)
        model.set_noise(eps_e, eps_c)
        loss = model(*call(model, batch, step))
        optim.zero_grad()
        loss.backward()
        optim.step()
        Pc, out, grads = O.train_step(Pc, batch, step, cfg, opt, st, eps_e, eps_c)
        scale = sum(abs(WEIGHTS[k] * float(out[k])) for k in TERMS)
        assert abs(float(loss) - float(out["loss"])) <= TOL_LOSS_OVER_SCALE * scale, (step, float(loss), float(out["loss"]))
    torch.cuda.synchronize()
    sd = model.state_dict()
    lr = opt.vae_lr
    opt_keys = set(O.optimised_keys(cfg, opt))
    frac, far, frac2 = {}, {}, {}
    for k, w in Pc.items():
        if k not in opt_keys:
            assert torch.equal(sd[k].detach().cpu(), P[k]), k            # quirk Q3
            continue
        d = (sd[k].detach().cpu() - w).abs()
        moved = (w - P[k]).abs() > 0                                     # (embedding rows no batch touched never move, on either side)
        if int(moved.sum()) == 0:
            continue
        if k.endswith("key.bias"):
            continue
        frac[k] = float((d[moved] <= 0.2 * lr).float().mean())
        frac2[k] = float((d[moved] <= 2.0 * lr).float().mean())
        far[k] = float(d.max()) / lr
    _report("five_adam_steps", dict(min_fraction_within_0p2_lr=min(frac.values()), min_key=min(frac, key=frac.get), max_distance_in_lr=max(far.values()),
                                    min_fraction_within_2_lr=min(frac2.values()), lowest=sorted(frac.items(), key=lambda kv: kv[1])[:8]))
    low = {k: v for k, v in frac.items() if v < 0.97}
    assert not low, low
    low2 = {k: v for k, v in frac2.items() if v < 0.99}
    assert not low2, low2


@pytest.mark.experiments
@pytest.mark.parametrize("name,varlen,cls_only", [("zh_small", False, False), ("zh_ragged", True, True), ("zh_full12", True, True), ("zh_full12", False, False)])
def test_layernorm_residual_recomputed_in_the_next_epilogue_is_bitwise_identical(golden_dir, name, varlen, cls_only):
    """The encoder's LayerNorms no longer write their f32 output (hook 231, default): the out-projection / FFN2 epilogue that adds it as the
    residual recomputes it from the pre-LayerNorm rows and the saved statistics with the LayerNorm kernel's own expression.  Against hook
    230 (f32 rows written and read back): the loss and every gradient bit-identical, dense and packed, with and without the [CLS]-only
    last layer (whose gathered residual rows and the encoder's final output still come from stored rows), dropout on."""
    lib = L.load()
    cfg, opt = CASES[name]
    opt = O.Opt(**{**vars(opt), "dropout": 0.3})
    z, batch = load(golden_dir, name)
    B, S, Lr, vocab, V, wseed, bseed, steps, it0 = (int(v) for v in z["meta"])
    res = {}
    for hook in (230, 231):
        L.check(lib.carel_gemm_set_variant(hook))
        try:
            model, P = build(cfg, opt, wseed, train_dropout=True)
            model.train()
            model.overlap_wgrad, model.varlen, model.cls_only_last = False, varlen, cls_only
            out = []
            for s in range(2):
                model.set_noise(torch.from_numpy(z["eps_e_0"]), torch.from_numpy(z["eps_c_0"]))
                loss = model(*call(model, batch, it0 + s))
                for p in model.parameters():
                    p.grad = None
                loss.backward()
                out.append((float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters()}))
            torch.cuda.synchronize()
            res[hook] = out
        finally:
            L.check(lib.carel_gemm_set_variant(231))
    for (l0, g0), (l1, g1) in zip(res[230], res[231]):
        assert l0 == l1
        for k in g0:
            if "embeddings.word" in k or "embeddings.position" in k or "embeddings.token_type" in k:
                assert relnorm(g1[k], g0[k]) < 1e-5, k
            else:
                assert torch.equal(g0[k], g1[k]), k
