"""Pins the CPU oracle (oracle/carel_oracle.py) to golden vectors produced by executing the
reference's own DrlClassifier / MMDStatistic / pdist / HSIC code (tests/golden/gen_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import carel_oracle as O

CASES = {
    "zh_small": (O.EncoderConfig(layers=2, vocab_size=1000), O.Opt(pair_bow_dim=257, dropout=0.0)),
    "zh_negmid": (O.EncoderConfig(layers=1, vocab_size=500), O.Opt(pair_bow_dim=130, dropout=0.0)),   # 4 steps, step 1 all-negative
    "zh_ragged": (O.EncoderConfig(layers=2, vocab_size=1000), O.Opt(pair_bow_dim=513, dropout=0.0)),
    "zh_allneg": (O.EncoderConfig(layers=1, vocab_size=500), O.Opt(pair_bow_dim=130, dropout=0.0)),
    "zh_s64": (O.EncoderConfig(layers=2, vocab_size=800), O.Opt(pair_bow_dim=300, dropout=0.0)),
    "en_small": (O.EncoderConfig(layers=2, vocab_size=1200, max_pos=514, type_vocab=1, ln_eps=1e-5,
                                 variant="roberta", pad_id=1), O.Opt(language="en", pair_bow_dim=257, dropout=0.0)),
    "zh_full12": (O.EncoderConfig(), O.Opt(pair_bow_dim=1000, dropout=0.0)),
}


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)
    batch = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    return z, batch


def gslice(t, n=64):
    f = t.reshape(-1)
    step = max(1, f.numel() // n)
    return torch.cat((f[:n], f[-n:], f[::step][:n])).numpy()


@pytest.mark.parametrize("name", list(CASES))
def test_forward_terms_and_training_steps(golden_dir, name):
    cfg, opt = CASES[name]
    z, batch = load(golden_dir, name)
    B, S, L, vocab, V, wseed, bseed, steps, it0 = (int(v) for v in z["meta"])
    # the batch is regenerable from its seed: the fixture inputs are what synthetic_batch makes
    regen = O.synthetic_batch(B, S, cfg, V, seed=bseed, shape=str(z["shape"]))
    assert torch.equal(regen["input_ids"], batch["input_ids"])
    P = O.init_params(cfg, opt, seed=wseed)
    st = O.AdamState()
    losses = []
    neg = set(z["neg_steps"].tolist()) if "neg_steps" in z.files else set()
    batch_neg = dict(batch, labels=torch.zeros_like(batch["labels"]), cau_labels=torch.zeros_like(batch["cau_labels"]))
    for s in range(steps):
        eps_e, eps_c = torch.from_numpy(z[f"eps_e_{s}"]), torch.from_numpy(z[f"eps_c_{s}"])
        P, out, grads = O.train_step(P, batch_neg if s in neg else batch, it0 + s, cfg, opt, st, eps_e, eps_c)
        losses.append(float(out["loss"]))
        if s == 0:
            np.testing.assert_allclose(out["pooled"].numpy(), z["pooled"], atol=2e-5, rtol=1e-4)
            for k in ("mu_e", "lv_e", "mu_c", "lv_c"):
                np.testing.assert_allclose(out[k].numpy(), z[k], atol=2e-5, rtol=1e-4)
            for k in ("mmd", "emo", "cau", "pair", "kl_e", "kl_c", "rec"):
                np.testing.assert_allclose(float(out[k]), float(z["t_" + k]), rtol=2e-5, atol=1e-7, err_msg=k)
            for k in z.files:
                if k.startswith("g_"):
                    pk = k[2:]
                    ref = z[k]
                    scale = max(float(z["gn_" + pk]), 1e-12)
                    got = gslice(grads[pk]) if grads[pk] is not None else np.zeros_like(ref)
                    # quirk Q3: latent heads DO receive gradients
                    assert np.abs(got - ref).max() <= 2e-4 * scale + 1e-7, (pk, np.abs(got - ref).max(), scale)
    np.testing.assert_allclose(losses, z["losses"], rtol=2e-4, atol=2e-4)
    # post-Adam weights: optimised tensors moved like the reference; latent heads did not move (Q3)
    P0 = O.init_params(cfg, opt, seed=wseed)
    for k in z.files:
        if k.startswith("w_"):
            pk = k[2:]
            got = gslice(P[pk])
            if pk.endswith("key.bias"):
                # d(loss)/d(key bias) is analytically 0 (softmax is shift-invariant per query row); Adam's
                # m/sqrt(v) normalisation turns the fp32 rounding residue into +-lr steps of random sign
                np.testing.assert_allclose(got, z[k], atol=steps * opt.vae_lr * 1.01, rtol=0, err_msg=pk)
                continue
            np.testing.assert_allclose(got, z[k], atol=3e-6, rtol=0, err_msg=pk)
            moved = not torch.equal(P[pk], P0[pk])
            dead_pair = name == "zh_allneg" and pk.startswith("pair_classifier")   # loss term replaced by int 0
            assert moved == (not pk.startswith(O.UNOPTIMISED_PREFIXES) and not dead_pair), pk
    # eval-mode predictions (get_pair_preds :265-282)
    prob = O.pair_preds(P, batch["input_ids"], batch["attention_masks"], batch["token_type_ids"], cfg, opt,
                        torch.from_numpy(z["pred_eps_e"]), torch.from_numpy(z["pred_eps_c"]))
    far = (prob - 0.5).abs().squeeze(1) > 1e-4
    assert torch.equal(prob.round().squeeze(1)[far], torch.from_numpy(z["preds"]).squeeze(1)[far])


def test_mmd_pdist_hsic_statistics(golden_dir):
    z = np.load(os.path.join(golden_dir, "statistics.npz"), allow_pickle=False)
    for tag in "abcde":
        s1, s2 = torch.from_numpy(z[f"{tag}_s1"]), torch.from_numpy(z[f"{tag}_s2"])
        mmd, kern = O.mmd_statistic(s1, s2, [0.1], ret_matrix=True)
        np.testing.assert_allclose(float(mmd), float(z[f"{tag}_mmd"]), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(float(O.mmd_statistic(s1, s2, [0.1, 0.5, 2.0])), float(z[f"{tag}_mmd3"]),
                                   rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(gslice(kern, 32), z[f"{tag}_kern_slice"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(gslice(O.pdist(s1, s2), 32), z[f"{tag}_pdist_slice"], rtol=1e-5, atol=1e-6)
        a, b = s1.clone().requires_grad_(True), s2.clone().requires_grad_(True)
        (-O.mmd_statistic(a, b, [0.1])).backward()
        np.testing.assert_allclose(a.grad.numpy(), z[f"{tag}_g1"], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(b.grad.numpy(), z[f"{tag}_g2"], rtol=1e-4, atol=1e-7)
    for tag in "ab":
        x, y = torch.from_numpy(z[f"h{tag}_x"]), torch.from_numpy(z[f"h{tag}_y"])
        np.testing.assert_allclose(float(O.hsic_statistic(x, y)), float(z[f"h{tag}_hsic"]), rtol=1e-4, atol=1e-7)


def test_dropout_mask_statistics_and_shard_consistency():
    p = 0.1
    m = O.dropout_scale_mask(7, O.site_attn_out(3), (64, 128, 768), p)
    keep = (m > 0).float().mean().item()
    assert abs(keep - (1 - p)) < 2e-3
    assert abs(m.mean().item() - 1.0) < 3e-3
    # a DP shard (rows 16..31) sees exactly the unsharded batch's mask rows
    sh = O.dropout_scale_mask(7, O.site_attn_out(3), (16, 128, 768), p, row_offset=16)
    assert torch.equal(sh, m[16:32])
    # different sites / seeds decorrelate
    m2 = O.dropout_scale_mask(7, O.site_ffn_out(3), (64, 128, 768), p)
    agree = ((m > 0) == (m2 > 0)).float().mean().item()
    assert abs(agree - (0.81 + 0.01)) < 5e-3


def test_kl_anneal_matches_reference_formula():
    opt = O.Opt()
    # (tanh((it - 30000)/6666.67) + 1) * 0.03, host double (:515-523)
    assert abs(O.kl_anneal_weight(0, opt) - 7.4e-6) < 1e-6
    assert O.kl_anneal_weight(10, opt) > O.kl_anneal_weight(0, opt)


def test_vi_club_two_phase_steps(golden_dir):
    """VI ablation (drl_classifier_ec_vi.py): approximation-net loss, CLUB bound, beta ramp, two optimisers -- against
    the reference class run by tests/golden/gen_golden_vi.py."""
    cfg, opt = O.EncoderConfig(layers=2, vocab_size=900), O.Opt(pair_bow_dim=211, dropout=0.0, e_num_class=1)
    z, batch = load(golden_dir, "vi_small")
    B, S, L, vocab, V, wseed, bseed, steps = (int(v) for v in z["meta"])
    P = {**O.init_params(cfg, opt, seed=wseed), **O.init_vi_params(opt, seed=wseed + 1)}
    st_vae, st_aprx = O.AdamState(), O.AdamState()
    for s in range(steps):
        eps_e, eps_c = torch.from_numpy(z[f"eps_e_{s}"]), torch.from_numpy(z[f"eps_c_{s}"])
        perm = torch.from_numpy(z[f"perm_{s}"])
        P0 = P
        P, out = O.vi_train_step(P, batch, 5 + s, int(z["epochs"][s]), cfg, opt, eps_e, eps_c, perm, st_vae, st_aprx)
        np.testing.assert_allclose(out["z_e"].numpy(), z[f"z_e_{s}"], atol=3e-5, rtol=1e-4)
        np.testing.assert_allclose(out["z_c"].numpy(), z[f"z_c_{s}"], atol=3e-5, rtol=1e-4)
        for k in ("aprx", "vae", "upper", "total"):
            np.testing.assert_allclose(float(out[k]), float(z[f"{k}_{s}"]), rtol=3e-5, atol=2e-6, err_msg=f"{k} step {s}")
        if s == 0:
            leaf = {k: P0[k].clone().requires_grad_(True) for k in O.VI_KEYS}
            O.vi_aprx_loss(leaf, out["z_e"], out["z_c"]).backward()
            for k in O.VI_KEYS:
                np.testing.assert_allclose(leaf[k].grad.numpy(), z["ga_" + k], rtol=1e-4, atol=1e-5, err_msg=k)
        if s == 1:
            ze, zc = out["z_e"].clone().requires_grad_(True), out["z_c"].clone().requires_grad_(True)
            O.vi_upper_loss({k: P[k] for k in O.VI_KEYS}, ze, zc, perm).backward()
            np.testing.assert_allclose(ze.grad.numpy(), z["dz_e_1"], rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(zc.grad.numpy(), z["dz_c_1"], rtol=1e-4, atol=1e-6)
    lr_a, lr_v = opt.aprx_lr, opt.vae_lr
    for k in z.files:
        if k.startswith("w_"):
            pk = k[2:]
            got = P[pk].numpy() if pk in O.VI_KEYS else gslice(P[pk])
            np.testing.assert_allclose(got, z[k], atol=(0.6 * lr_a if pk in O.VI_KEYS else 0.6 * lr_v), rtol=0, err_msg=pk)


def test_en_adversarial_three_space_steps(golden_dir):
    """Config 4 (drl_classifier_en.py): three latent spaces, five discriminators, six optimisers -- the seven losses,
    the gradients every optimiser sees and the weights after three steps, against the reference class run by
    tests/golden/gen_golden_en_adv.py."""
    from oracle import carel_oracle_en as OE
    cfg = O.EncoderConfig(layers=2, vocab_size=900, max_pos=514, type_vocab=1, ln_eps=1e-5, variant="roberta", pad_id=1)
    opt = OE.OptEn(pair_bow_dim=211, dropout=0.0)
    z, batch = load(golden_dir, "en_adv_small")
    B, S, L, vocab, V, wseed, bseed, steps = (int(v) for v in z["meta"])
    P = OE.init_params(cfg, opt, seed=wseed)
    states = [O.AdamState() for _ in range(6)]
    for s in range(steps):
        eps = dict(con=torch.from_numpy(z[f"eps_con_{s}"]), e=torch.from_numpy(z[f"eps_e_{s}"]), c=torch.from_numpy(z[f"eps_c_{s}"]))
        P, out, grads = OE.train_step(P, batch, 7 + s, cfg, opt, states, eps)
        got = np.array([float(out[n]) for n in OE.LOSS_NAMES])
        np.testing.assert_allclose(got, z[f"losses_{s}"], rtol=3e-5, atol=2e-6, err_msg=f"losses step {s}")
        if s == 1:
            for k in z.files:
                if k.startswith("g_"):
                    ref = z[k]
                    np.testing.assert_allclose(gslice(grads[k[2:]]), ref, rtol=2e-3, atol=2e-6 + 2e-4 * float(np.abs(ref).max()), err_msg=k)
                    np.testing.assert_allclose(float(grads[k[2:]].norm()), float(z["gn_" + k[2:]]), rtol=1e-3, atol=1e-7, err_msg=k)  # key bias: exactly 0 in theory
    disc = tuple(g + "." for g in OE.DISC_GROUPS)
    for k in z.files:
        if k.startswith("w_"):
            lr = 10 * opt.adv_lr if k[2:].startswith(disc) else opt.vae_lr      # an RMSprop step is up to lr / sqrt(1 - alpha)
            np.testing.assert_allclose(gslice(P[k[2:]]), z[k], atol=0.6 * lr, rtol=0, err_msg=k)
    got = OE.pair_logits(P, batch["input_ids"], batch["attention_masks"], batch["token_type_ids"], cfg, opt,
                         torch.from_numpy(z["pp_eps_e"]), torch.from_numpy(z["pp_eps_c"]))
    np.testing.assert_allclose(got.numpy(), z["pp_logits"], rtol=1e-4, atol=1e-5)
