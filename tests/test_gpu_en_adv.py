"""Config 4: the three-space adversarial model of drl_classifier_en.py on the HIP path (carel_vae_amd.drl_classifier_en)
against (a) the golden vectors produced by the reference's own class (fp32 CPU, tests/golden/gen_golden_en_adv.py) and
(b) the CPU oracle (oracle/carel_oracle_en.py) run with bf16 rounding at the encoder kernels' storage points."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from carel_vae_amd import _lib as L
from carel_vae_amd import drl_classifier as M
from carel_vae_amd import drl_classifier_en as ME
from oracle import carel_oracle as O
from oracle import carel_oracle_en as OE

pytestmark = pytest.mark.gpu

CFG = O.EncoderConfig(layers=2, vocab_size=900, max_pos=514, type_vocab=1, ln_eps=1e-5, variant="roberta", pad_id=1)
DISC = tuple(g + "." for g in OE.DISC_GROUPS)


def build(opt, wseed, cfg=CFG, train_dropout=False):
    mcfg = M.encoder_config("en", vocab_size=cfg.vocab_size, max_pos=cfg.max_pos, type_vocab=cfg.type_vocab, ln_eps=cfg.ln_eps,
                            layers=cfg.layers, hidden_dropout=cfg.hidden_dropout if train_dropout else 0.0,
                            attn_dropout=cfg.attn_dropout if train_dropout else 0.0)
    model = ME.DrlClassifier(ME.make_opt(**{k: v for k, v in vars(opt).items() if k in ME.DEFAULT_OPT}), mcfg)
    P = OE.init_params(cfg, opt, seed=wseed)
    model.load_state_dict(P)
    model.to("cuda")
    return model, P


def load(golden_dir, name="en_adv_small"):
    z = np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)
    batch = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    return z, batch


def call(batch, it):
    b = {k: v.cuda() for k, v in batch.items()}
    return (b["input_ids"], b["attention_masks"], b["token_type_ids"], b["emo_labels"], b["cau_labels"], b["labels"], b["bow_reps"], it)


def eps_of(z, s):
    return dict(con=torch.from_numpy(z[f"eps_con_{s}"]), e=torch.from_numpy(z[f"eps_e_{s}"]), c=torch.from_numpy(z[f"eps_c_{s}"]))


def relnorm(a, b):
    a, b = a.double().cpu().flatten(), b.double().cpu().flatten()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def gslice(t, n=64):
    f = t.detach().cpu().reshape(-1)
    step = max(1, f.numel() // n)
    return torch.cat((f[:n], f[-n:], f[::step][:n])).numpy()


def reference_step(losses, opts):
    """The backward / zero_grad sequence of the reference's loop (drl_classifier_en.py:919-939)."""
    cd_e, cd_c, ed, ecd, cad, ced, vae = losses
    opts[0].zero_grad(); (cd_e + cd_c).backward(retain_graph=True)        # noqa: E702
    opts[1].zero_grad(); ed.backward(retain_graph=True)                  # noqa: E702
    opts[3].zero_grad(); ecd.backward(retain_graph=True)                 # noqa: E702
    opts[2].zero_grad(); cad.backward(retain_graph=True)                 # noqa: E702
    opts[4].zero_grad(); ced.backward(retain_graph=True)                 # noqa: E702
    opts[5].zero_grad(); vae.backward()                                  # noqa: E702


@pytest.mark.parametrize("M_,N_,K_", [(64, 23771, 432), (16, 211, 24), (5, 70, 33), (130, 64, 17)])
def test_sgemm_forms(M_, N_, K_):
    lib = L.load()
    g = torch.Generator().manual_seed(M_ + N_)
    A = torch.randn(M_, K_, generator=g).cuda()
    B = torch.randn(N_, K_, generator=g).cuda()
    bias = torch.randn(N_, generator=g).cuda()
    st = L.current_stream()
    Cc = torch.empty(M_, N_, device="cuda")
    L.check(lib.carel_sgemm_f32(A.data_ptr(), K_, 0, B.data_ptr(), K_, 0, Cc.data_ptr(), N_, M_, N_, K_, bias.data_ptr(), 0, 1, 0, st))
    ref = A.double() @ B.double().t() + bias.double()
    assert relnorm(Cc, ref) < 2e-6
    # weight-gradient form: C[N, K] = Cc^T A (reduction over the M rows), accumulated onto an existing image
    G = torch.ones(N_, K_, device="cuda")
    L.check(lib.carel_sgemm_f32(Cc.data_ptr(), N_, 1, A.data_ptr(), K_, 1, G.data_ptr(), K_, N_, K_, M_, None, 1, 1, 0, st))
    assert relnorm(G, Cc.double().t() @ A.double() + 1.0) < 2e-6
    # input-gradient form with the reduction split into slabs
    sp = 7
    parts = torch.full((sp, M_, K_), 7.0, device="cuda")
    L.check(lib.carel_sgemm_f32(Cc.data_ptr(), N_, 0, B.data_ptr(), K_, 1, parts.data_ptr(), K_, M_, K_, N_, None, 0, sp, M_ * K_, st))
    assert relnorm(parts.sum(0), Cc.double() @ B.double()) < 5e-6
    assert lib.carel_sgemm_f32(Cc.data_ptr(), N_, 0, B.data_ptr(), K_, 1, parts.data_ptr(), K_, M_, K_, N_, bias.data_ptr(), 0, sp, M_ * K_, st) != 0


def test_terms_and_gradients_vs_oracle_and_golden(golden_dir):
    opt = OE.OptEn(pair_bow_dim=211, dropout=0.0)
    z, batch = load(golden_dir)
    B, S, Lr, vocab, V, wseed, bseed, steps = (int(v) for v in z["meta"])
    model, P = build(opt, wseed)
    model.train()
    eps = eps_of(z, 0)
    model.set_noise(eps["con"], eps["e"], eps["c"])
    losses = model(*call(batch, 7))
    got = np.array([float(v.detach()) for v in losses])
    np.testing.assert_allclose(got, z["losses_0"], rtol=2e-2, atol=1e-3)                     # fp32 reference, bf16 encoder here
    ref, grads = OE.loss_and_grads(P, batch, 7, CFG, opt, eps, quant=O.bf16_round)
    terms = {k: float(v) for k, v in model.last_terms().items()}
    names = dict(zip(ME.TERM_NAMES, OE.LOSS_NAMES + ("cent_e", "cent_c", "ent_ed", "ent_cad", "ent_ec", "ent_ce", "emo_mul", "cau_mul", "con_mul",
                                                    "pair", "kl_e", "kl_c", "kl_con", "rec")))
    for mine, theirs in names.items():
        r = float(ref[theirs])
        assert abs(terms[mine] - r) <= 3e-3 * max(abs(r), 1e-3) + 1e-6, (mine, terms[mine], r)
    ft = model._last_call.buf
    D, Cd = opt.ec_dim, opt.con_dim
    assert relnorm(ft.lat[:, :Cd], ref["mu_con"]) < 1e-2 and relnorm(ft.lat[:, 2 * Cd:2 * Cd + D], ref["mu_e"]) < 1e-2
    assert relnorm(ft.z, ref["z"]) < 1e-2
    opts = [torch.optim.RMSprop(g, lr=opt.adv_lr) for g in model.get_params()[:5]] + [torch.optim.Adam(model.get_params()[5], lr=opt.vae_lr)]
    reference_step(losses, opts)
    torch.cuda.synchronize()
    named = dict(model.named_parameters())
    worst = {}
    for k, g in grads.items():
        got = named[k].grad
        assert got is not None, k
        if float(g.norm()) < 1e-7 or k.endswith("key.bias"):      # key bias: analytically zero (softmax shift invariance)
            continue
        if g.numel() == 1:           # one-logit bias: a signed mean over the batch that nearly cancels; compare on the scale of its terms
            assert abs(float(got) - float(g)) <= 4e-2 * abs(float(g)) + 2e-2, (k, float(got), float(g))
            continue
        worst[k] = relnorm(got, g)
    bad = {k: v for k, v in worst.items() if v > (1.5e-2 if not k.startswith("encoder.") else 4e-2)}
    assert not bad, bad
    for h in OE.LATENT_HEADS:                         # in no optimiser group (:357-376): no gradient is produced
        assert named[h + ".weight"].grad is None


def test_upstream_gradients_scale_each_share(golden_dir):
    """backward() of an arbitrary combination of the returned losses: every discriminator share is linear in its own
    upstream gradient and the shares add up in .grad."""
    opt = OE.OptEn(pair_bow_dim=211, dropout=0.0)
    z, batch = load(golden_dir)
    wseed = int(z["meta"][5])
    model, P = build(opt, wseed)
    model.train()
    eps = eps_of(z, 0)
    ref_grads = {}
    for combo in ("unit_e", "unit_c", "mix"):
        for p in model.parameters():
            p.grad = None
        model.set_noise(eps["con"], eps["e"], eps["c"])
        losses = model(*call(batch, 7))
        if combo == "unit_e":
            losses[0].backward()
        elif combo == "unit_c":
            losses[1].backward()
        else:
            (2.0 * losses[0] - 3.0 * losses[1] + 0.5 * losses[3]).backward()
        ref_grads[combo] = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in model.named_parameters()}
    w = "content_disc.weight"
    want = 2.0 * ref_grads["unit_e"][w] - 3.0 * ref_grads["unit_c"][w]
    assert relnorm(ref_grads["mix"][w], want) < 1e-6
    assert ref_grads["mix"]["ec_disc.weight"] is not None and ref_grads["mix"]["emotion_disc.weight"] is None
    assert ref_grads["mix"]["decoder.weight"] is None and ref_grads["unit_e"]["ec_disc.weight"] is None


@pytest.mark.parametrize("fused", [False, True])
def test_three_steps_follow_the_reference(golden_dir, fused):
    opt = OE.OptEn(pair_bow_dim=211, dropout=0.0)
    z, batch = load(golden_dir)
    B, S, Lr, vocab, V, wseed, bseed, steps = (int(v) for v in z["meta"])
    model, P = build(opt, wseed)
    model.train()
    if fused:
        opts = model.make_fused_optimizers(fuse_into_backward=True)
    else:
        gp = model.get_params()
        opts = [torch.optim.RMSprop(g, lr=opt.adv_lr) for g in gp[:5]] + [torch.optim.Adam(gp[5], lr=opt.vae_lr)]
    for s in range(steps):
        eps = eps_of(z, s)
        model.set_noise(eps["con"], eps["e"], eps["c"])
        losses = model(*call(batch, 7 + s))
        reference_step(losses, opts)
        for o in opts:
            o.step()
        got = np.array([float(v.detach()) for v in losses])
        np.testing.assert_allclose(got, z[f"losses_{s}"], rtol=2e-2, atol=2e-3, err_msg=f"step {s}")
    sd = model.state_dict()
    for k in z.files:
        if k.startswith("w_"):
            pk = k[2:]
            lr = 10 * opt.adv_lr if pk.startswith(DISC) else opt.vae_lr       # an RMSprop step is up to lr / sqrt(1 - alpha)
            d = np.abs(gslice(sd[pk]) - z[k])
            assert d.max() <= 2 * steps * lr * 1.01, pk
            if not pk.endswith("key.bias"):
                assert (d <= 1.2 * lr).mean() >= 0.95, (pk, float((d <= 1.2 * lr).mean()))
    P0 = OE.init_params(CFG, opt, seed=wseed)
    for n in ("content_mu.weight", "emotion_log_var.bias"):          # latent heads never move
        assert torch.equal(sd[n].cpu(), P0[n])


def test_dropout_masks_match_the_oracle(golden_dir):
    """Dropout ON (p = 0.5 on the ten head inputs, 0.1 inside the encoder): same counter-based masks in the oracle."""
    opt = OE.OptEn(pair_bow_dim=211, dropout=0.5)
    z, batch = load(golden_dir)
    wseed = int(z["meta"][5])
    model, P = build(opt, wseed, train_dropout=True)
    model.train()
    eps = eps_of(z, 1)
    model.set_noise(eps["con"], eps["e"], eps["c"])
    losses = model(*call(batch, 8))
    seed = model._last_call.seed
    ref, grads = OE.loss_and_grads(P, batch, 8, CFG, opt, eps, train=True, seed=seed, quant=O.bf16_round)
    got = np.array([float(v.detach()) for v in losses])
    want = np.array([float(ref[n]) for n in OE.LOSS_NAMES])
    np.testing.assert_allclose(got, want, rtol=4e-3, atol=1e-5)
    gp = model.get_params()
    opts = [torch.optim.RMSprop(g, lr=opt.adv_lr) for g in gp[:5]] + [torch.optim.Adam(gp[5], lr=opt.vae_lr)]
    reference_step(losses, opts)
    named = dict(model.named_parameters())
    for k in ("content_disc.weight", "ec_disc.weight", "emotion_disc.bias", "content_classifier.weight", "decoder.weight", "pair_classifier.weight",
              "emotion_classifier.weight", "encoder.pooler.dense.weight", "encoder.encoder.layer.1.output.dense.weight"):
        assert relnorm(named[k].grad, grads[k]) < 3e-2, (k, relnorm(named[k].grad, grads[k]))


def test_get_pair_preds_and_eval_forward(golden_dir):
    opt = OE.OptEn(pair_bow_dim=211, dropout=0.0)
    z, batch = load(golden_dir)
    wseed = int(z["meta"][5])
    model, P = build(opt, wseed)
    model.eval()
    b = {k: v.cuda() for k, v in batch.items()}
    model.set_noise(torch.zeros(opt.con_dim), torch.from_numpy(z["pp_eps_e"]), torch.from_numpy(z["pp_eps_c"]))
    got = model.get_pair_preds(b["input_ids"], b["attention_masks"], b["token_type_ids"])
    assert got.shape == (b["input_ids"].shape[0], 1)
    want = OE.pair_logits(P, batch["input_ids"], batch["attention_masks"], batch["token_type_ids"], CFG, opt, torch.from_numpy(z["pp_eps_e"]),
                          torch.from_numpy(z["pp_eps_c"]), quant=O.bf16_round)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=5e-3, atol=2e-3)
    with torch.no_grad():
        eps = eps_of(z, 0)
        model.set_noise(eps["con"], eps["e"], eps["c"])
        losses = model(*call(batch, 7))
    assert len(losses) == 7 and not losses[6].requires_grad
    np.testing.assert_allclose(np.array([float(v) for v in losses]), z["losses_0"], rtol=2e-2, atol=1e-3)
    with pytest.raises(L.CarelError):
        model.cpu()(*[v.cpu() if torch.is_tensor(v) else v for v in call(batch, 7)])


def test_full_size_heads_finite_and_consistent():
    """BASELINE size of the vocabulary-wide heads (V = 23 771, con_dim 384, batch 64) on a 1-layer encoder: losses finite,
    and the decoder / content-classifier gradients agree with torch autograd on the same logits (fp32 reference of the op)."""
    cfg = O.EncoderConfig(layers=1, vocab_size=600, max_pos=514, type_vocab=1, ln_eps=1e-5, variant="roberta", pad_id=1)
    opt = OE.OptEn(dropout=0.0)
    model, P = build(opt, 5, cfg=cfg)
    model.train()
    batch = OE.synthetic_batch(64, 128, cfg, opt.pair_bow_dim, seed=3, shape="B")
    g = torch.Generator().manual_seed(9)
    eps = dict(con=torch.randn(opt.con_dim, generator=g), e=torch.randn(opt.ec_dim, generator=g), c=torch.randn(opt.ec_dim, generator=g))
    model.set_noise(eps["con"], eps["e"], eps["c"])
    losses = model(*call(batch, 3))
    assert all(bool(torch.isfinite(v)) for v in losses)
    buf = model._last_call.buf
    zc = buf.z.detach().clone().requires_grad_(True)
    W = model.decoder.weight.detach().clone().requires_grad_(True)
    bvec = model.decoder.bias.detach().clone().requires_grad_(True)
    bow = batch["bow_reps"].cuda()
    rec = OE.O.bce_prob(torch.softmax(zc @ W.t() + bvec, dim=1), bow * 0.9 + 0.1 / opt.pair_bow_dim).mean()
    rec.backward()
    assert abs(float(rec) - float(model.last_terms()["rec"])) < 1e-5 * max(1.0, abs(float(rec)))
    gp = model.get_params()
    opts = [torch.optim.RMSprop(g_, lr=opt.adv_lr) for g_ in gp[:5]] + [torch.optim.Adam(gp[5], lr=opt.vae_lr)]
    reference_step(losses, opts)
    assert relnorm(model.decoder.weight.grad, W.grad) < 1e-4
    assert relnorm(model.decoder.bias.grad, bvec.grad) < 1e-4


@pytest.mark.parametrize("nb,S,fast", [(3, 64, True), (1, 32, True), (7, 128, True), (5, 96, False)])
def test_small_and_odd_batches(golden_dir, nb, S, fast):
    """Ragged sizes: odd batch, a single sample, shorter sequences (padding of the batch to a tile multiple must not leak into
    any term or gradient)."""
    opt = OE.OptEn(pair_bow_dim=211, dropout=0.0)
    z, batch = load(golden_dir)
    wseed = int(z["meta"][5])
    model, P = build(opt, wseed)
    model.train()
    if not fast:                 # every speed switch off: padded positions go through the encoder, the last layer runs on all rows,
        model.varlen = model.cls_only_last = model.overlap_wgrad = False         # one stream -- same results by construction
    sub = {k: (v[:nb, :S] if v.dim() == 2 and v.shape[1] == 128 else v[:nb]).contiguous() for k, v in batch.items()}
    sub["attention_masks"][:, 0] = 1
    sub["labels"][0] = 1.0                                   # at least one positive pair (pos_weight divides by their number, :599)
    eps = eps_of(z, 2)
    model.set_noise(eps["con"], eps["e"], eps["c"])
    losses = model(*call(sub, 5))
    ref, grads = OE.loss_and_grads(P, sub, 5, CFG, opt, eps, quant=O.bf16_round)
    got = np.array([float(v.detach()) for v in losses])
    want = np.array([float(ref[n]) for n in OE.LOSS_NAMES])
    np.testing.assert_allclose(got, want, rtol=4e-3, atol=1e-5)
    gp = model.get_params()
    opts = [torch.optim.RMSprop(g, lr=opt.adv_lr) for g in gp[:5]] + [torch.optim.Adam(gp[5], lr=opt.vae_lr)]
    reference_step(losses, opts)
    named = dict(model.named_parameters())
    for k in ("content_disc.weight", "cause_disc.weight", "content_classifier.bias", "decoder.weight", "pair_classifier.weight",
              "encoder.pooler.dense.weight", "encoder.encoder.layer.0.intermediate.dense.weight", "encoder.embeddings.word_embeddings.weight"):
        assert relnorm(named[k].grad, grads[k]) < 4e-2, (k, relnorm(named[k].grad, grads[k]))
