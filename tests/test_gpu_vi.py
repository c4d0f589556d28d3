"""VI / CLUB ablation head (drl_classifier_ec_vi.py) on the GPU: the two approximation-network kernels against the CPU
oracle, and the two-phase training step of the model (reference loop :754-774) against the golden vectors produced by
the reference's own class (tests/golden/gen_golden_vi.py) and against the bf16-emulating oracle."""
import os

import numpy as np
import pytest
import torch

from carel_vae_amd import drl_classifier as M
from carel_vae_amd import ops
from oracle import carel_oracle as O

pytestmark = pytest.mark.gpu


def _net(opt, seed):
    P = O.init_vi_params(opt, seed=seed)
    return P, [P[k].cuda().contiguous() for k in O.VI_KEYS]


@pytest.mark.parametrize("B,D", [(64, 24), (16, 24), (1, 24), (200, 24), (33, 7), (8, 32)])
def test_vi_kernels_vs_oracle(B, D):
    opt = O.Opt(ec_dim=D)
    P, net = _net(opt, 3 + B)
    rs = np.random.RandomState(B * 7 + D)
    z = torch.from_numpy(rs.standard_normal((B, 2 * D)).astype(np.float32) * 0.8)
    perm = torch.from_numpy(rs.permutation(B).astype(np.int32))
    # approximation loss + its parameter gradients
    loss, grads = ops.vi_aprx(z.cuda(), net)
    leaf = {k: P[k].clone().requires_grad_(True) for k in O.VI_KEYS}
    ref = O.vi_aprx_loss(leaf, z[:, :D], z[:, D:])
    ref.backward()
    np.testing.assert_allclose(float(loss), float(ref.detach()), rtol=2e-5)
    for k, g in zip(O.VI_KEYS, grads):
        r = leaf[k].grad
        assert float((g.cpu() - r).abs().max()) <= 2e-5 * max(float(r.abs().max()), 1e-3), k
    # CLUB bound + gradient wrt the embeddings
    up, dz = ops.vi_upper(z.cuda(), net, perm.cuda())
    ze, zc = z[:, :D].clone().requires_grad_(True), z[:, D:].clone().requires_grad_(True)
    ref = O.vi_upper_loss(P, ze, zc, perm)
    ref.backward()
    scale = float(sum(abs(float(x)) for x in ((O.vi_net(P, z[:, D:])[0] - z[:, :D]) ** 2).sum(1))) / B   # the bound is a difference of sums this big
    assert abs(float(up) - float(ref.detach())) <= 2e-6 * max(scale, 1.0)
    rd = torch.cat((ze.grad, zc.grad), 1)
    assert float((dz.cpu() - rd).abs().max()) <= 2e-5 * max(float(rd.abs().max()), 1e-6)


def test_vi_argument_checks():
    opt = O.Opt()
    _, net = _net(opt, 1)
    z = torch.zeros(8, 48, device="cuda")
    with pytest.raises(Exception):
        ops.vi_aprx(z[:, :47].contiguous(), net)
    with pytest.raises(Exception):
        ops.vi_upper(z, net, torch.zeros(7, dtype=torch.int32, device="cuda"))
    with pytest.raises(Exception):
        ops.vi_aprx(z, net[:7])
    with pytest.raises(Exception):
        ops.vi_aprx(torch.zeros(8, 80, device="cuda"), net)       # ec_dim 40 > 32


def _build(golden_dir):
    cfg, opt = O.EncoderConfig(layers=2, vocab_size=900), O.Opt(pair_bow_dim=211, dropout=0.0, e_num_class=1)
    z = np.load(os.path.join(golden_dir, "vi_small.npz"), allow_pickle=False)
    batch = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    wseed = int(z["meta"][5])
    mopt = M.make_opt(**vars(opt), disentangle="vi", emotion_head="bce")
    model = M.DrlClassifier(mopt, M.encoder_config("zh", vocab_size=cfg.vocab_size, layers=cfg.layers, hidden_dropout=0.0, attn_dropout=0.0))
    P = {**O.init_params(cfg, opt, seed=wseed), **O.init_vi_params(opt, seed=wseed + 1)}
    model.load_state_dict(P)
    model.to("cuda")
    return cfg, opt, z, batch, model, P


@pytest.mark.parametrize("fused", [True, False])
def test_vi_two_phase_training_vs_golden_and_oracle(golden_dir, fused):
    cfg, opt, z, batch, model, P = _build(golden_dir)
    steps = int(z["meta"][7])
    model.train()
    ec_aprx_params, other_params = model.get_params()
    assert len(ec_aprx_params) == 8
    ec_aprx_opt = torch.optim.Adam(ec_aprx_params, lr=opt.aprx_lr)
    vae_and_cls_opt = M.FusedAdam(model, lr=opt.vae_lr) if fused else torch.optim.Adam(other_params, lr=opt.vae_lr)
    b = {k: v.cuda() for k, v in batch.items()}
    st_vae, st_aprx = O.AdamState(), O.AdamState()
    Pq = dict(P)
    for s in range(steps):
        eps_e, eps_c = torch.from_numpy(z[f"eps_e_{s}"]), torch.from_numpy(z[f"eps_c_{s}"])
        perm = torch.from_numpy(z[f"perm_{s}"])
        epoch = int(z["epochs"][s])
        model.set_noise(eps_e, eps_c)
        # ---- the reference's step, line for line (:754-774)
        e_embedding, c_embedding, ec_aprx_loss, vae_and_cls_loss = model(b["input_ids"], b["attention_masks"], b["token_type_ids"],
                                                                         b["emo_labels"], b["cau_labels"], b["labels"], b["bow_reps"], 5 + s)
        ec_aprx_opt.zero_grad()
        ec_aprx_loss.backward(retain_graph=True)
        if s == 0:
            named = dict(model.named_parameters())
            for k in O.VI_KEYS:
                r = z["ga_" + k]
                assert float(np.abs(named[k].grad.cpu().numpy() - r).max()) <= 2e-2 * max(float(np.abs(r).max()), 1e-3), k
        ec_aprx_opt.step()
        Rj_loss = model.get_ec_upper_loss(e_embedding, c_embedding, random_index=perm)
        beta = min(1.0, (epoch - 1) * 0.1)
        vae_only = float(vae_and_cls_loss)
        vae_and_cls_loss += beta * Rj_loss
        vae_and_cls_opt.zero_grad()
        vae_and_cls_loss.backward()
        if s == 1:
            grads_1 = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        vae_and_cls_opt.step()
        # ---- (a) golden from the reference class (fp32): bf16-level tolerance
        assert float((e_embedding.detach().cpu() - torch.from_numpy(z[f"z_e_{s}"])).norm()) <= 2e-2 * float(np.linalg.norm(z[f"z_e_{s}"]))
        for name, got in (("aprx", float(ec_aprx_loss)), ("vae", vae_only), ("total", float(vae_and_cls_loss))):
            ref = float(z[f"{name}_{s}"])
            assert abs(got - ref) <= 2e-2 * abs(ref), (name, s, got, ref)
        assert abs(float(Rj_loss) - float(z[f"upper_{s}"])) <= 2e-2 * max(abs(float(z[f"upper_{s}"])), 1.0)
        # ---- (b) bf16-emulating oracle: tight
        Pq, out = O.vi_train_step(Pq, batch, 5 + s, epoch, cfg, opt, eps_e, eps_c, perm, st_vae, st_aprx, quant=O.bf16_round)
        tol = 3e-3
        assert abs(float(ec_aprx_loss) - float(out["aprx"])) <= tol * abs(float(out["aprx"])), s
        assert abs(vae_only - float(out["vae"])) <= tol * abs(float(out["vae"])), s
        assert abs(float(Rj_loss) - float(out["upper"])) <= tol * max(abs(float(out["upper"])), 1.0), s
    # gradients at step 1 (beta = 0.3) include the CLUB term routed through z: compare against the golden slices
    for k in z.files:
        if k.startswith("g_"):
            pk = k[2:]
            if pk.endswith("key.bias") or pk.startswith(O.UNOPTIMISED_PREFIXES):
                continue    # analytically zero / never zeroed by either optimiser in the reference (accumulates, unused)
            f = grads_1[pk].reshape(-1)
            n = 64
            step = max(1, f.numel() // n)
            got = torch.cat((f[:n], f[-n:], f[::step][:n])).cpu().numpy()
            scale = float(z["gn_" + pk]) / max(1.0, np.sqrt(f.numel() / 192.0))
            assert float(np.abs(got - z[k]).max()) <= 0.08 * max(float(np.abs(z[k]).max()), scale), pk
    # weights: the approximation net moved with lr 3e-3 for 3 steps, everything else with 1e-5
    named = dict(model.named_parameters())
    for k in O.VI_KEYS:
        d = np.abs(named[k].detach().cpu().numpy() - z["w_" + k])
        # an element whose gradient is at the bf16 noise level can take an Adam step of the opposite sign (2 lr apart)
        assert float((d > 0.7 * opt.aprx_lr).mean()) <= 0.03 and float(d.max()) <= 2.2 * steps * opt.aprx_lr, (k, float(d.max()))
        assert float((named[k].detach().cpu() - P[k]).abs().max()) > 0.5 * opt.aprx_lr, k     # and it did move
    for k in ("emotion_mu.weight", "cause_log_var.weight"):        # never optimised (quirk Q3)
        assert torch.equal(named[k].detach().cpu(), P[k])


def test_extra_loss_on_sampled_embeddings_reaches_encoder(golden_dir):
    """The z outputs are part of the autograd graph: d(loss + f(z))/d theta == d loss/d theta + routed f gradient.
    Checked by linearity: grad(loss + 2 f) - grad(loss + f) == grad(loss + f) - grad(loss)."""
    cfg, opt, z, batch, model, P = _build(golden_dir)
    model.train()
    b = {k: v.cuda() for k, v in batch.items()}
    eps_e, eps_c = torch.from_numpy(z["eps_e_0"]), torch.from_numpy(z["eps_c_0"])
    key = "encoder.encoder.layer.0.intermediate.dense.weight"
    got = []
    for w in (0.0, 1.0, 2.0):
        model.set_noise(eps_e, eps_c)
        ze, zc, _, loss = model(b["input_ids"], b["attention_masks"], b["token_type_ids"], b["emo_labels"], b["cau_labels"], b["labels"],
                                b["bow_reps"], 5)
        for p in model.parameters():
            p.grad = None
        (loss + w * ((ze * zc).sum() + ze.pow(2).mean())).backward()
        got.append(dict(model.named_parameters())[key].grad.detach().clone())
    d1, d2 = got[1] - got[0], got[2] - got[1]
    assert float(d1.norm()) > 1e-3 * float(got[0].norm())
    assert float((d1 - d2).norm()) <= 2e-2 * float(d1.norm())


def test_vi_final_variant_six_way_emotion_head(golden_dir):
    """drl_classifier_ec_vi_final.py = the VI step with the main script's six-way CE emotion head (:465-477):
    opt.disentangle = "vi", opt.emotion_head = "ce".  Against the bf16-emulating oracle (its CE head is pinned by the main
    goldens, its VI part by vi_small)."""
    cfg, opt = O.EncoderConfig(layers=2, vocab_size=900), O.Opt(pair_bow_dim=211, dropout=0.0, e_num_class=6)
    z = np.load(os.path.join(golden_dir, "vi_small.npz"), allow_pickle=False)
    wseed, bseed = int(z["meta"][5]), int(z["meta"][6])
    batch = O.synthetic_batch(16, 128, cfg, opt.pair_bow_dim, seed=bseed, shape="B")       # six-class emotion labels
    mopt = M.make_opt(**vars(opt), disentangle="vi", emotion_head="ce")
    model = M.DrlClassifier(mopt, M.encoder_config("zh", vocab_size=cfg.vocab_size, layers=cfg.layers, hidden_dropout=0.0, attn_dropout=0.0))
    P = {**O.init_params(cfg, opt, seed=wseed), **O.init_vi_params(opt, seed=wseed + 1)}
    model.load_state_dict(P)
    model.to("cuda").train()
    ec_aprx_params, other_params = model.get_params()
    ec_aprx_opt = torch.optim.Adam(ec_aprx_params, lr=opt.aprx_lr)
    vae_and_cls_opt = M.FusedAdam(model, lr=opt.vae_lr, fuse_into_backward=True)
    b = {k: v.cuda() for k, v in batch.items()}
    st_vae, st_aprx, Pq = O.AdamState(), O.AdamState(), dict(P)
    for s, epoch in enumerate((3, 8)):
        eps_e, eps_c = torch.from_numpy(z[f"eps_e_{s}"]), torch.from_numpy(z[f"eps_c_{s}"])
        perm = torch.from_numpy(z[f"perm_{s}"])
        model.set_noise(eps_e, eps_c)
        e_embedding, c_embedding, ec_aprx_loss, vae_and_cls_loss = model(b["input_ids"], b["attention_masks"], b["token_type_ids"],
                                                                         b["emo_labels"], b["cau_labels"], b["labels"], b["bow_reps"], 5 + s)
        ec_aprx_opt.zero_grad()
        ec_aprx_loss.backward(retain_graph=True)
        ec_aprx_opt.step()
        Rj_loss = model.get_ec_upper_loss(e_embedding, c_embedding, random_index=perm)
        vae_only = float(vae_and_cls_loss.detach())
        vae_and_cls_loss += min(1.0, (epoch - 1) * 0.1) * Rj_loss
        vae_and_cls_opt.zero_grad()
        vae_and_cls_loss.backward()
        vae_and_cls_opt.step()
        Pq, out = O.vi_train_step(Pq, batch, 5 + s, epoch, cfg, opt, eps_e, eps_c, perm, st_vae, st_aprx, quant=O.bf16_round, emotion_head="ce")
        assert abs(float(ec_aprx_loss.detach()) - float(out["aprx"])) <= 3e-3 * abs(float(out["aprx"])), s
        assert abs(vae_only - float(out["vae"])) <= 3e-3 * abs(float(out["vae"])), s
        assert abs(float(Rj_loss.detach()) - float(out["upper"])) <= 3e-3 * max(abs(float(out["upper"])), 1.0), s
        assert abs(float(vae_and_cls_loss.detach()) - float(out["total"])) <= 3e-3 * abs(float(out["total"])), s
