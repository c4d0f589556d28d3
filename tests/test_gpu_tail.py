"""VAE tail (pooler -> latents -> sample -> all loss terms -> gradients) through the C ABI vs the oracle
(fp32, autograd on CPU).  Includes classifier dropout with the shared counter masks, the all-negative
batch (pair loss replaced by 0) and the data-parallel global-batch hooks."""
import numpy as np
import pytest
import torch

from carel_vae_amd import _lib as L
from carel_vae_amd import ops
from oracle import carel_oracle as O

pytestmark = pytest.mark.gpu
TAIL_KEYS = ["encoder.pooler.dense.weight", "encoder.pooler.dense.bias",
             "emotion_mu.weight", "emotion_mu.bias", "emotion_log_var.weight", "emotion_log_var.bias",
             "cause_mu.weight", "cause_mu.bias", "cause_log_var.weight", "cause_log_var.bias",
             "emotion_classifier.weight", "emotion_classifier.bias", "cause_classifier.weight", "cause_classifier.bias",
             "pair_classifier.weight", "pair_classifier.bias", "decoder.weight", "decoder.bias"]


def setup(B, S, V, seed, all_negative=False):
    cfg = O.EncoderConfig(layers=0, vocab_size=50)
    opt = O.Opt(pair_bow_dim=V)
    P = {k: v for k, v in O.init_params(cfg, opt, seed=seed).items() if k in TAIL_KEYS}
    g = torch.Generator().manual_seed(seed)
    P["encoder.pooler.dense.weight"] = torch.randn((768, 768), generator=g) * 0.05
    x_last = torch.randn((B * S, 768), generator=g)
    batch = O.synthetic_batch(B, 8, O.EncoderConfig(layers=1, vocab_size=50), V, seed=seed)
    if all_negative:
        batch["labels"].zero_(); batch["cau_labels"].zero_()
    eps_e, eps_c = torch.randn(24, generator=g), torch.randn(24, generator=g)
    return cfg, opt, P, x_last, batch, eps_e, eps_c


def oracle_tail(P, x_last, batch, eps_e, eps_c, opt, B, S, it, train, seed, **kw):
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xg = x_last.clone().requires_grad_(True)
    pooled = torch.tanh(xg.view(B, S, 768)[:, 0] @ Pg["encoder.pooler.dense.weight"].t() + Pg["encoder.pooler.dense.bias"])
    out = O.tail_forward(Pg, pooled, batch["emo_labels"], batch["cau_labels"], batch["labels"], batch["bow_reps"], it, opt,
                         eps_e, eps_c, train=train, seed=seed, **kw)
    out["loss"].backward()
    return out, pooled, {k: v.grad for k, v in Pg.items()}, xg.grad


def hip_tail(P, x_last, batch, eps_e, eps_c, opt, B, S, V, it, drop, serial=0, **kw):
    dev = "cuda"
    W = {k: v.to(dev) for k, v in P.items()}
    G = {k: torch.full_like(v, float("nan")) for k, v in W.items()}
    buf = ops.TailBuffers(B, S, 24, opt.e_num_class, V, dev)
    labels = dict(emo=batch["emo_labels"].to(dev).view(-1).contiguous(), cau=batch["cau_labels"].to(dev).view(-1).contiguous(),
                  pair=batch["labels"].to(dev).view(-1).contiguous(), bow=batch["bow_reps"].to(dev).contiguous())
    xl = x_last.to(dev)
    a = ops.tail_args(buf, xl, W, labels, eps_e.to(dev), eps_c.to(dev), opt, ops.kl_anneal_weight(it, opt), grads=G, drop=drop, **kw)
    a._keep = (W, G, labels, xl)
    a.serial = serial
    ops.tail_latents(a)
    ops.tail_losses(a)
    ops.tail_backward(a, None)
    torch.cuda.synchronize()
    return buf, G


@pytest.mark.parametrize("B,V,train", [(64, 23771, True), (8, 257, False)])
def test_loss_kernel_beside_the_decoder_passes_same_bits(B, V, train):
    """carel_tail_args.serial = 0 (default: the single-workgroup loss kernel on the library's side stream while the decoder passes run on the
    caller's stream, joined inside the call) against serial = 1 (one stream, in order): every output and gradient bit-identical, also
    when the call is repeated back to back (the side stream's events are re-used)."""
    cfg, opt, P, x_last, batch, eps_e, eps_c = setup(B, 32, V, seed=5)
    drop = (opt.dropout if train else 0.0, 11, 0)
    ref_buf, ref_G = hip_tail(P, x_last, batch, eps_e, eps_c, opt, B, 32, V, 3, drop, serial=1)
    for rep in range(3):
        buf, G = hip_tail(P, x_last, batch, eps_e, eps_c, opt, B, 32, V, 3, drop, serial=0)
        assert torch.equal(buf.terms, ref_buf.terms), rep
        assert torch.equal(buf.z, ref_buf.z) and torch.equal(buf.dx_last, ref_buf.dx_last), rep
        for k in G:
            assert torch.equal(torch.nan_to_num(G[k]), torch.nan_to_num(ref_G[k])), (rep, k)


def close(got, ref, rtol, atol, name):
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=rtol, atol=atol, err_msg=name)


@pytest.mark.parametrize("B,V,train,allneg", [(64, 23771, False, False), (8, 257, False, False), (16, 1000, True, False),
                                              (8, 130, False, True), (3, 77, True, False)])
def test_tail_matches_oracle(B, V, train, allneg):
    S, it, seed = 4, 7, 31
    cfg, opt, P, x_last, batch, eps_e, eps_c = setup(B, S, V, seed=B + V, all_negative=allneg)
    out, pooled, grads, dx = oracle_tail(P, x_last, batch, eps_e, eps_c, opt, B, S, it, train, seed)
    drop = (opt.dropout if train else 0.0, seed, 0)
    buf, G = hip_tail(P, x_last, batch, eps_e, eps_c, opt, B, S, V, it, drop)
    close(buf.pooled, pooled, 1e-5, 5e-6, "pooled")   # fp32 summation-order noise of 768-long dots
    lat = torch.cat((out["mu_e"], out["lv_e"], out["mu_c"], out["lv_c"]), dim=1)
    close(buf.lat, lat, 1e-5, 1e-5, "lat")
    close(buf.z, torch.cat((out["z_e"], out["z_c"]), dim=1), 1e-5, 2e-5, "z")
    t = buf.terms.cpu().numpy()
    for i, k in ((1, "mmd"), (2, "emo"), (3, "cau"), (4, "pair"), (5, "kl_e"), (6, "kl_c"), (7, "rec"), (8, "loss")):
        np.testing.assert_allclose(t[i], float(out[k]), rtol=1e-4, atol=1e-5, err_msg=k)
    for k in TAIL_KEYS:
        ref = grads[k] if grads[k] is not None else torch.zeros_like(P[k])   # dead pair head -> zeros here
        scale = float(ref.abs().max()) + 1e-12
        close(G[k], ref, 2e-4, 2e-5 * scale + 1e-9, k)
    scale = float(dx.abs().max())
    close(buf.dx_last, dx, 2e-4, 2e-5 * scale, "dx_last")


@pytest.mark.parametrize("strided", [False, True])
def test_tail_global_batch_hooks(strided):
    """Two shards of a 32-sample batch with the DP hooks == the oracle on the unsharded batch.  strided: the gathered
    latents in the layout of DataParallel.fill_global (each rank's z followed by 16 spare floats)."""
    B, S, V, it, seed, R = 32, 2, 300, 3, 9, 2
    cfg, opt, P, x_last, batch, eps_e, eps_c = setup(B, S, V, seed=5)
    out, pooled, grads, dx = oracle_tail(P, x_last, batch, eps_e, eps_c, opt, B, S, it, True, seed)
    # pass 1: latents of each shard -> "all-gather" z
    Bl = B // R
    shards = []
    for r in range(R):
        sl = slice(r * Bl, (r + 1) * Bl)
        sb = {k: v[sl] for k, v in batch.items()}
        shards.append((sb, x_last[r * Bl * S:(r + 1) * Bl * S]))
    zs = []
    for r, (sb, xl) in enumerate(shards):
        buf, _ = hip_tail(P, xl, sb, eps_e, eps_c, opt, Bl, S, V, it, (opt.dropout, seed, r * Bl))
        zs.append(buf.z.clone())
    zg = torch.cat(zs, dim=0).contiguous()
    stride = 0
    if strided:
        stride = Bl * 48 + 16
        packed = torch.full((R, stride), float("nan"), device="cuda")
        for r in range(R):
            packed[r, :Bl * 48] = zs[r].reshape(-1)
            packed[r, Bl * 48] = float(shards[r][0]["labels"].sum())      # each rank's own label sum rides behind its z
        zg = packed
    ysum = batch["labels"].sum().reshape(1).cuda()
    label_ranks = 0
    if strided:
        ysum, label_ranks = packed.view(-1)[Bl * 48:], R
    tot = {k: torch.zeros_like(v) for k, v in P.items()}
    loss = 0.0
    for r, (sb, xl) in enumerate(shards):
        buf, G = hip_tail(P, xl, sb, eps_e, eps_c, opt, Bl, S, V, it, (opt.dropout, seed, r * Bl), global_label_sum=ysum,
                          global_n=B, global_row_offset=r * Bl, z_global=zg, mmd_grad_scale=float(R), global_rank_stride=stride, global_label_ranks=label_ranks)
        for k in tot:
            tot[k] += G[k].cpu() / R          # gradient averaging over ranks
        t = buf.terms.cpu().numpy()
        np.testing.assert_allclose(t[1], float(out["mmd"]), rtol=3e-5, atol=2e-6)      # global statistic on every rank
        loss += t[8] / R
        scale = float(dx.abs().max())
        close(buf.dx_last / R, dx[r * Bl * S:(r + 1) * Bl * S], 3e-4, 3e-5 * scale, "dx shard")
    np.testing.assert_allclose(loss, float(out["loss"]), rtol=1e-4, atol=1e-5)
    for k in TAIL_KEYS:
        scale = float(grads[k].abs().max()) + 1e-12
        close(tot[k], grads[k], 3e-4, 3e-5 * scale, k)


def test_tail_global_batch_eight_ranks_of_64():
    """The 8-GPU configuration of BASELINE.json: 8 shards x 64 samples, global-batch MMD over 512 samples per side (1 M
    pairs, its own multi-workgroup kernel).  Two of the shards are run and compared with the oracle on the whole batch."""
    B, S, V, it, seed, R = 512, 1, 200, 3, 9, 8
    cfg, opt, P, x_last, batch, eps_e, eps_c = setup(B, S, V, seed=6)
    out, pooled, grads, dx = oracle_tail(P, x_last, batch, eps_e, eps_c, opt, B, S, it, True, seed)
    Bl = B // R
    zs = []
    for r in range(R):
        sb = {k: v[r * Bl:(r + 1) * Bl] for k, v in batch.items()}
        buf, _ = hip_tail(P, x_last[r * Bl * S:(r + 1) * Bl * S], sb, eps_e, eps_c, opt, Bl, S, V, it, (opt.dropout, seed, r * Bl))
        zs.append(buf.z.clone())
    stride = Bl * 48 + 16
    packed = torch.full((R, stride), float("nan"), device="cuda")
    for r in range(R):
        packed[r, :Bl * 48] = zs[r].reshape(-1)
        packed[r, Bl * 48] = float(batch["labels"][r * Bl:(r + 1) * Bl].sum())
    for r in (0, 5):
        sb = {k: v[r * Bl:(r + 1) * Bl] for k, v in batch.items()}
        buf, G = hip_tail(P, x_last[r * Bl * S:(r + 1) * Bl * S], sb, eps_e, eps_c, opt, Bl, S, V, it, (opt.dropout, seed, r * Bl),
                          global_label_sum=packed.view(-1)[Bl * 48:], global_n=B, global_row_offset=r * Bl, z_global=packed,
                          mmd_grad_scale=float(R), global_rank_stride=stride, global_label_ranks=R)
        t = buf.terms.cpu().numpy()
        np.testing.assert_allclose(t[1], float(out["mmd"]), rtol=5e-5, atol=2e-6)
        scale = float(dx.abs().max())
        close(buf.dx_last / R, dx[r * Bl * S:(r + 1) * Bl * S], 5e-4, 5e-5 * scale, "dx shard %d" % r)


def test_pair_probs_matches_oracle():
    B, S, V = 50, 2, 64
    cfg, opt, P, x_last, batch, eps_e, eps_c = setup(B, S, V, seed=77)
    buf, _ = hip_tail(P, x_last, batch, eps_e, eps_c, opt, B, S, V, 0, (0.0, 0, 0))
    prob = ops.pair_probs(buf.lat, eps_e.cuda(), eps_c.cuda(), P["pair_classifier.weight"].cuda(), P["pair_classifier.bias"].cuda(), 24)
    lat = buf.lat.cpu()
    z = torch.cat((lat[:, :24] + eps_e * lat[:, 24:48].exp(), lat[:, 48:72] + eps_c * lat[:, 72:].exp()), dim=1)
    ref = torch.sigmoid(z @ P["pair_classifier.weight"].t() + P["pair_classifier.bias"]).squeeze(1)
    close(prob, ref, 1e-5, 1e-6, "pair prob")


@pytest.mark.parametrize("dis", ["hsic", "mmd"])
@pytest.mark.parametrize("train", [False, True])
def test_hsic_variant_of_the_tail(train, dis):
    """Ablation heads (config 5) with the one-logit BCE emotion head: +HSIC(z_e, z_c) instead of -30*MMD
    (drl_classifier_ec_hsic.py :214, :253, :455-470), and the earlier MMD scripts (drl_classifier_ec_mmd.py /
    drl_classifier_ec_mmd_final.py: the same RBF-MMD term, :211-212 / :235, with that emotion head), vs the oracle (whose
    HSIC and MMD are pinned to the reference's values)."""
    B, S, V, it, seed = 32, 2, 200, 5, 17
    cfg = O.EncoderConfig(layers=0, vocab_size=50)
    opt = O.Opt(pair_bow_dim=V, e_num_class=1)
    opt.disentangle, opt.emotion_head = dis, "bce"
    P = {k: v for k, v in O.init_params(cfg, opt, seed=3).items() if k in TAIL_KEYS}
    g = torch.Generator().manual_seed(4)
    P["encoder.pooler.dense.weight"] = torch.randn((768, 768), generator=g) * 0.05
    x_last = torch.randn((B * S, 768), generator=g)
    batch = O.synthetic_batch(B, 8, O.EncoderConfig(layers=1, vocab_size=50), V, seed=6)
    batch["emo_labels"] = (batch["emo_labels"] > 2).to(torch.int64)
    eps_e, eps_c = torch.randn(24, generator=g), torch.randn(24, generator=g)
    out, pooled, grads, dx = oracle_tail(P, x_last, batch, eps_e, eps_c, opt, B, S, it, train, seed, disentangle=dis, emotion_head="bce")
    buf, G = hip_tail(P, x_last, batch, eps_e, eps_c, opt, B, S, V, it, (opt.dropout if train else 0.0, seed, 0))
    t = buf.terms.cpu().numpy()
    for i, k in ((1, "mmd"), (2, "emo"), (3, "cau"), (4, "pair"), (7, "rec"), (8, "loss")):
        np.testing.assert_allclose(t[i], float(out[k]), rtol=2e-4, atol=1e-5, err_msg=k)
    for k in TAIL_KEYS:
        ref = grads[k] if grads[k] is not None else torch.zeros_like(P[k])
        scale = float(ref.abs().max()) + 1e-12
        close(G[k], ref, 5e-4, 5e-5 * scale + 1e-9, k)
    close(buf.dx_last, dx, 5e-4, 5e-5 * float(dx.abs().max()), "dx_last")
