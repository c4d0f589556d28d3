"""Embeddings+LN, LayerNorm fwd/bwd (with fused dropout backward), column sums: HIP vs torch fp64."""
import ctypes as C

import numpy as np
import pytest
import torch

from carel_vae_amd import _lib as L
from oracle import carel_oracle as O
from tests.gpu_util import rel_err

pytestmark = pytest.mark.gpu
H = 768


def _rand(shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).cuda()


def keep_mask(seed, site, n, p, off=0):
    idx = (np.arange(n, dtype=np.uint64) + np.uint64(off)).astype(np.uint32)
    return torch.from_numpy(O.dropout_keep(seed, site, idx, p).astype(np.float64) / (1 - p)).cuda()


@pytest.mark.parametrize("rows", [1000, 2049, 3000, 4100])
def test_layernorm_fwd_bwd(rows):
    """rows not a multiple of the block row counts; the backward kernel runs 1 / 2 / 4 rows per wave (<= 2048 / <= 4096 / more rows)"""
    lib = L.load()
    assert lib.carel_layernorm_bwd_blocks(rows) == {1000: 250, 2049: 257, 3000: 375, 4100: 257}[rows]
    h, g, b = _rand((rows, H), 2.0, 1) + 0.3, 1 + _rand((H,), 0.1, 2), _rand((H,), 0.1, 3)
    xf = torch.empty((rows, H), device="cuda")
    xb = torch.empty((rows, H), device="cuda", dtype=torch.bfloat16)
    st = torch.empty((rows, 2), device="cuda")
    L.check(lib.carel_layernorm_fwd(h.data_ptr(), g.data_ptr(), b.data_ptr(), 1e-12, rows, H, xf.data_ptr(),
                                    xb.data_ptr(), st.data_ptr(), L.current_stream()))
    hd = h.double().requires_grad_(True)
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(hd, (H,), gd, bd, 1e-12)
    assert rel_err(xf, ref.detach()) < 1e-6
    assert rel_err(xb, ref.detach()) < 4e-3
    np.testing.assert_allclose(st[:, 0].cpu().numpy(), h.double().mean(1).cpu().numpy(), rtol=1e-5, atol=1e-6)
    # backward, with the sub-layer dropout mask (site of attention-output of layer 1), p = 0.1
    dy = _rand((rows, H), 1.0, 4)
    seed, site, p, off = 99, O.site_attn_out(1), 0.1, 7 * H
    nblk = lib.carel_layernorm_bwd_blocks(rows)
    part = torch.empty(nblk * 3 * H, device="cuda")
    dh = torch.empty((rows, H), device="cuda")
    dyb = torch.empty((rows, H), device="cuda", dtype=torch.bfloat16)
    dg, db, dbias = (torch.empty(H, device="cuda") for _ in range(3))
    L.check(lib.carel_layernorm_bwd(dy.data_ptr(), h.data_ptr(), st.data_ptr(), g.data_ptr(), rows, H, seed, site, off, p,
                                    dh.data_ptr(), dyb.data_ptr(), dg.data_ptr(), db.data_ptr(), dbias.data_ptr(),
                                    part.data_ptr(), L.current_stream()))
    ref.backward(dy.double())
    assert rel_err(dh, hd.grad) < 1e-5
    assert rel_err(dg, gd.grad) < 1e-5
    assert rel_err(db, bd.grad) < 1e-5
    m = keep_mask(seed, site, rows * H, p, off).view(rows, H)
    assert rel_err(dyb, hd.grad * m) < 4e-3
    assert rel_err(dbias, (hd.grad * m).sum(0)) < 1e-5


@pytest.mark.parametrize("variant", ["bert", "roberta"])
def test_embeddings_fwd_bwd(variant):
    lib = L.load()
    B, S = 6, 128
    cfg = O.EncoderConfig(layers=1, vocab_size=900) if variant == "bert" else \
        O.EncoderConfig(layers=1, vocab_size=900, max_pos=514, type_vocab=1, ln_eps=1e-5, variant="roberta", pad_id=1)
    batch = O.synthetic_batch(B, S, cfg, 16, seed=3, shape="B")
    ids = batch["input_ids"].cuda()
    tt = batch["token_type_ids"].cuda()
    if variant == "bert":
        tt[:, 5:9] = 1
    word, pos, typ = _rand((cfg.vocab_size, H), 0.5, 1), _rand((cfg.max_pos, H), 0.5, 2), _rand((cfg.type_vocab, H), 0.5, 3)
    g, b = 1 + _rand((H,), 0.1, 4), _rand((H,), 0.1, 5)
    seed, p, off = 5, 0.1, 3 * S * H
    a = L.EmbedArgs()
    a.input_ids, a.token_type_ids = ids.data_ptr(), tt.data_ptr()
    a.word_emb, a.pos_emb, a.type_emb = word.data_ptr(), pos.data_ptr(), typ.data_ptr()
    a.ln_gamma, a.ln_beta, a.ln_eps = g.data_ptr(), b.data_ptr(), cfg.ln_eps
    a.batch, a.seq_len, a.hidden = B, S, H
    a.vocab_size, a.max_pos, a.type_vocab = cfg.vocab_size, cfg.max_pos, cfg.type_vocab
    a.roberta, a.pad_id = int(variant == "roberta"), cfg.pad_id
    a.drop_seed, a.drop_idx_offset, a.drop_p = seed, off, p
    xf = torch.empty((B * S, H), device="cuda")
    xb = torch.empty((B * S, H), device="cuda", dtype=torch.bfloat16)
    st = torch.empty((B * S, 2), device="cuda")
    a.x_f32, a.x_bf16, a.stats = xf.data_ptr(), xb.data_ptr(), st.data_ptr()
    L.check(lib.carel_embed_ln_fwd(C.byref(a), L.current_stream()))
    wd, pd_, td = (t.double().requires_grad_(True) for t in (word, pos, typ))
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    pid = O.position_ids(batch["input_ids"], cfg).cuda()
    e = wd[ids] + pd_[pid] + td[tt]
    m = keep_mask(seed, O.SITE_EMBED, B * S * H, p, off).view(B, S, H)
    ref = torch.nn.functional.layer_norm(e, (H,), gd, bd, cfg.ln_eps) * m
    assert rel_err(xf.view(B, S, H), ref.detach()) < 1e-6
    assert rel_err(xb.view(B, S, H), ref.detach()) < 4e-3
    dx = _rand((B * S, H), 1.0, 6)
    nblk = lib.carel_embed_ln_bwd_blocks(B * S)
    part = torch.empty(nblk * (2 + cfg.type_vocab) * H, device="cuda")
    dword, dpos, dtyp = torch.zeros_like(word), torch.zeros_like(pos), torch.zeros_like(typ)
    dg, db = torch.empty(H, device="cuda"), torch.empty(H, device="cuda")
    L.check(lib.carel_embed_ln_bwd(C.byref(a), dx.data_ptr(), dword.data_ptr(), dpos.data_ptr(), dtyp.data_ptr(),
                                   dg.data_ptr(), db.data_ptr(), part.data_ptr(), L.current_stream()))
    ref.backward(dx.double().view(B, S, H))
    assert rel_err(dword, wd.grad) < 1e-5
    assert rel_err(dpos, pd_.grad) < 1e-5
    assert rel_err(dtyp, td.grad) < 1e-5
    assert rel_err(dg, gd.grad) < 1e-5
    assert rel_err(db, bd.grad) < 1e-5


def test_colsum_bf16():
    lib = L.load()
    rows, n = 1000, 2304
    x = _rand((rows, n), 1.0, 8).bfloat16()
    out = torch.full((n,), 3.0, device="cuda")
    part = torch.empty(((rows + 255) // 256) * n, device="cuda")
    L.check(lib.carel_colsum_bf16(x.data_ptr(), n, rows, n, out.data_ptr(), 0, part.data_ptr(), L.current_stream()))
    assert rel_err(out, x.double().sum(0)) < 1e-6
    L.check(lib.carel_colsum_bf16(x.data_ptr(), n, rows, n, out.data_ptr(), 1, part.data_ptr(), L.current_stream()))
    assert rel_err(out, 2 * x.double().sum(0)) < 1e-6
