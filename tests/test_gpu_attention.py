"""Self-attention forward/backward kernels vs torch fp64 eager attention on the same bf16 inputs
(padding masks, dropout with the shared counter-based masks, S in {32, 64, 128})."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from carel_vae_amd import _lib as L
from oracle import carel_oracle as O
from tests.gpu_util import rel_err

pytestmark = pytest.mark.gpu
NH, HD, H = 12, 64, 768


def run_attn(qkv, mask, B, S, drop=(0, 0, 0, 0.0), dctx=None, q_rows=0):
    lib = L.load()
    a = L.AttnArgs()
    a.q_rows = q_rows
    ctx = torch.full((B * S, H), float("nan"), device="cuda", dtype=torch.bfloat16)
    lse = torch.full((B, NH, S), float("nan"), device="cuda")
    a.qkv, a.attention_mask, a.ctx, a.lse = qkv.data_ptr(), (None if mask is None else mask.data_ptr()), ctx.data_ptr(), lse.data_ptr()
    a.batch, a.seq_len, a.heads, a.head_dim = B, S, NH, HD
    a.drop_seed, a.drop_site, a.drop_idx_offset, a.drop_p = drop
    L.check(lib.carel_attention_fwd(C.byref(a), L.current_stream()), "attn fwd")
    dqkv = None
    if dctx is not None:
        dqkv = torch.empty((B * S, 3 * H), device="cuda", dtype=torch.bfloat16)
        a.dctx, a.dqkv = dctx.data_ptr(), dqkv.data_ptr()
        L.check(lib.carel_attention_bwd(C.byref(a), L.current_stream()), "attn bwd")
    torch.cuda.synchronize()
    return ctx, lse, dqkv


def ref_attn(qkv, mask, B, S, drop):
    x = qkv.double().view(B, S, 3, NH, HD).requires_grad_(True)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))            # [B,NH,S,HD]
    s = q @ k.transpose(-1, -2) / math.sqrt(HD)
    if mask is not None:
        s = s + (1.0 - mask.double())[:, None, None, :] * torch.finfo(torch.float32).min
    lse = torch.logsumexp(s, dim=-1)
    pr = torch.softmax(s, dim=-1)
    seed, site, off, p = drop
    if p > 0:
        idx = (np.arange(B * NH * S * S, dtype=np.uint64) + np.uint64(off)).astype(np.uint32)
        m = torch.from_numpy(O.dropout_keep(seed, site, idx, p).astype(np.float64) / (1 - p)).cuda().view(B, NH, S, S)
        pr = pr * m
    ctx = (pr @ v).transpose(1, 2).reshape(B * S, H)
    return x, ctx, lse


@pytest.mark.parametrize("B,S,masked,p", [(3, 128, False, 0.0), (4, 128, True, 0.0), (2, 128, True, 0.1),
                                           (5, 64, True, 0.1), (3, 32, False, 0.0), (2, 96, True, 0.0)])
def test_attention_fwd_bwd(B, S, masked, p):
    g = torch.Generator().manual_seed(B * 1000 + S)
    qkv = (torch.randn((B * S, 3 * H), generator=g) * 1.5).cuda().bfloat16()
    mask = None
    if masked:
        mask = torch.ones((B, S), dtype=torch.int64)
        for b in range(B):
            ln = int(torch.randint(3, S + 1, (1,), generator=g))
            mask[b, ln:] = 0
        mask = mask.cuda()
    dctx = (torch.randn((B * S, H), generator=g)).cuda().bfloat16()
    drop = (77, O.site_attn_probs(4), 11 * NH * S * S, p)
    ctx, lse, dqkv = run_attn(qkv, mask, B, S, drop, dctx)
    x, rctx, rlse = ref_attn(qkv, mask, B, S, drop)
    assert rel_err(ctx, rctx.detach()) < 8e-3          # bf16 probabilities + bf16 output
    np.testing.assert_allclose(lse.cpu().numpy(), rlse.detach().cpu().numpy(), rtol=1e-4, atol=1e-4)
    rctx.backward(dctx.double())
    rg = x.grad.reshape(B * S, 3 * H)
    got = dqkv.double()
    for name, sl in (("dq", slice(0, H)), ("dk", slice(H, 2 * H)), ("dv", slice(2 * H, 3 * H))):
        e = rel_err(got[:, sl], rg[:, sl])
        assert e < 1.5e-2, (name, e)                   # bf16 P, dS, dO operands
    if masked:   # padded keys get exactly zero dK / dV; padded queries still produce finite values
        for b in range(B):
            pad = (mask[b] == 0).nonzero().flatten()
            if len(pad):
                rows = b * S + pad
                assert float(got[rows, H:].abs().max()) == 0.0
    assert torch.isfinite(got).all()


def test_attention_rejects_bad_shapes():
    qkv = torch.zeros((100, 3 * H), device="cuda", dtype=torch.bfloat16)
    with pytest.raises(L.CarelError):
        run_attn(qkv, None, 1, 100)


def test_packed_attention_matches_per_sample_reference():
    """cu_seqlens mode: samples of different lengths stored back to back; tiles that run past a sample must not leak."""
    lib = L.load()
    g = torch.Generator().manual_seed(3)
    S = 128
    lens = [37, 128, 5, 64, 33, 1, 96]
    B = len(lens)
    T = sum(lens)
    Tp = (T + 127) // 128 * 128
    qkv = torch.zeros((B * S, 3 * H), dtype=torch.bfloat16, device="cuda")          # buffer is B*S rows like the encoder's
    qkv[:Tp] = (torch.randn((Tp, 3 * H), generator=g) * 1.5).cuda().bfloat16()
    dctx = torch.zeros((B * S, H), dtype=torch.bfloat16, device="cuda")
    dctx[:Tp] = torch.randn((Tp, H), generator=g).cuda().bfloat16()
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    a = L.AttnArgs()
    ctx = torch.full((B * S, H), 7.0, device="cuda", dtype=torch.bfloat16)
    lse = torch.zeros((B, NH, S), device="cuda")
    dqkv = torch.full((B * S, 3 * H), 7.0, device="cuda", dtype=torch.bfloat16)
    a.qkv, a.attention_mask, a.ctx, a.lse, a.dctx, a.dqkv = qkv.data_ptr(), None, ctx.data_ptr(), lse.data_ptr(), dctx.data_ptr(), dqkv.data_ptr()
    a.batch, a.seq_len, a.heads, a.head_dim = B, S, NH, HD
    p, seed, site, off = 0.1, 5, O.site_attn_probs(2), 0
    a.drop_seed, a.drop_site, a.drop_idx_offset, a.drop_p = seed, site, off, p
    a.cu_seqlens = cu.data_ptr()
    L.check(lib.carel_attention_fwd(C.byref(a), L.current_stream()), "attn fwd packed")
    L.check(lib.carel_attention_bwd(C.byref(a), L.current_stream()), "attn bwd packed")
    torch.cuda.synchronize()
    idx_all = np.arange(B * NH * S * S, dtype=np.uint64).astype(np.uint32)
    keep_all = torch.from_numpy(O.dropout_keep(seed, site, idx_all, p).astype(np.float64) / (1 - p)).view(B, NH, S, S).cuda()
    start = 0
    for b, n in enumerate(lens):
        x = qkv[start:start + n].double().view(n, 3, NH, HD).requires_grad_(True)
        q, k, v = (x[:, i].transpose(0, 1) for i in range(3))            # [NH, n, HD]
        s = q @ k.transpose(-1, -2) / math.sqrt(HD)
        pr = torch.softmax(s, dim=-1) * keep_all[b, :, :n, :n]
        out = (pr @ v).transpose(0, 1).reshape(n, H)
        assert rel_err(ctx[start:start + n], out.detach()) < 8e-3, b
        out.backward(dctx[start:start + n].double())
        e = rel_err(dqkv[start:start + n], x.grad.reshape(n, 3 * H))
        assert e < 2e-2, (b, n, e)
        start += n
    # rows that belong to no sample were not touched
    assert float((ctx[T:] - 7.0).abs().max()) == 0.0 and float((dqkv[T:] - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("packed", [False, True])
def test_relative_position_bias_forward_backward_and_table_gradient(packed):
    """MPNet: scores += table[bucket(key - query), head] (transformers MPNetAttention; en_ec_sentence_transformer.py:22).  The kernels
    read the bias by distance (carel_relpos_expand) and accumulate its gradient by distance; carel_relpos_reduce folds that into the
    32 x 12 table.  Checked against fp64 autograd through the same bucket map, dense (padding mask) and packed, with dropout."""
    lib = L.load()
    g = torch.Generator().manual_seed(11)
    S = 128
    lens = [128, 77, 33, 5]
    B = len(lens)
    table = (torch.randn((32, NH), generator=g) * 0.7).cuda()
    rp = O.mpnet_relative_position_bucket(torch.arange(-127, 129)).to(torch.int32).cuda().contiguous()      # entry i = distance i - 127
    dist, ddist = torch.empty((NH, 256), device="cuda"), torch.zeros((B * NH, 256), device="cuda")     # gradient by distance: one row per (sample, head)
    L.check(lib.carel_relpos_expand(table.data_ptr(), rp.data_ptr(), dist.data_ptr(), L.current_stream()), "relpos expand")
    T = sum(lens)
    if packed:
        Tp = (T + 127) // 128 * 128
        qkv = torch.zeros((B * S, 3 * H), dtype=torch.bfloat16, device="cuda")
        qkv[:Tp] = (torch.randn((Tp, 3 * H), generator=g) * 1.5).cuda().bfloat16()
        dctx = torch.zeros((B * S, H), dtype=torch.bfloat16, device="cuda")
        dctx[:Tp] = torch.randn((Tp, H), generator=g).cuda().bfloat16()
        cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
        mask = None
        row0 = [int(v) for v in np.cumsum([0] + lens[:-1])]
    else:
        qkv = (torch.randn((B * S, 3 * H), generator=g) * 1.5).cuda().bfloat16()
        dctx = torch.randn((B * S, H), generator=g).cuda().bfloat16()
        mask = torch.zeros((B, S), dtype=torch.int64)
        for b, n in enumerate(lens):
            mask[b, :n] = 1
            dctx[b * S + n:(b + 1) * S] = 0       # padded queries carry no gradient (nothing downstream reads them), as in the encoder
        mask = mask.cuda()
        cu = None
        row0 = [b * S for b in range(B)]
    p, seed, site = 0.1, 9, O.site_attn_probs(1)
    a = L.AttnArgs()
    ctx = torch.zeros((B * S, H), device="cuda", dtype=torch.bfloat16)
    lse = torch.zeros((B, NH, S), device="cuda")
    dqkv = torch.zeros((B * S, 3 * H), device="cuda", dtype=torch.bfloat16)
    a.qkv, a.attention_mask, a.ctx, a.lse, a.dctx, a.dqkv = qkv.data_ptr(), (None if mask is None else mask.data_ptr()), ctx.data_ptr(), lse.data_ptr(), dctx.data_ptr(), dqkv.data_ptr()
    a.batch, a.seq_len, a.heads, a.head_dim = B, S, NH, HD
    a.drop_seed, a.drop_site, a.drop_idx_offset, a.drop_p = seed, site, 0, p
    a.cu_seqlens = None if cu is None else cu.data_ptr()
    a.rel_bias_dist, a.d_rel_bias_dist = dist.data_ptr(), ddist.data_ptr()
    L.check(lib.carel_attention_fwd(C.byref(a), L.current_stream()), "attn fwd rel")
    L.check(lib.carel_attention_bwd(C.byref(a), L.current_stream()), "attn bwd rel")
    dtable = torch.empty((32, NH), device="cuda")
    L.check(lib.carel_relpos_reduce(ddist.data_ptr(), B, rp.data_ptr(), dtable.data_ptr(), 0, L.current_stream()), "relpos reduce")
    torch.cuda.synchronize()
    keep_all = torch.from_numpy(O.dropout_keep(seed, site, np.arange(B * NH * S * S, dtype=np.uint64).astype(np.uint32), p).astype(np.float64) / (1 - p)).view(B, NH, S, S).cuda()
    tab = table.double().requires_grad_(True)
    worst = 0.0
    for b, n in enumerate(lens):
        r0 = row0[b]
        x = qkv[r0:r0 + n].double().view(n, 3, NH, HD).requires_grad_(True)
        q, k, v = (x[:, i].transpose(0, 1) for i in range(3))
        bias = tab[O.mpnet_relative_position_bucket(torch.arange(n)[None, :] - torch.arange(n)[:, None]).cuda()].permute(2, 0, 1)     # [NH, n, n]
        s = q @ k.transpose(-1, -2) / math.sqrt(HD) + bias
        pr = torch.softmax(s, dim=-1) * keep_all[b, :, :n, :n]
        out = (pr @ v).transpose(0, 1).reshape(n, H)
        assert rel_err(ctx[r0:r0 + n], out.detach()) < 8e-3, b
        np.testing.assert_allclose(lse[b, :, :n].cpu().numpy(), torch.logsumexp(s, -1).detach().cpu().numpy(), rtol=1e-4, atol=1e-4)
        out.backward(dctx[r0:r0 + n].double())
        worst = max(worst, rel_err(dqkv[r0:r0 + n], x.grad.reshape(n, 3 * H)))
    assert worst < 2e-2, worst
    assert rel_err(dtable, tab.grad) < 1e-2, rel_err(dtable, tab.grad)
    # a second backward ACCUMULATES into the distance buffer (every layer of the encoder adds to the same one)
    L.check(lib.carel_attention_bwd(C.byref(a), L.current_stream()), "attn bwd rel 2")
    dtable2 = torch.empty((32, NH), device="cuda")
    L.check(lib.carel_relpos_reduce(ddist.data_ptr(), B, rp.data_ptr(), dtable2.data_ptr(), 0, L.current_stream()), "relpos reduce")
    assert rel_err(dtable2, 2 * dtable) < 1e-5
    # bit-reproducible (the gradient by distance used to be summed with LDS and global atomics): five fresh backward passes,
    # with a second stream keeping the memory system busy, give the same bits in the distance buffer, the table gradient and dqkv
    side, junk = torch.cuda.Stream(), torch.empty(32 << 20, device="cuda")
    ref = None
    for it in range(5):
        ddist.zero_(); dqkv.zero_()
        if it % 2:
            with torch.cuda.stream(side):
                junk.add_(1.0)
        L.check(lib.carel_attention_bwd(C.byref(a), L.current_stream()), "attn bwd rel rep")
        L.check(lib.carel_relpos_reduce(ddist.data_ptr(), B, rp.data_ptr(), dtable2.data_ptr(), 0, L.current_stream()), "relpos reduce")
        cur = (ddist.clone(), dtable2.clone(), dqkv.clone())
        if ref is None:
            ref = cur
            assert torch.equal(cur[1], dtable)
        else:
            assert all(torch.equal(x, y) for x, y in zip(cur, ref)), it
    torch.cuda.synchronize()


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_query_row_limit_is_exact_where_the_other_rows_are_dead(p):
    """carel_attn_args.q_rows = 32 (the encoder's [CLS]-only last layer): the first 32 positions of every sample are the only queries.
    Forward: their ctx / lse rows carry the bits of the unrestricted launch, the other rows are not written.  Backward, with dctx zero off
    those rows (what the encoder hands over): every dK / dV / dQ bit of the unrestricted launch -- the skipped query tiles only ever
    contributed exact zeros -- and zeros for the dQ rows past the limit."""
    B, S = 5, 128
    g = torch.Generator().manual_seed(77)
    qkv = (torch.randn((B * S, 3 * H), generator=g) * 1.5).cuda().bfloat16()
    mask = torch.ones((B, S), dtype=torch.int64); mask[1, 100:] = 0; mask[3, 40:] = 0
    mask = mask.cuda()
    dctx = torch.zeros((B * S, H), device="cuda", dtype=torch.bfloat16)
    live = torch.zeros(B * S, dtype=torch.bool, device="cuda").view(B, S)
    live[:, 0] = True; live[:, 17] = True                       # [CLS] and one more row inside the first tile
    dctx.view(B, S, H)[live] = (torch.randn((int(live.sum()), H), generator=g)).cuda().bfloat16()
    drop = (5, 7, 64, p)
    ctx0, lse0, dq0 = run_attn(qkv, mask, B, S, drop, dctx)
    ctx1, lse1, dq1 = run_attn(qkv, mask, B, S, drop, dctx, q_rows=32)
    first = torch.zeros((B, S), dtype=torch.bool, device="cuda"); first[:, :32] = True
    assert torch.equal(ctx1.view(B, S, H)[first], ctx0.view(B, S, H)[first])
    assert bool(torch.isnan(ctx1.view(B, S, H)[~first].float()).all())          # never written
    assert torch.equal(lse1[:, :, :32], lse0[:, :, :32]) and bool(torch.isnan(lse1[:, :, 32:]).all())
    # (+0 against -0 is the only licence: adding the skipped tiles' exact zeros can flip the sign of a zero sum)
    assert torch.equal(dq1.float(), dq0.float())
    assert bool((dq1.view(B, S, 3 * H)[:, 32:, :H] == 0).all())
