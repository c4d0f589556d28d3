"""bf16 MFMA GEMM (forward / dgrad / wgrad forms, every epilogue) against fp64 matmul of the same
bf16-rounded operands.  Tolerance: fp32 accumulation-order noise only (1e-5 relative, written below)."""
import math

import numpy as np
import pytest
import torch

from carel_vae_amd import _lib as L
from oracle import carel_oracle as O
from tests.gpu_util import gemm, rel_err

pytestmark = pytest.mark.gpu
TOL = 2e-5       # fp32 accumulate vs fp64 reference on identical bf16 inputs
TOL_BF16 = 6e-3  # outputs stored as bf16 (2^-8 half-ulp relative)


def _rand(shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).cuda()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 192), (1024, 768, 768), (512, 2304, 768), (256, 768, 3072)])
def test_forward_nt_bias_bf16(M, N, K):
    A, W, b = _rand((M, K), 1, 1).bfloat16(), _rand((N, K), 0.05, 2).bfloat16(), _rand((N,), 0.1, 3)
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    gemm(A, W, L.GEMM_NT, L.EPI_BIAS_BF16, M, N, K, out_bf16=out, bias=b)
    ref = A.double() @ W.double().t() + b.double()
    assert rel_err(out, ref) < TOL_BF16
    # exactness of the fp32 accumulator itself: f32 output path
    outf = torch.empty((M, N), device="cuda", dtype=torch.float32)
    gemm(A, W, L.GEMM_NT, L.EPI_ADD_F32, M, N, K, out_f32=outf)
    assert rel_err(outf, A.double() @ W.double().t()) < TOL


def test_forward_gelu_epilogue():
    M, N, K = 256, 3072, 768
    A, W, b = _rand((M, K), 1, 4).bfloat16(), _rand((N, K), 0.05, 5).bfloat16(), _rand((N,), 0.1, 6)
    u = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    g = torch.empty_like(u)
    gemm(A, W, L.GEMM_NT, L.EPI_BIAS_GELU, M, N, K, out_bf16=u, out2_bf16=g, bias=b)
    ref_u = (A.double() @ W.double().t() + b.double())
    assert rel_err(u, ref_u) < TOL_BF16
    uu = u.double()
    ref_g = 0.5 * uu * (1 + torch.erf(uu / math.sqrt(2)))
    assert rel_err(g, ref_g) < TOL_BF16
    # inference: the pre-activation is optional (nobody reads it without a backward) -- same gelu output, bit for bit
    g2 = torch.empty_like(g)
    gemm(A, W, L.GEMM_NT, L.EPI_BIAS_GELU, M, N, K, out2_bf16=g2, bias=b)
    assert torch.equal(g2, g)


@pytest.mark.parametrize("p,off,M", [(0.0, 5 * 768, 256), (0.1, 5 * 768, 256), (0.1, 5 * 768 + 1, 256), (0.1, 7, 8192)])
def test_forward_dropout_residual_epilogue(p, off, M):
    """(odd element offsets: the epilogues take the multipliers by PAIRS when the group starts on the even half of a pair -- every caller --
    and element by element otherwise; both must be the oracle's mask.  M = 8192: the ping-pong kernel's epilogue, M = 256: the 128x128 one)"""
    N, K = 768, 768
    A, W, b = _rand((M, K), 1, 7).bfloat16(), _rand((N, K), 0.05, 8).bfloat16(), _rand((N,), 0.1, 9)
    r = _rand((M, N), 1, 10)
    out = torch.empty((M, N), device="cuda", dtype=torch.float32)
    seed, site = 1234, O.site_attn_out(2)
    gemm(A, W, L.GEMM_NT, L.EPI_BIAS_DROP_RESID, M, N, K, out_f32=out, bias=b, resid=r, drop=(seed, site, off, p))
    y = (A.double() @ W.double().t() + b.double())
    if p > 0:
        idx = (np.arange(M * N, dtype=np.uint64) + np.uint64(off)).astype(np.uint32)
        keep = O.dropout_keep(seed, site, idx, p).reshape(M, N)
        y = y * torch.from_numpy(keep.astype(np.float64) / (1 - p)).cuda()
        assert abs(keep.mean() - (1 - p)) < 0.01
    assert rel_err(out, y + r.double()) < TOL


@pytest.mark.parametrize("M", [256, 8192])
def test_gelu_derivative_saved_by_the_forward_equals_dgelu_of_the_saved_preactivation(M):
    """What the training encoder runs: FFN1 saves gelu'(u) (EPI_BIAS_GELU_DG) and the FFN2 data gradient multiplies by it
    (EPI_MUL_BF16).  Against the older pair (save u, EPI_DGELU_BF16 evaluates gelu'(u) in backward): the gelu outputs are
    bit-identical, the saved derivative is the bf16 rounding of the fp32 derivative the old backward computed, so du differs
    by that one rounding (2^-9 relative) -- and both agree with fp64 to bf16 accuracy.  Bias-gradient column sums included."""
    N, K = 3072, 768
    A, W, b = _rand((M, K), 1, 71).bfloat16(), _rand((N, K), 0.05, 72).bfloat16(), _rand((N,), 0.1, 73)
    u, g = (torch.empty((M, N), device="cuda", dtype=torch.bfloat16) for _ in range(2))
    gp, g2 = torch.empty_like(u), torch.empty_like(u)
    gemm(A, W, L.GEMM_NT, L.EPI_BIAS_GELU, M, N, K, out_bf16=u, out2_bf16=g, bias=b)
    gemm(A, W, L.GEMM_NT, L.EPI_BIAS_GELU_DG, M, N, K, out_bf16=gp, out2_bf16=g2, bias=b)
    assert torch.equal(g, g2)
    ud = u.double()
    ref_gp = 0.5 * (1 + torch.erf(ud / math.sqrt(2))) + ud * torch.exp(-0.5 * ud * ud) / math.sqrt(2 * math.pi)
    assert float((gp.double() - ref_gp).abs().max()) < 2 ** -8            # |gelu'| <= 1.13: half a bf16 ulp there is 2^-9 * 1.13
    dy, W2 = _rand((M, 768), 1, 74).bfloat16(), _rand((768, N), 0.05, 75).bfloat16()
    du_old, du_new = torch.empty_like(u), torch.empty_like(u)
    cs_old, cs_new = (torch.empty((M // 128, N), device="cuda") for _ in range(2))
    gemm(dy, W2, L.GEMM_NN, L.EPI_DGELU_BF16, M, N, 768, out_bf16=du_old, aux=u, colsum_part=cs_old)
    gemm(dy, W2, L.GEMM_NN, L.EPI_MUL_BF16, M, N, 768, out_bf16=du_new, aux=gp, colsum_part=cs_new)
    ref = (dy.double() @ W2.double()) * ref_gp
    assert rel_err(du_old, ref) < TOL_BF16 and rel_err(du_new, ref) < TOL_BF16
    exact = (dy.double() @ W2.double()) * gp.double()                     # the new epilogue's own definition
    assert rel_err(du_new, exact) < TOL_BF16 / 2
    assert rel_err(cs_new.sum(0), exact.sum(0)) < 1e-4 and rel_err(cs_old.sum(0), ref.sum(0)) < 1e-4


def test_dgrad_nn_forms():
    M, Nout, Nin = 384, 3072, 768            # dX[M,Nin] = dY[M,Nout] @ W[Nout,Nin]
    dY, W = _rand((M, Nout), 1, 11).bfloat16(), _rand((Nout, Nin), 0.05, 12).bfloat16()
    r = _rand((M, Nin), 1, 13)
    out = torch.empty((M, Nin), device="cuda", dtype=torch.float32)
    gemm(dY, W, L.GEMM_NN, L.EPI_ADD_F32, M, Nin, Nout, out_f32=out, resid=r)
    assert rel_err(out, dY.double() @ W.double() + r.double()) < TOL
    ob = torch.empty((M, Nin), device="cuda", dtype=torch.bfloat16)
    gemm(dY, W, L.GEMM_NN, L.EPI_BIAS_BF16, M, Nin, Nout, out_bf16=ob)
    assert rel_err(ob, dY.double() @ W.double()) < TOL_BF16
    # dgrad through GELU: du = (dy @ W2) * gelu'(u)      W2 is [768, 3072]
    dy2, W2 = _rand((M, 768), 1, 14).bfloat16(), _rand((768, 3072), 0.05, 15).bfloat16()
    u = _rand((M, 3072), 1.5, 16).bfloat16()
    du = torch.empty((M, 3072), device="cuda", dtype=torch.bfloat16)
    gemm(dy2, W2, L.GEMM_NN, L.EPI_DGELU_BF16, M, 3072, 768, out_bf16=du, aux=u)
    uu = u.double()
    gp = 0.5 * (1 + torch.erf(uu / math.sqrt(2))) + uu * torch.exp(-0.5 * uu * uu) / math.sqrt(2 * math.pi)
    assert rel_err(du, (dy2.double() @ W2.double()) * gp) < TOL_BF16
    # fused bias-gradient partials: column sums of the (pre-rounding) output per 128-row tile
    part = torch.empty((M // 128, 3072), device="cuda")
    gemm(dy2, W2, L.GEMM_NN, L.EPI_DGELU_BF16, M, 3072, 768, out_bf16=du, aux=u, colsum_part=part)
    ref_cs = ((dy2.double() @ W2.double()) * gp).sum(0)
    assert rel_err(part.sum(0), ref_cs) < 1e-4
    lib = L.load()
    out = torch.empty(3072, device="cuda")
    L.check(lib.carel_partial_reduce_f32(part.data_ptr(), out.data_ptr(), 3072, M // 128, 0, L.current_stream()))
    assert rel_err(out, ref_cs) < 1e-4


@pytest.mark.parametrize("T,Nout,Nin,splits", [(1024, 768, 768, 8), (2048, 2304, 768, 4), (512, 128, 3072, 1)])
def test_wgrad_tn_split_k(T, Nout, Nin, splits):
    dY, X = _rand((T, Nout), 1, 17).bfloat16(), _rand((T, Nin), 1, 18).bfloat16()
    slabs = torch.empty((splits, Nout, Nin), device="cuda", dtype=torch.float32)
    gemm(dY, X, L.GEMM_TN, L.EPI_SLAB_F32, Nout, Nin, T, splits=splits, out_f32=slabs)
    dW = torch.full((Nout, Nin), 7.0, device="cuda", dtype=torch.float32)
    lib = L.load()
    L.check(lib.carel_slab_reduce_f32(slabs.data_ptr(), dW.data_ptr(), Nout * Nin, splits, 0, L.current_stream()))
    ref = dY.double().t() @ X.double()
    assert rel_err(dW, ref) < TOL
    L.check(lib.carel_slab_reduce_f32(slabs.data_ptr(), dW.data_ptr(), Nout * Nin, splits, 1, L.current_stream()))
    assert rel_err(dW, 2 * ref) < TOL
    # bias gradient from the same GEMM: column sums of dY per K-slice (ones-vector MFMA in the first tile column)
    cs = torch.full((splits, Nout), float("nan"), device="cuda")
    gemm(dY, X, L.GEMM_TN, L.EPI_SLAB_F32, Nout, Nin, T, splits=splits, out_f32=slabs, colsum_a=cs)
    assert rel_err(cs.sum(0), dY.double().sum(0)) < TOL
    assert rel_err(slabs.sum(0), ref) < TOL


def test_bad_shapes_are_refused():
    A = torch.zeros((100, 64), device="cuda", dtype=torch.bfloat16)
    out = torch.empty((100, 128), device="cuda", dtype=torch.bfloat16)
    with pytest.raises(L.CarelError):
        gemm(A, A, L.GEMM_NT, L.EPI_BIAS_BF16, 100, 128, 64, out_bf16=out)


@pytest.mark.experiments
@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("form", ["NT", "NN", "TN"])
def test_both_tile_variants_agree(variant, form):
    """The 128x128 and the 256x96 (3-stage pipelined) kernels on a shape both accept; many K steps so that every
    LDS stage is recycled several times (race screen on exact small-integer data: any stale tile shows)."""
    lib = L.load()
    M, N, K = 512, 384, 1024
    g = torch.Generator().manual_seed(5)
    ints = lambda shape: torch.randint(-3, 4, shape, generator=g).float().cuda().bfloat16()
    try:
        L.check(lib.carel_gemm_set_variant(variant))
        if form == "NT":
            A, B = ints((M, K)), ints((N, K))
            out = torch.empty((M, N), device="cuda")
            gemm(A, B, L.GEMM_NT, L.EPI_ADD_F32, M, N, K, out_f32=out)
            ref = A.double() @ B.double().t()
        elif form == "NN":
            A, B = ints((M, K)), ints((K, N))
            out = torch.empty((M, N), device="cuda")
            gemm(A, B, L.GEMM_NN, L.EPI_ADD_F32, M, N, K, out_f32=out)
            ref = A.double() @ B.double()
        else:
            A, B = ints((K, M)), ints((K, N))
            slabs = torch.empty((2, M, N), device="cuda")
            gemm(A, B, L.GEMM_TN, L.EPI_SLAB_F32, M, N, K, splits=2, out_f32=slabs)
            out = slabs.sum(0)
            ref = A.double().t() @ B.double()
        assert torch.equal(out.double(), ref), float((out.double() - ref).abs().max())
    finally:
        L.check(lib.carel_gemm_set_variant(0))


def test_v2_repeated_launches_are_identical():
    """Race screen for the counted-vmcnt pipeline: 20 launches of an encoder-sized GEMM must agree bit for bit."""
    M, N, K = 2048, 768, 3072
    A, W = _rand((M, K), 1, 21).bfloat16(), _rand((N, K), 0.05, 22).bfloat16()
    ref = None
    for _ in range(20):
        out = torch.empty((M, N), device="cuda")
        gemm(A, W, L.GEMM_NT, L.EPI_ADD_F32, M, N, K, out_f32=out)
        if ref is None:
            ref = out.clone()
            assert rel_err(out, A.double() @ W.double().t()) < TOL
        else:
            assert torch.equal(out, ref)


def test_internal_split_k_with_workspace_matches_single_pass():
    """Small grids + a workspace take the split-K + fused-epilogue path; results equal the single-pass kernel
    up to fp32 summation order, for an epilogue with bias, dropout (row-mapped) and residual."""
    M, N, K = 1792, 768, 3072
    A, W, b = _rand((M, K), 1, 31).bfloat16(), _rand((N, K), 0.05, 32).bfloat16(), _rand((N,), 0.1, 33)
    r = _rand((M, N), 1, 34)
    rowmap = torch.randperm(M).to(torch.int32).cuda()
    ws = torch.empty(4 * M * N, device="cuda")
    outs = []
    for use_ws in (None, ws):
        out = torch.empty((M, N), device="cuda")
        gemm(A, W, L.GEMM_NT, L.EPI_BIAS_DROP_RESID, M, N, K, out_f32=out, bias=b, resid=r, drop=(7, 5, 0, 0.1), splitk_ws=use_ws,
             **({"lda": K, "ldb": K}))
        outs.append(out)
    assert rel_err(outs[1], outs[0]) < 1e-6
    assert rel_err(outs[1], A.double() @ W.double().t() + b.double() + r.double()) < 0.4     # dropout on: loose sanity only
    A2, W2 = _rand((M, 2304), 1, 35).bfloat16(), _rand((2304, N), 0.05, 36).bfloat16()
    o1, o2 = torch.empty((M, N), device="cuda"), torch.empty((M, N), device="cuda")
    gemm(A2, W2, L.GEMM_NN, L.EPI_ADD_F32, M, N, 2304, out_f32=o1, resid=r)
    gemm(A2, W2, L.GEMM_NN, L.EPI_ADD_F32, M, N, 2304, out_f32=o2, resid=r, splitk_ws=ws)
    assert rel_err(o2, o1) < 1e-6 and rel_err(o2, A2.double() @ W2.double() + r.double()) < TOL


@pytest.mark.experiments
@pytest.mark.parametrize("M", [1792, 1920, 640])
def test_internal_split_on_the_ping_pong_kernel_is_bitwise_the_128_tile_split(M):
    """Packed ECPE batches (~1.8 k rows): the K slices of an internally split NT / NN GEMM run on the ping-pong kernel (hook 141, the
    default) where tiles x slices fit one round, on the 128x128 kernel otherwise (hook 140).  Same K partition, same order of additions:
    bit-identical results, for every epilogue the encoder uses on that path, including a row count that is not a multiple of 256."""
    lib = L.load()
    N = 768
    ws = torch.empty(8 * M * N, device="cuda")
    r, b = _rand((M, N), 1, 44), _rand((N,), 0.1, 45)
    rowmap = None
    cases = [("NT", L.EPI_BIAS_DROP_RESID, 3072, dict(bias=b, resid=r, drop=(7, 5, 0, 0.1))),
             ("NN", L.EPI_ADD_F32, 3072, dict(resid=r)), ("NN", L.EPI_ADD_F32, 2304, dict(resid=r)),
             ("NN", L.EPI_BIAS_BF16, 3072, dict())]
    for form, epi, K, kw in cases:
        A = _rand((M, K), 1, 41).bfloat16()
        W = (_rand((N, K), 0.05, 42) if form == "NT" else _rand((K, N), 0.05, 43)).bfloat16()
        outs = []
        for hook in (140, 141):
            L.check(lib.carel_gemm_set_variant(hook))
            try:
                of = torch.zeros((M, N), device="cuda")
                ob = torch.zeros((M, N), device="cuda", dtype=torch.bfloat16)
                gemm(A, W, L.GEMM_NT if form == "NT" else L.GEMM_NN, epi, M, N, K, out_f32=of, out_bf16=ob, splitk_ws=ws, **kw)
                outs.append((of.clone(), ob.clone()))
            finally:
                L.check(lib.carel_gemm_set_variant(141))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (form, epi, K)
        ref = (A.double() @ (W.double().t() if form == "NT" else W.double()))
        if epi == L.EPI_ADD_F32:
            assert rel_err(outs[1][0], ref + r.double()) < TOL
        if epi == L.EPI_BIAS_BF16:
            assert rel_err(outs[1][1], ref) < TOL_BF16


# ---------------------------------------------------------------------------------------------------------------
# Ping-pong kernel (gemm_pp.hip: 256 x 96n tiles, two wave groups alternating load / MFMA segments, LDS-DMA in flight
# across raw barriers behind counted vmcnt waits).  Variant 3 forces it; variant 1 forces the 128x128 kernel.  Both
# kernels add the K tiles in the same order with the same MFMA, so on shapes both accept they must agree BIT FOR BIT.
# ---------------------------------------------------------------------------------------------------------------
class _variant:
    """v: kernel choice (1 = 128x128 only, 3 = ping-pong wherever possible); wide: the ping-pong kernel's schedule
    (1 = wide phases, the default; 0 = the fine 12-MFMA phases)"""
    def __init__(self, v, wide=1): self.v, self.wide = v, wide
    def __enter__(self):
        L.check(L.load().carel_gemm_set_variant(self.v))
        L.check(L.load().carel_gemm_set_variant(90 + self.wide))
    def __exit__(self, *a):
        L.check(L.load().carel_gemm_set_variant(0))
        L.check(L.load().carel_gemm_set_variant(91))


def _ints(shape, seed, lo=-3, hi=4):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).float().cuda().bfloat16()


@pytest.mark.parametrize("wide", [1, 0])
@pytest.mark.parametrize("form", ["NT", "NN"])
@pytest.mark.parametrize("M,N,K", [(8192, 768, 768), (8192, 2304, 768), (8192, 3072, 768), (8192, 768, 3072), (8192, 768, 2304),
                                   (256, 96, 256), (2176, 192, 320), (1000, 288, 448), (3000, 576, 1024), (768, 192, 256), (2048, 960, 512)])
@pytest.mark.experiments
def test_pp_exact_integers(form, M, N, K, wide):
    """Exact small-integer data at the production shapes (M = 8192 x {768, 2304, 3072}) and at ragged M (edge rows are
    masked) / every npn: any stale LDS tile, wrong fragment map or missed k step shows as an integer difference."""
    A = _ints((M, K), 1)
    B = _ints((N, K), 2) if form == "NT" else _ints((K, N), 2)
    out = torch.full((M + 8, N), 7.0, device="cuda")          # 8 guard rows: nothing past row M may be written
    with _variant(3, wide):
        gemm(A, B, L.GEMM_NT if form == "NT" else L.GEMM_NN, L.EPI_ADD_F32, M, N, K, out_f32=out)
    ref = A.double() @ (B.double().t() if form == "NT" else B.double())
    assert torch.equal(out[:M].double(), ref), float((out[:M].double() - ref).abs().max())
    assert torch.equal(out[M:], torch.full((8, N), 7.0, device="cuda"))


@pytest.mark.experiments
@pytest.mark.parametrize("wide", [1, 0])
@pytest.mark.parametrize("N,K", [(768, 768), (2304, 768), (3072, 768), (768, 3072)])
def test_pp_bitwise_equals_128_tile_every_epilogue(N, K, wide):
    M = 1024
    A, W, b = _rand((M, K), 1, 41).bfloat16(), _rand((N, K), 0.05, 42).bfloat16(), _rand((N,), 0.1, 43)
    Wn = _rand((K, N), 0.05, 44).bfloat16()
    r, u = _rand((M, N), 1, 45), _rand((M, N), 1.5, 46).bfloat16()
    res = {}
    for v in (1, 3):
        with _variant(v, wide):
            o = {}
            o["qkv"] = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
            gemm(A, W, L.GEMM_NT, L.EPI_BIAS_BF16, M, N, K, out_bf16=o["qkv"], bias=b)
            o["u"], o["g"] = torch.empty_like(o["qkv"]), torch.empty_like(o["qkv"])
            gemm(A, W, L.GEMM_NT, L.EPI_BIAS_GELU, M, N, K, out_bf16=o["u"], out2_bf16=o["g"], bias=b)
            o["gp"], o["g2"] = torch.empty_like(o["qkv"]), torch.empty_like(o["qkv"])
            gemm(A, W, L.GEMM_NT, L.EPI_BIAS_GELU_DG, M, N, K, out_bf16=o["gp"], out2_bf16=o["g2"], bias=b)
            o["h"] = torch.empty((M, N), device="cuda")
            gemm(A, W, L.GEMM_NT, L.EPI_BIAS_DROP_RESID, M, N, K, out_f32=o["h"], bias=b, resid=r, drop=(9, 5, 3 * N, 0.1))
            o["dx"] = torch.empty((M, N), device="cuda")
            gemm(A, Wn, L.GEMM_NN, L.EPI_ADD_F32, M, N, K, out_f32=o["dx"], resid=r)
            o["dc"] = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
            gemm(A, Wn, L.GEMM_NN, L.EPI_BIAS_BF16, M, N, K, out_bf16=o["dc"])
            if N % 192 == 0:           # the fused column sums pair two 16-column fragments: even npn
                o["du"], o["cs"] = torch.empty((M, N), device="cuda", dtype=torch.bfloat16), torch.empty((M // 128, N), device="cuda")
                gemm(A, Wn, L.GEMM_NN, L.EPI_DGELU_BF16, M, N, K, out_bf16=o["du"], aux=u, colsum_part=o["cs"])
                o["dm"], o["cs_m"] = torch.empty((M, N), device="cuda", dtype=torch.bfloat16), torch.empty((M // 128, N), device="cuda")
                gemm(A, Wn, L.GEMM_NN, L.EPI_MUL_BF16, M, N, K, out_bf16=o["dm"], aux=u, colsum_part=o["cs_m"])
            res[v] = o
    assert torch.equal(res[3]["g2"], res[3]["g"])           # the gelu output does not depend on which companion is saved
    for k in res[1]:
        if k in ("cs", "cs_m"):       # column sums: same addends, different summation tree
            assert rel_err(res[3][k], res[1][k]) < 1e-5
        else:
            assert torch.equal(res[3][k], res[1][k]), (k, float((res[3][k].float() - res[1][k].float()).abs().max()))


@pytest.mark.experiments
def test_gelu_table_of_the_ping_pong_epilogue_equals_the_arithmetic_on_every_bf16_value():
    """The ping-pong kernel's GELU epilogues look gelu(u) / gelu'(u) up in a table indexed by the bf16 bits of u (gemm_epilogue.h,
    gelu_lut8); the 128x128 kernel evaluates erf and exp.  A GEMM whose pre-activations run through EVERY finite bf16 value (one-hot A
    rows pick one B entry each) must give identical bits from both, for both GELU epilogues -- in-range values (table) and the rest
    (the arithmetic fall-back inside the ping-pong kernel) alike."""
    M, N, K = 256, 384, 256
    A = torch.zeros((M, K)); A[torch.arange(M), torch.arange(M) % K] = 1.0
    bits = ((torch.arange(K).view(K, 1) * N + torch.arange(N).view(1, N)) & 0xFFFF).to(torch.int32)
    ex = (bits >> 7) & 0xFF
    bits = torch.where((ex == 0xFF) | (ex == 0), torch.zeros_like(bits), bits)              # no Inf / NaN patterns; the MFMA flushes bf16 denormals
    vals = (bits << 16).view(torch.float32)                                                  # [K, N]: the bf16 value with those bits
    assert torch.unique(bits).numel() == 65536 - 4 * 128 + 1                                 # every normal bf16 value, and zero
    A, W = A.cuda().bfloat16(), vals.t().contiguous().cuda().bfloat16()                      # NT: W is [N, K]
    assert torch.equal(W.float().t().cpu(), vals)
    res = {}
    for v in (1, 3):
        with _variant(v):
            u, g = torch.empty((M, N), device="cuda", dtype=torch.bfloat16), torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
            gemm(A, W, L.GEMM_NT, L.EPI_BIAS_GELU, M, N, K, out_bf16=u, out2_bf16=g)
            dg, g2 = torch.empty_like(u), torch.empty_like(u)
            gemm(A, W, L.GEMM_NT, L.EPI_BIAS_GELU_DG, M, N, K, out_bf16=dg, out2_bf16=g2)
            res[v] = (u, g, dg, g2)
    assert torch.equal(res[3][0].view(torch.int16).cpu(), vals.bfloat16().view(torch.int16))  # the pre-activations really are those values
    for a, b, name in zip(res[3], res[1], ("u", "gelu", "gelu'", "gelu (with gelu')")):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), name
    L.check(L.load().carel_gemm_set_variant(160))             # the ping-pong kernel with the table switched off: same bits again
    try:
        with _variant(3):
            dg, g2 = torch.empty_like(res[3][0]), torch.empty_like(res[3][0])
            gemm(A, W, L.GEMM_NT, L.EPI_BIAS_GELU_DG, M, N, K, out_bf16=dg, out2_bf16=g2)
    finally:
        L.check(L.load().carel_gemm_set_variant(161))
    assert torch.equal(dg.view(torch.int16), res[3][2].view(torch.int16)) and torch.equal(g2.view(torch.int16), res[3][3].view(torch.int16))
    ref = torch.nn.functional.gelu(vals.double())
    assert float((res[3][1].double().cpu() - ref).abs().max() / 1.0) < 0.07 and rel_err(res[3][1], ref.float()) < TOL_BF16   # and they are GELU


@pytest.mark.experiments
@pytest.mark.parametrize("wide", [1, 0])
def test_pp_race_screen_repeated_launches(wide):
    """20 launches of each production shape on fresh random data, compared with the first launch bit for bit while a
    second stream keeps the memory system busy (uneven load is what exposes a too-early LDS read)."""
    M = 8192
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, device="cuda")
    with _variant(3, wide):
        for (N, K, form) in [(2304, 768, "NT"), (3072, 768, "NT"), (768, 3072, "NT"), (3072, 768, "NN"), (768, 2304, "NN")]:
            A = _rand((M, K), 1, 51).bfloat16()
            B = (_rand((N, K), 0.05, 52) if form == "NT" else _rand((K, N), 0.05, 52)).bfloat16()
            ref = None
            for it in range(20):
                if it % 3 == 0:
                    with torch.cuda.stream(side):
                        junk.add_(1.0)
                out = torch.empty((M, N), device="cuda")
                gemm(A, B, L.GEMM_NT if form == "NT" else L.GEMM_NN, L.EPI_ADD_F32, M, N, K, out_f32=out)
                if ref is None:
                    ref = out.clone()
                    assert rel_err(out, A.double() @ (B.double().t() if form == "NT" else B.double())) < TOL
                else:
                    assert torch.equal(out, ref), (N, K, form, it)
    torch.cuda.synchronize()


@pytest.mark.parametrize("T,Nout,Nin,splits", [(8192, 768, 3072, 5), (8192, 3072, 768, 5), (8192, 768, 768, 16), (8192, 2304, 768, 7),
                                              (1920, 768, 768, 3), (1024, 256, 96, 1), (4096, 512, 288, 4)])
@pytest.mark.experiments
def test_pp_wgrad_exact_integers_uneven_slices(T, Nout, Nin, splits):
    """A^T B form of the ping-pong kernel (weight gradients): K tiles dealt unevenly to the z slices (T/64 is not a
    multiple of `splits`), exact integer data, slabs summed against the fp64 product; the bias gradient (column sums
    of dY per slice, ones-vector MFMA) comes out of the same launch."""
    dY, X = _ints((T, Nout), 61, -2, 3), _ints((T, Nin), 62, -2, 3)
    slabs = torch.full((splits, Nout, Nin), float("nan"), device="cuda")
    cs = torch.full((splits, Nout), float("nan"), device="cuda")
    with _variant(3):
        gemm(dY, X, L.GEMM_TN, L.EPI_SLAB_F32, Nout, Nin, T, splits=splits, out_f32=slabs, colsum_a=cs)
    ref = dY.double().t() @ X.double()
    assert torch.equal(slabs.double().sum(0), ref), float((slabs.double().sum(0) - ref).abs().max())
    assert torch.equal(cs.double().sum(0), dY.double().sum(0))
    # slice z covers K tiles [z*nk/splits, (z+1)*nk/splits)
    nk = T // 64
    z = splits - 1
    k0, k1 = (z * nk // splits) * 64, ((z + 1) * nk // splits) * 64
    assert torch.equal(slabs[z].double(), dY[k0:k1].double().t() @ X[k0:k1].double())


@pytest.mark.experiments
def test_pp_wgrad_equal_slices_bitwise_equal_128_tile():
    T, Nout, Nin, splits = 4096, 768, 768, 4
    dY, X = _rand((T, Nout), 1, 63).bfloat16(), _rand((T, Nin), 1, 64).bfloat16()
    out = {}
    for v in (1, 3):
        slabs = torch.empty((splits, Nout, Nin), device="cuda")
        cs = torch.empty((splits, Nout), device="cuda")
        with _variant(v):
            gemm(dY, X, L.GEMM_TN, L.EPI_SLAB_F32, Nout, Nin, T, splits=splits, out_f32=slabs, colsum_a=cs)
        out[v] = (slabs, cs)
    assert torch.equal(out[1][0], out[3][0]) and torch.equal(out[1][1], out[3][1])


@pytest.mark.experiments
def test_wgrad_splits_helper_matches_what_the_kernels_accept():
    lib = L.load()
    for (M, N) in [(768, 3072), (3072, 768), (768, 768), (2304, 768)]:
        for T in (8192, 4096, 1920, 1664, 1024, 640, 512, 128):          # packed batches give any multiple of 128 tokens
            for v in (0, 1, 3):
                with _variant(v):
                    s = lib.carel_gemm_wgrad_splits(M, N, T)
                    assert 1 <= s <= 16
                    dY, X = _rand((T, M), 1, 65).bfloat16(), _rand((T, N), 1, 66).bfloat16()
                    slabs = torch.empty((s, M, N), device="cuda")
                    gemm(dY, X, L.GEMM_TN, L.EPI_SLAB_F32, M, N, T, splits=s, out_f32=slabs)
                    if T <= 1664 and v != 1:
                        assert rel_err(slabs.sum(0), dY.double().t() @ X.double()) < TOL


@pytest.mark.experiments
@pytest.mark.parametrize("M", [8192, 1000, 8000 - 37])
def test_pp_epilogue_inputs_requested_before_the_main_loop_same_bits(M):
    """Round 3: the ping-pong kernel requests the residual rows / saved gelu'(u) values of its epilogue right behind the prologue's DMA
    units (all four row blocks at N = 768, two at N = 3072) and widens the counted vmcnt waits of K tile 0 by those loads
    (tools/gemm_sched.py: first_tile_waits_target_prologue).  With the hook off (170) the inputs are loaded after the main loop as in
    round 2: identical bits for every epilogue with inputs, at the production shapes and at ragged M (clamped rows, guard rows
    untouched); each against fp64 as well.  Several launches per setting on fresh buffers: a wait that was widened one tile too long
    would read a stale LDS stage in some of them."""
    lib = L.load()
    cases = [("out NT", L.GEMM_NT, L.EPI_BIAS_DROP_RESID, 768, 768), ("ffn2 NT", L.GEMM_NT, L.EPI_BIAS_DROP_RESID, 768, 3072),
             ("ffn1 dgrad NN", L.GEMM_NN, L.EPI_ADD_F32, 768, 3072), ("qkv dgrad NN", L.GEMM_NN, L.EPI_ADD_F32, 768, 2304),
             ("ffn2 dgrad NN", L.GEMM_NN, L.EPI_MUL_BF16, 3072, 768), ("ffn2 dgrad erf NN", L.GEMM_NN, L.EPI_DGELU_BF16, 3072, 768),
             ("short K NT", L.GEMM_NT, L.EPI_BIAS_DROP_RESID, 768, 256)]
    for name, form, epi, N, K in cases:
        A = _rand((M, K), 1, 61).bfloat16()
        B = (_rand((N, K), 0.05, 62) if form == L.GEMM_NT else _rand((K, N), 0.05, 62)).bfloat16()
        bias, resid, aux = _rand((N,), 0.1, 63), _rand((M, N), 1, 64), _rand((M, N), 1.0, 65).bfloat16()
        outs = {}
        for hook in (171, 170):
            L.check(lib.carel_gemm_set_variant(hook))
            try:
                with _variant(3, 1):
                    for rep in range(3):
                        of = torch.full((M + 8, N), 7.0, device="cuda")
                        ob = torch.full((M + 8, N), 7.0, device="cuda", dtype=torch.bfloat16)
                        kw = dict(out_f32=of, out_bf16=ob)
                        if epi == L.EPI_BIAS_DROP_RESID: kw.update(bias=bias, resid=resid, drop=(9, 5, 3 * N, 0.1))
                        elif epi == L.EPI_ADD_F32: kw.update(resid=resid)
                        else: kw.update(aux=aux, colsum_part=torch.empty(((M + 127) // 128, N), device="cuda"))
                        gemm(A, B, form, epi, M, N, K, **kw)
                        o = of if epi in (L.EPI_BIAS_DROP_RESID, L.EPI_ADD_F32) else ob
                        assert torch.equal(o[M:].float(), torch.full((8, N), 7.0, device="cuda")), (name, hook, "guard rows written")
                        if (hook, 0) in outs:
                            assert torch.equal(o[:M], outs[(hook, 0)]), (name, hook, rep)
                        else:
                            outs[(hook, 0)] = o[:M].clone()
            finally:
                L.check(lib.carel_gemm_set_variant(171))
        assert torch.equal(outs[(171, 0)], outs[(170, 0)]), (name, float((outs[(171, 0)].float() - outs[(170, 0)].float()).abs().max()))
        prod = A.double() @ (B.double().t() if form == L.GEMM_NT else B.double())
        if epi == L.EPI_ADD_F32:
            assert rel_err(outs[(171, 0)], prod + resid.double()) < TOL
        elif epi == L.EPI_MUL_BF16:
            assert rel_err(outs[(171, 0)], prod * aux.double()) < TOL_BF16


def _rowln(A, W, bias, resid, gamma, beta, eps, M, K, drop=(0, 0, 0, 0.0), want_h=True, packed=False):
    import ctypes as C
    a = L.GemmRowLnArgs()
    if packed:
        Wp = torch.empty(768 * K, device="cuda", dtype=torch.bfloat16)
        L.check(L.load().carel_gemm_rowln_pack(W.data_ptr(), K, K, Wp.data_ptr(), L.current_stream()), "carel_gemm_rowln_pack")
        # the documented order: [n / 16][k / 64][(k / 32) % 2][(k / 8) % 4][n % 16][k % 8]
        ref = W.view(48, 16, K // 64, 2, 4, 8).permute(0, 2, 3, 4, 1, 5).contiguous().view(-1)
        assert torch.equal(Wp.view(torch.int16), ref.view(torch.int16))
        W = Wp
    a.w_packed = 1 if packed else 0
    a.A, a.W, a.lda, a.ldb, a.M, a.K = A.data_ptr(), W.data_ptr(), K, K, M, K
    a.bias, a.resid_f32, a.gamma, a.beta, a.eps = bias.data_ptr(), resid.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps
    out = dict(h=torch.full((M + 4, 768), 7.0, device="cuda"), xf=torch.full((M + 4, 768), 7.0, device="cuda"),
               xb=torch.full((M + 4, 768), 7.0, device="cuda", dtype=torch.bfloat16), st=torch.full((M + 4, 2), 7.0, device="cuda"))
    a.h_f32 = out["h"].data_ptr() if want_h else None
    a.x_f32, a.x_bf16, a.stats = out["xf"].data_ptr(), out["xb"].data_ptr(), out["st"].data_ptr()
    a.drop_seed, a.drop_site, a.drop_idx_offset, a.drop_p = drop
    a.drop_row_map = None
    L.check(L.load().carel_gemm_rowln(C.byref(a), L.current_stream()), "carel_gemm_rowln")
    return out


@pytest.mark.experiments
@pytest.mark.parametrize("packed", [True, False])
@pytest.mark.parametrize("M,K", [(8192, 768), (8192, 3072), (6144 + 32 * 3 + 5, 768), (256, 256), (128, 2304)])
def test_rowln_equals_gemm_then_layernorm_bitwise(M, K, packed):
    """gemm_rowln.hip (round 3): linear + bias + dropout + residual + LayerNorm with 32 complete rows per workgroup, against the
    two-kernel path it replaces in the dense encoder -- carel_gemm_bf16(CAREL_EPI_BIAS_DROP_RESID) on the ping-pong / 128x128 kernel, then
    carel_layernorm_fwd: the pre-LayerNorm sum, the normalised rows (f32 and bf16) and the row statistics must be identical BIT FOR BIT
    (same MFMA sequence per accumulator, the epilogue expression of epi_out8, the lane map and summation tree of ln_fwd_kernel), with
    dropout on, for the row-major weight and for the packed operand order of carel_gemm_rowln_pack (checked against its documented index map); ragged M (rows past M clamped on load, never stored: guard rows stay untouched), K = one chunk, several chunks, a short
    last chunk; and against fp64."""
    lib = L.load()
    A = _rand((M, K), 1, 71).bfloat16()
    W = _rand((768, K), 0.05, 72).bfloat16()
    bias, resid = _rand((768,), 0.1, 73), _rand((M, 768), 1, 74)
    gamma, beta = 1.0 + _rand((768,), 0.1, 75), _rand((768,), 0.1, 76)
    drop = (9, 5, 3 * 768, 0.1)
    got = _rowln(A, W, bias, resid, gamma, beta, 1e-12, M, K, drop, packed=packed)
    h = torch.empty((M, 768), device="cuda")
    gemm(A, W, L.GEMM_NT, L.EPI_BIAS_DROP_RESID, M, 768, K, out_f32=h, bias=bias, resid=resid, drop=drop)
    xf, xb, st = torch.empty((M, 768), device="cuda"), torch.empty((M, 768), device="cuda", dtype=torch.bfloat16), torch.empty((M, 2), device="cuda")
    L.check(lib.carel_layernorm_fwd(h.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-12, M, 768, xf.data_ptr(), xb.data_ptr(), st.data_ptr(),
                                    L.current_stream()))
    assert torch.equal(got["h"][:M], h), float((got["h"][:M] - h).abs().max())
    assert torch.equal(got["xf"][:M], xf), float((got["xf"][:M] - xf).abs().max())
    assert torch.equal(got["xb"][:M].view(torch.int16), xb.view(torch.int16))
    assert torch.equal(got["st"][:M], st)
    for k in ("h", "xf", "st"):
        assert bool((got[k][M:] == 7.0).all()), k + ": guard rows written"
    assert bool((got["xb"][M:].float() == 7.0).all())
    # without dropout against fp64, and the optional h output switched off
    got0 = _rowln(A, W, bias, resid, gamma, beta, 1e-12, M, K, want_h=False, packed=packed)
    ref_h = A.double() @ W.double().t() + bias.double() + resid.double()
    ref = torch.nn.functional.layer_norm(ref_h, (768,), gamma.double(), beta.double(), 1e-12)
    assert rel_err(got0["xf"][:M], ref) < 2e-5
    assert bool((got0["h"] == 7.0).all())


@pytest.mark.experiments
@pytest.mark.parametrize("case", ["ffn2 fwd NT", "ffn1 dgrad NN", "qkv dgrad NN"])
def test_pp_pair_split_k_exact_and_repeatable(case):
    """Pair split-K (round 3): the N = 768, K >= 1536 GEMMs at M = 8192 run as 128 tiles of 256 x 192, two workgroups per tile, each half of
    K; the second waits (one-directionally, per wave) for the first's partial sums in a zero-initialised workspace whose last 4 KiB hold
    the flags.  Exact small-integer data: any missed / stale / double-counted partial sum shows as an integer difference; random data
    against fp64 and against the one-workgroup kernel (hook 200, the default: same addends, different tree -> 2e-6); 30 launches on one workspace --
    each launch must see ITS partner's sums (sequence numbers), never the previous launch's -- bit-identical, with a second stream busy;
    guard rows untouched; without the caller's zero-fill promise the path is not taken."""
    lib = L.load()
    L.check(lib.carel_gemm_set_variant(201))          # the path is off by default (measured slower than the 256 x 96 tiling: gemm_pp.hip)
    try:
        _pair_case(case, lib)
    finally:
        L.check(lib.carel_gemm_set_variant(200))


def _pair_case(case, lib):
    M, N = 8192, 768
    form, epi, K = {"ffn2 fwd NT": (L.GEMM_NT, L.EPI_BIAS_DROP_RESID, 3072), "ffn1 dgrad NN": (L.GEMM_NN, L.EPI_ADD_F32, 3072),
                    "qkv dgrad NN": (L.GEMM_NN, L.EPI_ADD_F32, 2304)}[case]
    ws = torch.zeros(96 << 20, dtype=torch.uint8, device="cuda")            # zero-filled ONCE
    def run(A, B, resid, bias, out, zeroed=True, drop=(0, 0, 0, 0.0)):
        kw = dict(out_f32=out, resid=resid, splitk_ws=ws, ws_zeroed=zeroed)
        if epi == L.EPI_BIAS_DROP_RESID: kw.update(bias=bias, drop=drop)
        gemm(A, B, form, epi, M, N, K, **kw)
    # exact integers
    A = _ints((M, K), 1)
    B = _ints((N, K), 2) if form == L.GEMM_NT else _ints((K, N), 2)
    resid, bias = _ints((M, N), 3).float(), _ints((N,), 4).float()
    out = torch.full((M + 8, N), 7.0, device="cuda")
    run(A, B, resid, bias, out)
    ref = A.double() @ (B.double().t() if form == L.GEMM_NT else B.double()) + resid.double() + (bias.double() if epi == L.EPI_BIAS_DROP_RESID else 0)
    assert torch.equal(out[:M].double(), ref), float((out[:M].double() - ref).abs().max())
    assert bool((out[M:] == 7.0).all())
    flags = ws[-4096:].view(torch.int32)
    assert int((flags != 0).sum()) == 128 * 8 and int(flags.max()) == int(flags.min())        # the pair path ran: every (tile, wave) flag = this launch's number
    # random data: fp64, the one-workgroup kernel, repeatability across launches that re-use the workspace
    A = _rand((M, K), 1, 81).bfloat16()
    B = (_rand((N, K), 0.05, 82) if form == L.GEMM_NT else _rand((K, N), 0.05, 82)).bfloat16()
    resid, bias = _rand((M, N), 1, 83), _rand((N,), 0.1, 84)
    drop = (9, 5, 3 * N, 0.1)
    side, junk = torch.cuda.Stream(), torch.empty(32 << 20, device="cuda")
    first = None
    for it in range(30):
        if it % 3 == 0:
            with torch.cuda.stream(side):
                junk.add_(1.0)
        o = torch.empty((M, N), device="cuda")
        run(A, B, resid, bias, o, drop=drop)
        if first is None:
            first = o.clone()
        else:
            assert torch.equal(o, first), it
    torch.cuda.synchronize()
    L.check(lib.carel_gemm_set_variant(200))
    try:
        single = torch.empty((M, N), device="cuda")
        run(A, B, resid, bias, single, drop=drop)
    finally:
        L.check(lib.carel_gemm_set_variant(201))          # back on for the rest of this case
    assert rel_err(first, single) < 2e-6 and not torch.equal(first, single)          # same addends, (K/2) + (K/2) instead of one chain
    before = ws[-4096:].clone()
    unz = torch.empty((M, N), device="cuda")
    run(A, B, resid, bias, unz, zeroed=False, drop=drop)                                # no promise, no pair path: the one-workgroup bits, flags untouched
    assert torch.equal(unz, single) and torch.equal(ws[-4096:], before)
    if epi == L.EPI_ADD_F32:
        assert rel_err(first, A.double() @ B.double() + resid.double()) < TOL


@pytest.mark.experiments
@pytest.mark.parametrize("M,K,epi", [(8192, 3072, "drop_resid"), (8192, 768, "bias"), (7936, 768, "add_f32"), (7999, 2304, "drop_resid")])
def test_three_group_kernel_has_the_ping_pong_kernels_bits(M, K, epi):
    """gemm_tri.hip (hook 221, off by default: an experiment that measured no faster -- its header): the 256 x 96 tile by twelve waves in
    three rotating groups adds each accumulator's K tiles in the ping-pong kernel's order, so every output bit is that kernel's; exact
    small integers against fp64 as well (a stage overwritten early or read late shows as an integer difference); ragged M; guard rows."""
    lib = L.load()
    N = 768
    code = {"drop_resid": L.EPI_BIAS_DROP_RESID, "bias": L.EPI_BIAS_BF16, "add_f32": L.EPI_ADD_F32}[epi]
    A, B = _rand((M, K), 1, 91).bfloat16(), _rand((N, K), 0.05, 92).bfloat16()
    resid, bias = _rand((M, N), 1, 93), _rand((N,), 0.1, 94)
    outs = {}
    for hook in (220, 221):
        L.check(lib.carel_gemm_set_variant(hook))
        try:
            of = torch.full((M + 8, N), 7.0, device="cuda"); ob = torch.full((M + 8, N), 7.0, device="cuda").bfloat16()
            gemm(A, B, L.GEMM_NT, code, M, N, K, out_f32=of, out_bf16=ob, bias=bias, resid=resid, drop=(9, 5, 3 * N, 0.1))
            torch.cuda.synchronize()
            outs[hook] = (of, ob)
        finally:
            L.check(lib.carel_gemm_set_variant(220))
    assert torch.equal(outs[220][0], outs[221][0]) and torch.equal(outs[220][1], outs[221][1])
    assert bool((outs[221][0][M:] == 7.0).all()) and bool((outs[221][1][M:].float() == 7.0).all())
    if epi == "add_f32":
        Ai, Bi, ri = _ints((M, K), 1), _ints((N, K), 2), _ints((M, N), 3).float()
        L.check(lib.carel_gemm_set_variant(221))
        try:
            of = torch.empty((M, N), device="cuda")
            gemm(Ai, Bi, L.GEMM_NT, code, M, N, K, out_f32=of, resid=ri)
        finally:
            L.check(lib.carel_gemm_set_variant(220))
        assert torch.equal(of.double(), Ai.double() @ Bi.double().t() + ri.double())


@pytest.mark.experiments
@pytest.mark.parametrize("M,K,hook", [(8192, 768, 0), (8192, 3072, 0), (7999, 768, 0), (1664, 3072, 0), (1664, 768, 0), (384, 768, 0), (8192, 3072, 221)])
def test_residual_recomputed_from_pre_layernorm_rows_same_bits(M, K, hook):
    """carel_gemm_args.resid_ln_* (ABI 6): the residual epilogue given the PRE-LayerNorm rows h, the row statistics and gamma / beta must
    produce the bits it produces when given carel_layernorm_fwd's stored f32 output of the same h -- on the ping-pong kernel (M = 8192 /
    ragged), the split-K slab path with its separate epilogue kernel and the 128x128 kernel (packed-ECPE row counts; they get a workspace),
    and the three-group kernel (hook 221).  Dropout on, guard rows untouched."""
    lib = L.load()
    N = 768
    A, B = _rand((M, K), 1, 61).bfloat16(), _rand((N, K), 0.05, 62).bfloat16()
    h = _rand((M, N), 2.0, 63) + 0.5
    gamma, beta, bias = _rand((N,), 1.0, 64) + 1.0, _rand((N,), 0.3, 65), _rand((N,), 0.1, 66)
    xf = torch.empty((M, N), device="cuda"); st = torch.empty((M, 2), device="cuda")
    L.check(lib.carel_layernorm_fwd(h.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-12, M, 768, xf.data_ptr(), None, st.data_ptr(), L.current_stream()))
    ws = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
    outs = []
    if hook: L.check(lib.carel_gemm_set_variant(hook))
    try:
        for resid, rl in ((xf, None), (h, (st, gamma, beta))):
            out = torch.full((M + 8, N), 7.0, device="cuda")
            a = dict(out_f32=out, bias=bias, resid=resid, drop=(9, 5, 64, 0.1), splitk_ws=ws, resid_ln=rl)
            gemm(A, B, L.GEMM_NT, L.EPI_BIAS_DROP_RESID, M, N, K, **a)
            torch.cuda.synchronize()
            outs.append(out)
    finally:
        if hook: L.check(lib.carel_gemm_set_variant(220))
    assert torch.equal(outs[0], outs[1])
    assert bool((outs[1][M:] == 7.0).all())
    # sanity against fp64, dropout off
    out = torch.empty((M, N), device="cuda")
    if hook: L.check(lib.carel_gemm_set_variant(hook))
    try:
        gemm(A, B, L.GEMM_NT, L.EPI_BIAS_DROP_RESID, M, N, K, out_f32=out, bias=bias, resid=h, splitk_ws=ws, resid_ln=(st, gamma, beta))
    finally:
        if hook: L.check(lib.carel_gemm_set_variant(220))
    h64 = h.double()
    x64 = (h64 - h64.mean(1, keepdim=True)) / torch.sqrt(h64.var(1, unbiased=False, keepdim=True) + 1e-12) * gamma.double() + beta.double()
    ref = A.double() @ B.double().t() + bias.double() + x64
    assert float((out.double() - ref).abs().max()) < 2e-2


class _small_m:
    """the small-M kernel (gemm_sm.hip) on (hook 342) / off (340) for the duration of the block"""
    def __init__(self, on): self.on = on
    def __enter__(self): L.check(L.load().carel_gemm_set_variant(342 if self.on else 340))
    def __exit__(self, *a): L.check(L.load().carel_gemm_set_variant(340))


@pytest.mark.experiments
@pytest.mark.parametrize("M,N,K", [(1664, 768, 768), (1664, 2304, 768), (1792, 768, 3072), (2048, 768, 2304), (256, 1536, 768), (1664, 1536, 192)])
def test_small_m_kernel_bitwise_equals_the_128_tile_kernel_every_epilogue(M, N, K):
    """gemm_sm.hip (128 x 128 tiles, eight waves, four-stage LDS-DMA ring, counted vmcnt: packed ECPE row counts) against the 128x128 kernel of
    gemm.hip (variant 1) on the same calls: the K loop adds in the same order, so every output is bit-identical -- single pass (no workspace).
    Covers K tiles fewer than the ring's stages (K = 192: three tiles) and a two-row-tile grid."""
    A, W, b = _rand((M, K), 1, 41).bfloat16(), _rand((N, K), 0.05, 42).bfloat16(), _rand((N,), 0.1, 43)
    Wn = _rand((K, N), 0.05, 44).bfloat16()
    r, u = _rand((M, N), 1, 45), _rand((M, N), 1.5, 46).bfloat16()
    st, gm, bt = torch.stack([_rand((M,), 0.2, 47), 1 + _rand((M,), 0.1, 48).abs()], 1).contiguous(), 1 + _rand((N,), 0.1, 49), _rand((N,), 0.1, 50)
    res = {}
    for on in (False, True):
        with _variant(1 if not on else 0), _small_m(on):
            o = {}
            o["qkv"] = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
            gemm(A, W, L.GEMM_NT, L.EPI_BIAS_BF16, M, N, K, out_bf16=o["qkv"], bias=b)
            o["u"], o["g"] = torch.empty_like(o["qkv"]), torch.empty_like(o["qkv"])
            gemm(A, W, L.GEMM_NT, L.EPI_BIAS_GELU, M, N, K, out_bf16=o["u"], out2_bf16=o["g"], bias=b)
            o["gp"], o["g2"] = torch.empty_like(o["qkv"]), torch.empty_like(o["qkv"])
            gemm(A, W, L.GEMM_NT, L.EPI_BIAS_GELU_DG, M, N, K, out_bf16=o["gp"], out2_bf16=o["g2"], bias=b)
            o["h"] = torch.empty((M, N), device="cuda")
            gemm(A, W, L.GEMM_NT, L.EPI_BIAS_DROP_RESID, M, N, K, out_f32=o["h"], bias=b, resid=r, drop=(9, 5, 3 * N, 0.1))
            o["hl"] = torch.empty((M, N), device="cuda")
            gemm(A, W, L.GEMM_NT, L.EPI_BIAS_DROP_RESID, M, N, K, out_f32=o["hl"], bias=b, resid=r, drop=(9, 5, 3 * N, 0.1), resid_ln=(st, gm, bt))
            o["dx"] = torch.empty((M, N), device="cuda")
            gemm(A, Wn, L.GEMM_NN, L.EPI_ADD_F32, M, N, K, out_f32=o["dx"], resid=r)
            o["dc"] = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
            gemm(A, Wn, L.GEMM_NN, L.EPI_BIAS_BF16, M, N, K, out_bf16=o["dc"])
            o["du"], o["cs"] = torch.empty((M, N), device="cuda", dtype=torch.bfloat16), torch.empty((M // 128, N), device="cuda")
            gemm(A, Wn, L.GEMM_NN, L.EPI_DGELU_BF16, M, N, K, out_bf16=o["du"], aux=u, colsum_part=o["cs"])
            o["dm"], o["cs_m"] = torch.empty((M, N), device="cuda", dtype=torch.bfloat16), torch.empty((M // 128, N), device="cuda")
            gemm(A, Wn, L.GEMM_NN, L.EPI_MUL_BF16, M, N, K, out_bf16=o["dm"], aux=u, colsum_part=o["cs_m"])
            res[on] = o
    for k in res[False]:
        if k in ("cs", "cs_m"):       # column sums: same addends, different summation tree (64 row slots instead of 32)
            assert rel_err(res[True][k], res[False][k]) < 1e-5
        else:
            assert torch.equal(res[True][k], res[False][k]), (k, float((res[True][k].float() - res[False][k].float()).abs().max()))


@pytest.mark.experiments
@pytest.mark.parametrize("form,M,N,K", [(L.GEMM_NT, 1664, 768, 3072), (L.GEMM_NN, 1664, 768, 2304), (L.GEMM_NN, 1920, 768, 3072)])
def test_small_m_kernel_split_k_exact_integers(form, M, N, K):
    """with a workspace the small-M kernel splits K >= 1536 into up to four slabs (one round of <= 256 workgroups) + slab epilogue: exact on
    small-integer data, and the repeat launch is bit-identical"""
    A = _ints((M, K), 7)
    B = _ints((N, K), 8) if form == L.GEMM_NT else _ints((K, N), 8)
    ref = A.double() @ (B.double().t() if form == L.GEMM_NT else B.double())
    ws = torch.full((16 << 20,), float("nan"), device="cuda")
    out = torch.empty((M, N), device="cuda")
    with _small_m(True):
        gemm(A, B, form, L.EPI_ADD_F32, M, N, K, out_f32=out, splitk_ws=ws)
        assert torch.equal(out.double(), ref), float((out.double() - ref).abs().max())
        out2 = torch.empty_like(out)
        gemm(A, B, form, L.EPI_ADD_F32, M, N, K, out_f32=out2, splitk_ws=ws)
        assert torch.equal(out, out2)
