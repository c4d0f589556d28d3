"""Grouped weight gradients (carel_gemm_wgrad_group, round 4): the four dW = dY^T X of an encoder layer -- the dW half of
`loss.backward()` (drl_classifier_ec_mmd_final_mul.py:841) for the q/k/v, attention-output, intermediate and output linears -- as ONE
launch of the 256 x 96 ping-pong kernel (whole tiles written in place, the remainder of the tile count split along K through
compact partial tiles) + one reduction that also sums the bias-gradient partials and the LayerNorm-backward partials.
Exact small-integer data: any wrong tile map, K range, partial-tile address or missed slice shows as an integer difference."""
import ctypes as C

import pytest
import torch

from carel_vae_amd import _lib as L

pytestmark = pytest.mark.gpu

ENC = [(768, 3072, False), (3072, 768, True), (2304, 768, True), (768, 768, False)]      # (M, N, bias gradient) in the encoder's problem order


def _ints(shape, seed, lo=-2, hi=3):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).float().cuda().bfloat16()


def _group(probs, T, ln=(), ws=None):
    """probs: [(dY, X, dW, db or None)]; ln: [(partials, rows, dgamma, dbeta, dbias)]"""
    lib = L.load()
    a = L.WgradGroupArgs()
    for i, (dY, X, dW, db) in enumerate(probs):
        a.prob[i].dY, a.prob[i].X, a.prob[i].dW = dY.data_ptr(), X.data_ptr(), dW.data_ptr()
        a.prob[i].db = None if db is None else db.data_ptr()
        a.prob[i].M, a.prob[i].N = dY.shape[1], X.shape[1]
    a.n_prob, a.T = len(probs), T
    need = lib.carel_gemm_wgrad_group_ws_bytes(C.byref(a))
    assert need >= 0, "shape refused"
    if ws is None:
        ws = torch.full((max(need, 16) // 4 + 64,), float("nan"), device="cuda")        # NaN-filled: a partial tile that is read but never written poisons the result
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    for i, (part, rows, dg, dbt, dbi) in enumerate(ln):
        a.ln[i].partials, a.ln[i].rows = part.data_ptr(), rows
        a.ln[i].dgamma, a.ln[i].dbeta, a.ln[i].dbias = dg.data_ptr(), dbt.data_ptr(), dbi.data_ptr()
    a.n_ln = len(ln)
    L.check(lib.carel_gemm_wgrad_group(C.byref(a), L.current_stream()), "carel_gemm_wgrad_group")
    return need, ws


def _make(shapes, T, seed=0):
    probs, refs = [], []
    for i, (M, N, bias) in enumerate(shapes):
        dY, X = _ints((T, M), seed + 2 * i + 1), _ints((T, N), seed + 2 * i + 2)
        dW = torch.full((M + 1, N), 7.0, device="cuda")               # one guard row behind every dW
        db = torch.full((M + 8,), 7.0, device="cuda") if bias else None
        probs.append((dY, X, dW, db))
        refs.append((dY.double().t() @ X.double(), dY.double().sum(0)))
    return probs, refs


def _check(probs, refs):
    for (dY, X, dW, db), (rW, rb) in zip(probs, refs):
        M = dY.shape[1]
        assert torch.equal(dW[:M].double(), rW), float((dW[:M].double() - rW).abs().max())
        assert torch.equal(dW[M:], torch.full_like(dW[M:], 7.0))
        if db is not None:
            assert torch.equal(db[:M].double(), rb), float((db[:M].double() - rb).abs().max())
            assert torch.equal(db[M:], torch.full_like(db[M:], 7.0))


@pytest.mark.parametrize("T", [8192, 1792, 1664, 2048, 4096, 512, 256])
def test_encoder_layer_group_exact_integers(T):
    """The encoder's problem list (FFN2, FFN1 + bias, QKV + bias, attention output) at the dense token count, at packed ECPE
    token counts (28 / 26 / 32 K tiles: 7 / 6 / 8 slices per split tile) and at the smallest ones the kernel takes."""
    probs, refs = _make(ENC, T)
    need, _ = _group(probs, T)
    _check(probs, refs)
    if T == 8192:
        assert need == 256 * (256 * 96 + 256) * 4 + 256              # 32 split tiles x 8 slices: 25 MB instead of 165 MB of slabs


@pytest.mark.parametrize("shapes,T", [([(768, 768, True)], 8192), ([(768, 768, False)], 1024), ([(256, 96, True)], 256), ([(768, 3072, True), (3072, 768, True)], 2048),
                                      ([(4096, 3072, True)], 512), ([(768, 768, True), (768, 768, False), (768, 768, True)], 1536), ([(2304, 768, True), (256, 192, False)], 4096)])
def test_other_problem_lists_exact_integers(shapes, T):
    """Fewer tiles than CUs (everything split), exactly whole rounds, a partial last round of whole tiles, the largest tile grid."""
    probs, refs = _make(shapes, T, seed=100)
    _group(probs, T)
    _check(probs, refs)


def test_layernorm_partials_ride_on_the_reduction_and_repeat_launches_are_bitwise_identical():
    T = 8192
    lib = L.load()
    g = torch.Generator().manual_seed(5)
    probs = []
    for i, (M, N, bias) in enumerate(ENC):
        dY = (torch.randn((T, M), generator=g) * 0.3).cuda().bfloat16()
        X = torch.randn((T, N), generator=g).cuda().bfloat16()
        probs.append((dY, X, torch.empty((M, N), device="cuda"), torch.empty((M,), device="cuda") if bias else None))
    nb = lib.carel_layernorm_bwd_blocks(T)
    parts = [torch.randn((nb, 3 * 768), generator=g).cuda() for _ in range(2)]
    outs = [[torch.full((768,), float("nan"), device="cuda") for _ in range(3)] for _ in range(2)]
    ln = [(parts[i], T, *outs[i]) for i in range(2)]
    _, ws = _group(probs, T, ln)
    first = [(p[2].clone(), None if p[3] is None else p[3].clone()) for p in probs]
    for i in range(2):
        want = parts[i].double().sum(0).view(3, 768)
        for k in range(3):
            assert float((outs[i][k].double() - want[k]).abs().max()) < 1e-4 * float(want[k].abs().max())
    for (dY, X, dW, db) in probs:
        ref = dY.double().t() @ X.double()
        assert float((dW.double() - ref).norm() / ref.norm()) < 2e-6
        if db is not None:
            assert float((db.double() - dY.double().sum(0)).norm() / dY.double().sum(0).norm()) < 2e-6
    side = torch.cuda.Stream()
    noise = torch.empty(64 << 20, device="cuda")
    for rep in range(6):                                                 # repeat launches, every second one beside a memory-bound kernel on another stream
        for p in probs:
            p[2].fill_(float("nan"))
        if rep & 1:
            with torch.cuda.stream(side):
                noise.add_(1.0)
        _group(probs, T, ln, ws=ws)
        for p, (w0, b0) in zip(probs, first):
            assert torch.equal(p[2], w0) and (b0 is None or torch.equal(p[3], b0))
    torch.cuda.synchronize()


def test_bad_problem_lists_are_refused():
    lib = L.load()
    a = L.WgradGroupArgs()
    a.n_prob, a.T = 1, 8192
    a.prob[0].M, a.prob[0].N = 768, 100                                 # N not a multiple of 96
    assert lib.carel_gemm_wgrad_group_ws_bytes(C.byref(a)) == -1
    a.prob[0].N = 768
    a.T = 8192 + 32                                                     # T not a multiple of 64
    assert lib.carel_gemm_wgrad_group_ws_bytes(C.byref(a)) == -1
    a.T = 128                                                           # two K tiles: below the static schedule's minimum
    assert lib.carel_gemm_wgrad_group_ws_bytes(C.byref(a)) == -1
    a.T, a.n_prob = 8192, 5
    assert lib.carel_gemm_wgrad_group_ws_bytes(C.byref(a)) == -1
    a.n_prob = 1
    assert lib.carel_gemm_wgrad_group(C.byref(a), L.current_stream()) != 0        # null operands
    x = torch.zeros((8192, 768), device="cuda", dtype=torch.bfloat16)
    w = torch.zeros((768, 768), device="cuda")
    a.prob[0].dY, a.prob[0].X, a.prob[0].dW = x.data_ptr(), x.data_ptr(), w.data_ptr()
    a.workspace, a.workspace_bytes = None, 0
    assert lib.carel_gemm_wgrad_group(C.byref(a), L.current_stream()) != 0        # 24 tiles are all split: a workspace is required
