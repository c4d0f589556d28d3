#!/usr/bin/env python3
"""(round 4) The four weight gradients of an encoder layer: one split-K GEMM + slab reduction each (carel_gemm_bf16 TN +
carel_slab_reduce_f32, the round-3 path) against ONE grouped launch + one reduction (carel_gemm_wgrad_group).  Operands rotate over
`sets` buffer sets (cold operands, as in a training step: 12 layers) or stay on one (hot)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
from tests.gpu_util import gemm
lib = L.load()
L.check(lib.carel_init(0))
ENC = [("ffn2", 768, 3072, False), ("ffn1", 3072, 768, True), ("qkv", 2304, 768, True), ("out", 768, 768, False)]


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for T in (8192, 1792):
    for sets in (1, 12):
        g = torch.Generator().manual_seed(0)
        bufs = []
        for _ in range(sets):
            ps = []
            for name, M, N, bias in ENC:
                dY = (torch.randn((T, M), generator=g) * 0.3).cuda().bfloat16()
                X = torch.randn((T, N), generator=g).cuda().bfloat16()
                ps.append((dY, X, torch.empty((M, N), device="cuda"), torch.empty((M,), device="cuda") if bias else None))
            bufs.append(ps)
        slabs = torch.empty(16 * (768 * 3072 + 3072), device="cuda")
        it = [0]

        def old():
            ps = bufs[it[0] % sets]; it[0] += 1
            for (dY, X, dW, db), (name, M, N, bias) in zip(ps, ENC):
                s = lib.carel_gemm_wgrad_splits(M, N, T)
                cs = slabs[s * M * N:s * M * N + s * M] if bias else None
                gemm(dY, X, L.GEMM_TN, L.EPI_SLAB_F32, M, N, T, splits=s, out_f32=slabs, colsum_a=cs)
                if s > 1:
                    L.check(lib.carel_slab_reduce_f32(slabs.data_ptr(), dW.data_ptr(), M * N, s, 0, L.current_stream()))

        a = L.WgradGroupArgs()
        a.n_prob, a.T = 4, T
        for i, (name, M, N, bias) in enumerate(ENC):
            a.prob[i].M, a.prob[i].N = M, N
        need = lib.carel_gemm_wgrad_group_ws_bytes(C.byref(a))
        ws = torch.empty(need // 4 + 64, device="cuda")
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 4

        def new():
            ps = bufs[it[0] % sets]; it[0] += 1
            for i, (dY, X, dW, db) in enumerate(ps):
                a.prob[i].dY, a.prob[i].X, a.prob[i].dW = dY.data_ptr(), X.data_ptr(), dW.data_ptr()
                a.prob[i].db = None if db is None else db.data_ptr()
            L.check(lib.carel_gemm_wgrad_group(C.byref(a), L.current_stream()))
        to, tn = timeit(old), timeit(new)
        to2, tn2 = timeit(old), timeit(new)
        fl = sum(2.0 * M * N * T for _, M, N, _ in ENC)
        print("T=%5d %-4s: four GEMMs + reductions %6.1f / %6.1f us | grouped %6.1f / %6.1f us (%4.0f TF), workspace %.1f MB" % (
            T, "hot" if sets == 1 else "cold", to, to2, tn, tn2, fl / min(tn, tn2) / 1e6, need / 1e6), flush=True)
        del bufs

# ---- (experiment) two launches side by side: FFN pair as 256 x 192 whole tiles (96 workgroups) on one stream, QKV + attention output as
# 256 x 96 whole tiles (96 workgroups) on another -- no partial tiles at all, 64 CUs left idle
if os.environ.get("TWO_STREAMS", "1") == "1":
    explib = L.load_experiments()
    with L.experiments():
        L.ensure_init()
        for T in (8192, 1792):
            sets = 12
            g = torch.Generator().manual_seed(0)
            bufs = []
            for _ in range(sets):
                ps = []
                for name, M, N, bias in ENC:
                    dY = (torch.randn((T, M), generator=g) * 0.3).cuda().bfloat16()
                    X = torch.randn((T, N), generator=g).cuda().bfloat16()
                    ps.append((dY, X, torch.empty((M, N), device="cuda"), torch.empty((M,), device="cuda") if bias else None))
                bufs.append(ps)
            s2 = torch.cuda.Stream()
            it = [0]
            aa, ab = L.WgradGroupArgs(), L.WgradGroupArgs()
            aa.n_prob, aa.T, ab.n_prob, ab.T = 2, T, 2, T
            ev0, ev1 = torch.cuda.Event(), torch.cuda.Event()

            def fill(a, ps, idx):
                for i, j in enumerate(idx):
                    dY, X, dW, db = ps[j]
                    a.prob[i].dY, a.prob[i].X, a.prob[i].dW = dY.data_ptr(), X.data_ptr(), dW.data_ptr()
                    a.prob[i].db = None if db is None else db.data_ptr()
                    a.prob[i].M, a.prob[i].N = dY.shape[1], X.shape[1]

            def two(solo=0):
                ps = bufs[it[0] % sets]; it[0] += 1
                fill(aa, ps, (0, 1)); fill(ab, ps, (2, 3))
                main = torch.cuda.current_stream()
                if solo != 2:
                    L.check(explib.carel_gemm_set_variant(251))
                    L.check(explib.carel_gemm_wgrad_group(C.byref(aa), L.current_stream()))
                if solo != 1:
                    ev0.record(main); s2.wait_event(ev0)
                    L.check(explib.carel_gemm_set_variant(252))
                    L.check(explib.carel_gemm_wgrad_group(C.byref(ab), C.c_void_p(s2.cuda_stream)))
                    ev1.record(s2); main.wait_event(ev1)
                L.check(explib.carel_gemm_set_variant(250))
            for solo, name in ((0, "both side by side"), (1, "FFN pair alone (256 x 192 whole tiles)"), (2, "QKV + out alone (256 x 96 whole tiles)")):
                t = min(timeit(lambda: two(solo)), timeit(lambda: two(solo)))
                print("T=%5d cold, two launches: %-42s %6.1f us" % (T, name, t), flush=True)
            del bufs
