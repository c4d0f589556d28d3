"""Where does a ping-pong GEMM variant (HOOK = 90: fine, 91: wide phases) differ from fp64 on exact small-integer data?
Prints the wrong 16x16 blocks per 256 x BN tile and the first wrong element with its per-K-tile partial sums."""
import os
os.environ.setdefault("CAREL_USE_EXPERIMENTS", "1")      # tuning hooks live in libcarel_hip_exp.so only
import sys
import torch
from carel_vae_amd import _lib as L
from tests.gpu_util import gemm

def ints(shape, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randint(-3, 4, shape, generator=g).float().cuda().bfloat16()

HOOK = int(__import__("os").environ.get("HOOK", 91))


def run(M, N, K, form):
    A = ints((M, K), 1)
    B = ints((N, K), 2) if form == "NT" else ints((K, N), 2)
    out = torch.zeros((M, N), device="cuda")
    lib = L.load()
    L.check(lib.carel_gemm_set_variant(3)); L.check(lib.carel_gemm_set_variant(HOOK))
    gemm(A, B, L.GEMM_NT if form == "NT" else L.GEMM_NN, L.EPI_ADD_F32, M, N, K, out_f32=out)
    torch.cuda.synchronize()
    L.check(lib.carel_gemm_set_variant(0)); L.check(lib.carel_gemm_set_variant(91))
    ref = A.double() @ (B.double().t() if form == "NT" else B.double())
    bad = (out.double() != ref)
    print(f"{form} {M}x{N}x{K}: wrong {int(bad.sum())} of {bad.numel()}  max err {float((out.double()-ref).abs().max())}")
    if bad.any():
        blk = bad.view(M // 16, 16, N // 16, 16).any(3).any(1).cpu()
        for r in range(min(M // 16, 32)):
            print("".join("X" if blk[r, c] else "." for c in range(min(N // 16, 96))))
        # is the wrong value a partial K sum?  compare with sums over K tiles
        idx = bad.nonzero()[0].tolist()
        i, j = idx
        a, b = A[i].double(), (B[j].double() if form == "NT" else B[:, j].double())
        parts = [(a[k*64:(k+1)*64] * b[k*64:(k+1)*64]).sum().item() for k in range(K // 64)]
        print("first wrong at", idx, "got", out[i, j].item(), "ref", ref[i, j].item(), "per-K-tile parts", parts)

if __name__ == "__main__":
    for arg in sys.argv[1:]:
        M, N, K, form = arg.split(",")
        run(int(M), int(N), int(K), form)
