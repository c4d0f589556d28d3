"""Helpers for the -m gpu tests: all calls go through the C ABI (carel_vae_amd._lib)."""
import ctypes as C

import numpy as np
import torch

from carel_vae_amd import _lib as L


def dev():
    return torch.device("cuda:0")


def to_bf16_bits(t: torch.Tensor) -> torch.Tensor:
    """fp32 tensor -> bf16 tensor (round to nearest even), on the same device."""
    return t.to(torch.bfloat16)


def gemm(A, B, form, epi, M, N, K, splits=1, out_bf16=None, out2_bf16=None, out_f32=None, bias=None,
         resid=None, aux=None, drop=(0, 0, 0, 0.0), lda=None, ldb=None, ldc=None, colsum_part=None, colsum_a=None, splitk_ws=None,
         ws_zeroed=False, resid_ln=None):
    a = L.GemmArgs()
    a.A, a.B = A.data_ptr(), B.data_ptr()
    a.lda = lda if lda is not None else A.stride(0)
    a.ldb = ldb if ldb is not None else B.stride(0)
    a.ldc = ldc if ldc is not None else N
    a.M, a.N, a.K = M, N, K
    a.form, a.epilogue, a.splits = form, epi, splits
    for name, t in (("out_bf16", out_bf16), ("out2_bf16", out2_bf16), ("out_f32", out_f32), ("bias", bias),
                    ("resid_f32", resid), ("aux_bf16", aux)):
        setattr(a, name, None if t is None else t.data_ptr())
    a.drop_seed, a.drop_site, a.drop_idx_offset, a.drop_p = drop
    a.colsum_part = None if colsum_part is None else colsum_part.data_ptr()
    a.colsum_a = None if colsum_a is None else colsum_a.data_ptr()
    a.splitk_ws = None if splitk_ws is None else splitk_ws.data_ptr()
    a.splitk_ws_bytes = 0 if splitk_ws is None else splitk_ws.numel() * splitk_ws.element_size()
    a.splitk_ws_zeroed = 1 if ws_zeroed else 0
    if resid_ln is not None:           # (stats [M, 2], gamma [N], beta [N]): resid holds the pre-LayerNorm rows
        a.resid_ln_stats, a.resid_ln_gamma, a.resid_ln_beta = (t.data_ptr() for t in resid_ln)
    L.check(L.load().carel_gemm_bf16(C.byref(a), L.current_stream()), "carel_gemm_bf16")


def rel_err(got: torch.Tensor, ref: torch.Tensor) -> float:
    got, ref = got.double().cpu(), ref.double().cpu()
    return float((got - ref).norm() / max(ref.norm().item(), 1e-30))
