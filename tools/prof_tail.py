#!/usr/bin/env python3
"""Time the VAE-tail kernels in isolation (B=64, V=23771) and print the phase stamps of the fused loss kernel.
    python tools/prof_tail.py            # on the GPU box
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from carel_vae_amd import _lib as L, ops  # noqa: E402
import test_gpu_tail as T  # noqa: E402


def main():
    B, S, V = 64, 128, 23771
    cfg, opt, P, x_last, batch, eps_e, eps_c = T.setup(B, S, V, 3)
    dev = "cuda"
    W = {k: v.to(dev) for k, v in P.items()}
    G = {k: torch.zeros_like(v) for k, v in W.items()}
    buf = ops.TailBuffers(B, S, 24, opt.e_num_class, V, dev)
    labels = dict(emo=batch["emo_labels"].to(dev).view(-1).contiguous(), cau=batch["cau_labels"].to(dev).view(-1).contiguous(),
                  pair=batch["labels"].to(dev).view(-1).contiguous(), bow=batch["bow_reps"].to(dev).contiguous())
    xl = x_last.to(dev)
    a = ops.tail_args(buf, xl, W, labels, eps_e.to(dev), eps_c.to(dev), opt, ops.kl_anneal_weight(3, opt), grads=G, drop=(0.5, 7, 0))
    prof = torch.zeros(16, dtype=torch.int64, device=dev)
    lib = L.load()
    for name, fn in (("latents", lambda: ops.tail_latents(a)), ("losses", lambda: ops.tail_losses(a)), ("backward", lambda: ops.tail_backward(a, None))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:10s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us")
    lib.carel_tail_profile(C.c_void_p(prof.data_ptr()))
    ops.tail_losses(a)
    torch.cuda.synchronize()
    lib.carel_tail_profile(None)
    p = prof.cpu().tolist()
    names = ["z", "stage+mmd_fwd", "mmd_bwd", "heads", "kl", "dz_heads", "param_grads"]
    for i, n in enumerate(names):
        print(f"  tail_core {n:14s} {(p[i + 1] - p[i]) * 10 / 1e3:7.2f} us")


if __name__ == "__main__":
    main()
