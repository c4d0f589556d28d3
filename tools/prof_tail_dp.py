#!/usr/bin/env python3
"""Time carel_tail_losses in data-parallel mode (global-batch MMD over world x 64 samples) for world = 1, 2, 4, 8."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from carel_vae_amd import ops  # noqa: E402
import test_gpu_tail as T  # noqa: E402

B, S, V = 64, 128, 23771
cfg, opt, P, x_last, batch, eps_e, eps_c = T.setup(B, S, V, 3)
dev = "cuda"
W = {k: v.to(dev) for k, v in P.items()}
G = {k: torch.zeros_like(v) for k, v in W.items()}
labels = dict(emo=batch["emo_labels"].to(dev).view(-1).contiguous(), cau=batch["cau_labels"].to(dev).view(-1).contiguous(),
              pair=batch["labels"].to(dev).view(-1).contiguous(), bow=batch["bow_reps"].to(dev).contiguous())
xl = x_last.to(dev)
for world in (1, 2, 4, 8):
    buf = ops.TailBuffers(B, S, 24, opt.e_num_class, V, dev)
    kw = {}
    if world > 1:
        zg = torch.randn(world * B, 48, device=dev) * 0.5
        kw = dict(global_label_sum=torch.tensor([world * 30.0], device=dev), global_n=world * B, global_row_offset=B, z_global=zg, mmd_grad_scale=float(world))
    a = ops.tail_args(buf, xl, W, labels, eps_e.to(dev), eps_c.to(dev), opt, ops.kl_anneal_weight(3, opt), grads=G, drop=(0.5, 7, 0), **kw)
    ops.tail_latents(a)
    for _ in range(3): ops.tail_losses(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.tail_losses(a)
    e1.record(); torch.cuda.synchronize()
    print("world %d (MMD over %4d samples per side): tail_losses %.1f us" % (world, world * B, e0.elapsed_time(e1) / 20 * 1e3))
