"""The data-parallel path on the GPU with a single RCCL rank: DataParallel hooks, bucketed all-reduce on the RCCL stream,
global-batch exchange and the per-layer Adam updates chained behind each bucket's all-reduce.  With one rank every
collective is the identity, so the trajectory must equal the plain single-process one.  (Two ranks are covered with
gloo on CPU tensors in tests/test_dp_gloo.py; multi-GPU runs belong to the driver.)"""
import os
import socket

import pytest
import torch

from carel_vae_amd import drl_classifier as M
from oracle import carel_oracle as O
from tests.test_gpu_model import CASES, build, call, load, relnorm

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("fuse", [False, True])
def test_single_rank_rccl_equals_plain(golden_dir, fuse):
    import torch.distributed as dist
    from carel_vae_amd.dp import DataParallel
    cfg, opt = CASES["zh_small"]
    opt = O.Opt(**{**vars(opt), "dropout": 0.3})
    z, batch = load(golden_dir, "zh_small")
    B, S, Lr, vocab, V, wseed, bseed, steps, it0 = (int(v) for v in z["meta"])
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        res = {}
        for tag in ("plain", "dp"):
            model, P = build(cfg, opt, wseed, train_dropout=True)
            model.train()
            if tag == "dp":
                DataParallel(model)
            optim = M.FusedAdam(model, lr=1e-5, fuse_into_backward=fuse)
            losses, first = [], None
            for s in range(2):
                model.set_noise(torch.from_numpy(z["eps_e_0"]), torch.from_numpy(z["eps_c_0"]))
                loss = model(*call(model, batch, it0 + s))
                optim.zero_grad()
                loss.backward()
                optim.step()
                losses.append(float(loss.detach()))
                if s == 0:
                    torch.cuda.synchronize()
                    first = {k: p.detach().clone() for k, p in model.named_parameters()}
            torch.cuda.synchronize()
            res[tag] = (losses, first)
        noisy = ("embeddings.word", "embeddings.position", "embeddings.token_type")
        assert res["plain"][0][0] == res["dp"][0][0]
        for k, w in res["plain"][1].items():
            if any(n in k for n in noisy):
                assert relnorm(res["dp"][1][k], w) < 1e-6, k
            else:
                assert torch.equal(w, res["dp"][1][k]), k
        for a, b in zip(res["plain"][0], res["dp"][0]):
            assert abs(a - b) <= 1e-3 * max(abs(a), 1.0)
    finally:
        if created:
            dist.destroy_process_group()


def test_reducer_handle_waited_on_a_side_stream_then_read_on_main():
    """with RCCL, work.wait() orders only the stream it is called on.  A bucket's handle is waited for on an auxiliary
    stream first (what the layer's fused Adam update does), then FlatGradReducer.wait() runs on the main stream and the gradients are read
    there: every stream that waits must be ordered after the collective and after the bf16-wire widening copy (the once-only `done` flag of
    round 2 skipped the main stream's wait).  One RCCL rank, so the average is the identity and the expected values are exact: fp32 buckets
    unchanged, bf16-wire buckets = the bf16 rounding of the gradients.  Many rounds with a busy main stream in front of the collective, so
    that an un-ordered read would see the pre-collective (here: pre-widening) contents."""
    import torch.distributed as dist
    from carel_vae_amd.dp import FlatGradReducer
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n = 8 << 20
        aux = torch.cuda.Stream()
        busy = torch.empty(64 << 20, device="cuda")
        for wire in (None, torch.bfloat16):
            flat = torch.empty(n, device="cuda")
            red = FlatGradReducer(flat, {"a": (0, n // 2), "b": (n // 2, n)}, wire_dtype=wire)
            for rnd in range(6):
                src = torch.randn(n, device="cuda") * (1.0 + rnd)
                flat.copy_(src)
                busy.add_(1.0)                                   # the collective queues behind work on the main stream
                ha, hb = red.reduce("a"), red.reduce("b")
                if wire is not None:
                    flat.fill_(float("nan"))                     # what a reader that is NOT ordered after the widening copy would see
                aux.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(aux):
                    ha.wait()                                    # the consumer on the auxiliary stream ...
                    seen_aux = flat[: n // 2].clone()
                red.wait()                                       # ... and then the main stream, for everything
                seen_main = flat.clone()
                torch.cuda.synchronize()
                want = src if wire is None else src.to(wire).float()
                assert torch.equal(seen_main, want), (wire, rnd)
                assert torch.equal(seen_aux, want[: n // 2]), (wire, rnd)
                assert red.pending == []
    finally:
        if created:
            dist.destroy_process_group()
