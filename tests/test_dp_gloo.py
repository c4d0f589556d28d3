"""Data-parallel plumbing on CPU: two gloo ranks exercise the bucketed gradient reducer and the
global-batch exchange (z all-gather, label sum, noise broadcast) used by carel_vae_amd.dp.DataParallel."""
import os
import socket
from types import SimpleNamespace

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from carel_vae_amd import _lib as L
        from carel_vae_amd.dp import DataParallel, FlatGradReducer
        # ---- reducer: three buckets of a flat gradient, reduced asynchronously, then averaged
        flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        red = FlatGradReducer(flat, {"tail": (900, 1000), "layer0": (100, 900), "embeddings": (0, 100)})
        for name in ("tail", "layer0", "embeddings"):
            red.reduce(name)
        red.wait()
        expect = torch.arange(1000, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
        ok_reduce = bool(torch.allclose(flat, expect))
        # ---- DataParallel hooks on a stand-in model (flat buffers + offsets only)
        n_layers = 2
        offs = {"encoder.embeddings.word_embeddings.weight": 0}
        o = 64
        for l in range(n_layers):
            offs[f"encoder.encoder.layer.{l}.attention.self.query.weight"] = o
            o += 128
        offs["encoder.pooler.dense.weight"] = o
        total = o + 96
        model = SimpleNamespace(_flat=torch.full((total,), float(rank + 1)), _flat_grad=torch.full((total,), float(rank + 1)),
                                _offs=offs, cfg=SimpleNamespace(layers=n_layers), _dp=None, _shadow_versions=1)
        dp = DataParallel(model, embed_chunks=2)
        ok_bcast = bool((model._flat == 1.0).all()) and model._shadow_versions is None     # replicas start identical
        b = dp.reducer.buckets
        # the embedding range travels in pieces (64-element aligned cuts; here 64 floats -> one cut at 0: the first piece is empty and dropped)
        emb = sorted(v for k, v in b.items() if k.startswith("embeddings"))
        ok_buckets = (emb[0][0] == 0 and emb[-1][1] == 64 and all(x[1] == y[0] for x, y in zip(emb, emb[1:]))
                      and b["layer0"] == (64, 192) and b["layer1"] == (192, 320) and b["tail"] == (320, total))
        e, c = dp.broadcast_noise(torch.full((24,), float(rank)), torch.full((24,), 10.0 + rank))
        ok_noise = bool((e == 0).all() and (c == 10).all())
        # collective-free noise: every rank draws the same vectors from a generator seeded by rank 0's broadcast seed
        torch.manual_seed(1234 + rank)                # the global streams differ per rank; the shared generator must not
        e1, c1 = dp.draw_noise(24, "cpu")
        e2, c2 = dp.draw_noise(24, "cpu")
        both = [torch.empty(96) for _ in range(world)]
        dist.all_gather(both, torch.cat((e1, c1, e2, c2)))
        ok_noise = ok_noise and all(torch.equal(b, both[0]) for b in both) and not torch.equal(e1, e2) and not torch.equal(e1, c1)
        # the three-sample model draws (con_dim, ec_dim, ec_dim) from the same shared generator; its pair-label sum is one float summed
        # over ranks (pos_weight of the global batch)
        n3 = dp.draw_noise_sizes((384, 24, 24), "cpu")
        three = [torch.empty(432) for _ in range(world)]
        dist.all_gather(three, torch.cat(n3))
        ok_noise = ok_noise and [t.numel() for t in n3] == [384, 24, 24] and all(torch.equal(b, three[0]) for b in three)
        ysum = dp.all_reduce_sum(torch.tensor([float(rank + 2)]))
        ok_noise = ok_noise and float(ysum) == float(sum(r + 2 for r in range(world)))
        B = 4
        zpack = torch.zeros(B * 48 + 16)
        zpack[:B * 48] = float(rank)
        call = SimpleNamespace(buf=SimpleNamespace(z=zpack[:B * 48].view(B, 48), zpack=zpack), labels={"pair": torch.ones(B) * (rank == 0)})
        ta = L.TailArgs()
        dp.prepare(call)
        dp.fill_global(ta, call)
        gathered, = call.dp_keep                 # [world, B*48 + 16]: rank r's z, then its label sum
        ok_global = (ta.global_n == world * B and ta.global_row_offset == rank * B and ta.mmd_grad_scale == float(world)
                     and ta.global_rank_stride == B * 48 + 16 and gathered.shape == (world, B * 48 + 16)
                     and ta.global_label_ranks == world and ta.global_label_sum == gathered.data_ptr() + 4 * B * 48
                     and bool((gathered[0, :B * 48] == 0).all()) and bool((gathered[1, :B * 48] == 1).all())
                     and float(gathered[0, B * 48]) == B and float(gathered[1, B * 48]) == 0.0
                     and ta.z_global == gathered.data_ptr())
        g1 = dp._gather
        dp.fill_global(ta, call)
        ok_global = ok_global and dp._gather is g1 and ta.z_global == g1.data_ptr()        # receive buffer allocated once
        dp.tail_done()
        h1 = dp.layer_done(1)
        # the handle is all a consumer waits for (what FusedAdam._layer_ready does on its stream): averaged, no later step
        h1.wait()
        lo1, hi1 = dp.reducer.buckets["layer1"]
        ok_avg = h1 is not None and bool(torch.allclose(model._flat_grad[lo1:hi1], torch.full((hi1 - lo1,), 1.5)))
        h0 = dp.layer_done(0)
        # a fused optimiser's hook sees every embedding piece with its handle (FusedAdam._range_ready); the pieces tile the range
        seen = []
        hook = SimpleNamespace(_range_ready=lambda rng, after=None: (after.wait(), seen.append(rng)))
        dp.backward_done(hook)
        ok_avg = ok_avg and seen == emb and dp.reducer.pending == [] and bool(torch.allclose(model._flat_grad, torch.full((total,), (1.0 + 2.0) / 2)))
        # reduced-precision wire format: bf16 on the wire, fp32 result, within bf16 rounding of the exact average
        flat2 = (torch.arange(1000, dtype=torch.float32) * 0.37 + 1.0) * (rank + 1)
        red2 = FlatGradReducer(flat2, {"all": (0, 1000)}, wire_dtype=torch.bfloat16)
        red2.reduce("all"); red2.wait()
        exact = (torch.arange(1000, dtype=torch.float32) * 0.37 + 1.0) * (sum(range(1, world + 1)) / world)
        ok_avg = ok_avg and flat2.dtype == torch.float32 and bool(((flat2 - exact).abs() <= exact.abs() * 2 ** -7).all())
        ok_rows = dp.row_offset(B) == rank * B
        q.put((rank, ok_reduce, ok_bcast, ok_buckets, ok_noise, ok_global, ok_avg, ok_rows))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_data_parallel_plumbing():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert all(r[1:]), r


class _FakeWork:
    """Stands in for a c10d Work: records which 'stream' (a label the test sets) each wait() was called on."""
    current = "main"

    def __init__(self):
        self.waited_on = []

    def wait(self):
        self.waited_on.append(_FakeWork.current)
        return True


def test_pending_orders_every_stream_that_waits_and_widens_once():
    """with RCCL, work.wait() orders only the stream it is called on.  The layer's fused Adam update waits on the
    auxiliary stream; the main stream's wait at the end of backward must still reach the collective (no once-only flag), while the
    bf16-wire widening copy runs exactly once."""
    from carel_vae_amd.dp import _Pending, FlatGradReducer
    w = _FakeWork()
    view = torch.zeros(8)
    p = _Pending(w, view, None)
    _FakeWork.current = "aux"; p.wait()
    _FakeWork.current = "main"; p.wait()
    assert w.waited_on == ["aux", "main"]
    # wire format: the copy back into the fp32 range happens on the first wait only (a second copy would be harmless but is 2x traffic)
    w2 = _FakeWork()
    wire = torch.arange(8, dtype=torch.bfloat16)
    copies = []
    class V:                       # a view that counts copy_ calls
        is_cuda = False
        def copy_(self, src): copies.append(src); view.copy_(src)
    p2 = _Pending(w2, V(), wire)
    _FakeWork.current = "aux"; p2.wait()
    _FakeWork.current = "main"; p2.wait()
    assert w2.waited_on == ["aux", "main"] and len(copies) == 1 and torch.equal(view, wire.float())
    # FlatGradReducer.wait() waits for every pending handle, including ones a consumer already waited for
    red = FlatGradReducer.__new__(FlatGradReducer)
    a, b = _Pending(_FakeWork(), view, None), _Pending(_FakeWork(), view, None)
    _FakeWork.current = "aux"; a.wait()
    red.pending = [a, b]
    _FakeWork.current = "main"; red.wait()
    assert a.work.waited_on == ["aux", "main"] and b.work.waited_on == ["main"] and red.pending == []
