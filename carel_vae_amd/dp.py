"""Data-parallel training of DrlClassifier: one process per GPU, `torch.distributed` (backend "nccl" is
RCCL on ROCm, over xGMI).  The reference has no distributed code at all; this is
the MI355X-native scaling path asked for by BASELINE.json.

What is exchanged per step
  * gradients: one all-reduce (average) per bucket, issued as soon as the bucket's gradients exist --
    after the tail backward (pooler + heads + decoder), after each encoder layer's backward (7.09 M
    elements each, contiguous in the flat gradient buffer), and after the embedding backward -- so that
    all but the last bucket overlap with the remaining backward kernels (RCCL runs on its own stream);
  * the three batch-coupled quantities of the loss, so that N ranks x B samples
    reproduce a single-process step on the N*B batch exactly: the sampled latents z (all-gather,
    N*B x 48 floats) for the global-batch RBF-MMD, the label sum (one float) for `pos_weight`, and the two
    reparameterisation noise vectors (broadcast from rank 0, 48 floats).
Dropout masks are already shard-consistent: the kernels hash the GLOBAL element index (row offset =
rank * B).
"""
import torch
import torch.distributed as dist


class _Pending:
    """One bucket's collective.  `wait()` is what a consumer of the averaged gradients calls on ITS stream: it makes
    that stream wait for the collective (RCCL: `work.wait()` orders only the stream it is called on; gloo: host-side) and,
    with a reduced-precision wire format, for the widening of the result back into the fp32 gradient range.  It may be
    called from several streams (the auxiliary stream of the layer's fused Adam update, then the main stream at the end of
    backward): EVERY call orders its own stream; only the widening copy itself runs once, on the first stream that waits,
    and later callers wait for the event recorded behind it."""

    def __init__(self, work, view, wire):
        self.work, self.view, self.wire = work, view, wire
        self.widened, self.event = False, None

    def wait(self):
        self.work.wait()                    # always: a stream that has not waited itself is not ordered after the collective
        if self.wire is None:
            return
        if not self.widened:
            self.view.copy_(self.wire)
            self.widened = True
            if self.view.is_cuda:
                self.event = torch.cuda.Event()
                self.event.record()
        elif self.event is not None:
            torch.cuda.current_stream().wait_event(self.event)


class FlatGradReducer:
    """Bucketed, asynchronous gradient averaging over named contiguous ranges of one flat fp32 tensor.
    Device-agnostic (gloo on CPU in the tests, RCCL on the GPUs) and with ONE control flow for both: every bucket is
    averaged by the collective itself -- RCCL's ReduceOp.AVG, or a pre-division by the world size followed by SUM where the
    backend has no AVG (gloo) -- so the handle that `reduce` returns is all a consumer has to wait for; there is no
    host-side post-processing step that only one backend would take.
    wire_dtype = torch.bfloat16 halves the bytes on the links (207 instead of 414 MB per step for BERT-base; xGMI rings are
    per-link bound): the bucket is rounded to bf16 once, summed by the collective in bf16, and widened back into the fp32
    buffer that Adam reads (fp32 master gradients, moments and weights are untouched).  Off by default: it perturbs every
    gradient by up to 2^-9 relative, which the single-process equality tests of this path would have to absorb, and its
    benefit can only be measured on a multi-GPU node."""

    def __init__(self, flat_grad, buckets, group=None, wire_dtype=None):
        self.flat, self.buckets, self.group = flat_grad, dict(buckets), group
        self.world = dist.get_world_size(group)
        backend = dist.get_backend(group)
        self.use_avg = backend == "nccl"
        self.wire_dtype = wire_dtype
        self._wire = {}            # bucket name -> preallocated wire buffer
        self.pending = []

    def reduce(self, name):
        """Start the averaging all-reduce of one bucket; returns its handle (None for an empty bucket)."""
        lo, hi = self.buckets[name]
        if hi <= lo:
            return None
        view = self.flat[lo:hi]
        if not self.use_avg:
            view.mul_(1.0 / self.world)
        wire = None
        if self.wire_dtype is not None:
            wire = self._wire.get(name)
            if wire is None:
                wire = self._wire[name] = torch.empty(hi - lo, dtype=self.wire_dtype, device=view.device)
            wire.copy_(view)
        op = dist.ReduceOp.AVG if self.use_avg else dist.ReduceOp.SUM
        work = dist.all_reduce(view if wire is None else wire, op=op, group=self.group, async_op=True)
        pend = _Pending(work, view, wire)
        self.pending.append(pend)
        return pend

    def wait(self):
        for pend in self.pending:
            pend.wait()
        self.pending = []


class DataParallel:
    """Attach to a DrlClassifier: `dp = DataParallel(model)`; then train as usual (same forward / backward /
    optimiser calls).  Every rank must call forward with the same local batch size."""

    def __init__(self, model, group=None, global_batch_terms=True, wire_dtype=None, embed_chunks=4, overlap_wgrad=False):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised (init_process_group) before DataParallel")
        self.model, self.group = model, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.global_batch_terms = global_batch_terms
        model._dp = self
        # RCCL brings its own stream; with the weight-gradient and Adam streams that makes four -- the number of hardware
        # queues HIP uses by default.  The forward pass's second chain would be a fifth and ends up sharing a queue
        # (measured with one RCCL rank: 10.9 ms per step with it, 10.0 ms without), so it is switched off under DP.
        if hasattr(model, "forward_chains"):
            model.forward_chains = False
        # Round 4: the weight gradients stay on the MAIN stream under DP (overlap_wgrad=True re-enables the side stream).  The stream budget
        # again: main + RCCL's stream + the input copy stream are three; the weight-gradient stream and the Adam stream would make five
        # on four hardware queues.  Measured with one RCCL rank on one box (bench.py, CAREL_FORCE_DP=1; the one-rank "all-reduce" is a
        # 414-MB copy per step): 15.6 ms per step with the weight-gradient side stream, 8.5 ms with the weight gradients on the main
        # stream and Adam in step(), 9.2 ms with Adam per layer on the auxiliary stream; 7.7 ms without DataParallel.  What DP needs to
        # overlap is the gradient all-reduce with the remaining backward pass -- asynchronous on RCCL's stream either way.
        if hasattr(model, "overlap_wgrad"):
            model.overlap_wgrad = bool(overlap_wgrad)
        dist.broadcast(model._flat, src=0, group=group)        # identical replicas
        seed = torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int64).to(model._flat.device)
        dist.broadcast(seed, src=0, group=group)
        self._noise_seed, self._noise_gen = int(seed.item()), None
        model._shadow_versions = None
        buckets = self._buckets(model)
        # The embedding bucket (16.2 M floats for BERT-base, a sixth of the payload) is complete only when backward is: nothing is
        # left to hide its all-reduce behind.  It therefore travels in `embed_chunks` pieces, so that the Adam update of piece k
        # (FusedAdam(fuse_into_backward=True): auxiliary stream) runs while piece k + 1 is on the links.
        lo, hi = buckets.pop("embeddings")
        n = max(1, int(embed_chunks))
        cuts = [lo + ((hi - lo) * i // n) // 64 * 64 for i in range(n)] + [hi]
        self._embed_names = []
        for i in range(n):
            if cuts[i + 1] > cuts[i]:
                buckets[f"embeddings{i}"] = (cuts[i], cuts[i + 1])
                self._embed_names.append(f"embeddings{i}")
        self.reducer = FlatGradReducer(model._flat_grad, buckets, group, wire_dtype=wire_dtype)
        self._gather = None        # the all-gather's receive buffer, allocated once (same size every step)

    @staticmethod
    def _buckets(model):
        offs, n_layers = model._offs, model.cfg.layers
        starts = [offs[f"encoder.encoder.layer.{l}.attention.self.query.weight"] for l in range(n_layers)]
        pooler = offs["encoder.pooler.dense.weight"]
        b = {"embeddings": (0, starts[0]), "tail": (pooler, model._flat.numel())}
        for l in range(n_layers):
            b[f"layer{l}"] = (starts[l], starts[l + 1] if l + 1 < n_layers else pooler)
        return b

    # ---- hooks called by DrlClassifier -------------------------------------------------------
    def row_offset(self, local_batch):
        return self.rank * local_batch

    def broadcast_noise(self, eps_e, eps_c):
        both = torch.cat((eps_e, eps_c))
        dist.broadcast(both, src=0, group=self.group)
        n = eps_e.numel()
        return both[:n].contiguous(), both[n:].contiguous()

    def draw_noise(self, n, device):
        """The two reparameterisation noise vectors, identical on every rank WITHOUT a collective: a per-device generator
        seeded once (at construction) with a seed broadcast from rank 0."""
        if self._noise_gen is None or self._noise_gen.device != torch.device(device):
            self._noise_gen = torch.Generator(device=device)
            self._noise_gen.manual_seed(self._noise_seed)
        return (torch.randn(n, device=device, generator=self._noise_gen), torch.randn(n, device=device, generator=self._noise_gen))

    def draw_noise_sizes(self, sizes, device):
        """Noise vectors of the given sizes from the same shared generator (models with more than two samples)."""
        if self._noise_gen is None or self._noise_gen.device != torch.device(device):
            self._noise_gen = torch.Generator(device=device)
            self._noise_gen.manual_seed(self._noise_seed)
        return tuple(torch.randn(n, device=device, generator=self._noise_gen) for n in sizes)

    def all_reduce_sum(self, t):
        """In-place sum over ranks of a small device tensor (stream-ordered; e.g. the pair-label sum for pos_weight)."""
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def prepare(self, call):
        """Called when the batch is known (before the encoder runs): park this rank's label sum behind the z buffer, so
        that nothing but the all-gather itself sits between the latents and the loss kernel."""
        if self.global_batch_terms:
            n = call.buf.z.numel()
            call.buf.zpack[n:] = call.labels["pair"].sum()

    def fill_global(self, ta, call):
        """ONE all-gather carries the sampled latents and each rank's label sum; the loss kernel reads both straight
        out of the gathered buffer (per-rank stride)."""
        if not self.global_batch_terms:
            return
        z, mine = call.buf.z, call.buf.zpack
        n, stride = z.numel(), call.buf.zpack.numel()
        if self._gather is None or self._gather.numel() != self.world * stride or self._gather.device != z.device:
            self._gather = torch.empty(self.world * stride, device=z.device, dtype=z.dtype)
        flat = self._gather        # read by this step's loss kernels only; the next step's gather is stream-ordered behind them
        dist.all_gather_into_tensor(flat, mine, group=self.group)
        ta.z_global, ta.global_n, ta.global_rank_stride = flat.data_ptr(), self.world * z.shape[0], stride
        ta.global_row_offset = self.rank * z.shape[0]
        ta.global_label_sum, ta.global_label_ranks = flat.data_ptr() + 4 * n, self.world
        ta.mmd_grad_scale = float(self.world)       # gradients are averaged over ranks afterwards
        call.dp_keep = (flat.view(self.world, stride),)

    def tail_done(self):
        self.reducer.reduce("tail")

    def layer_done(self, layer):
        """-> handle of the layer's averaging all-reduce; `handle.wait()` on a stream is all the consumer (the fused Adam
        update of that layer) needs before it reads the gradients -- the same under RCCL and gloo."""
        return self.reducer.reduce(f"layer{layer}")

    def backward_done(self, adam_hook=None):
        """End of backward: the embedding pieces go out back to back; with a fused optimiser each piece's update follows its
        own all-reduce on the auxiliary stream.  Then the MAIN stream waits for every collective of the step (also those an
        auxiliary-stream consumer has already waited for: that wait ordered the auxiliary stream only), so whatever reads
        `.grad` / the flat gradient buffer after backward() -- a stock optimiser, gradient clipping, a norm -- is ordered."""
        for name in self._embed_names:
            work = self.reducer.reduce(name)
            if adam_hook is not None and work is not None:
                adam_hook._range_ready(self.reducer.buckets[name], after=work)
        self.reducer.wait()

    def reduce_scalar_mean(self, t):
        """Average a scalar (e.g. the loss for logging) over ranks."""
        t = t.detach().clone().reshape(1)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t / self.world
