#!/usr/bin/env python3
"""(round 4) Where do the extra milliseconds of SURVEY 8(d)'s step -- H2D of a ready batch + the loss read-back -- go?
Variants of the dense bench step, same box, interleaved: resident inputs | seven .to() copies on the compute stream | one packed copy on a
copy stream one step ahead | + read-back every 10 steps by .item() | + read-back through a pinned buffer and an event (no queue drain).
Also the host's enqueue time per step (the time the Python loop needs when the GPU is not the limit)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carel_vae_amd import _lib as L
from carel_vae_amd import drl_classifier as M
from carel_vae_amd.data import synthetic_ecpe_batch

dev = torch.device("cuda", 0)
lib = L.load()
L.check(lib.carel_init(0), "carel_init")
opt, cfg = M.make_opt(), M.encoder_config("zh")
model = M.DrlClassifier(opt, cfg, seed=0).to(dev)
model.train()
optim = M.FusedAdam(model, lr=opt.vae_lr)
B = 64
host = [synthetic_ecpe_batch(B, 128, cfg.vocab_size, opt.pair_bow_dim, seed=1 + i, shape="A") for i in range(4)]
lengths = [b["attention_masks"].sum(1).tolist() for b in host]
pinned = [{k: v.pin_memory() for k, v in b.items()} for b in host]
resident = [{k: v.to(dev) for k, v in b.items()} for b in host]
KEYS = ("input_ids", "attention_masks", "token_type_ids", "emo_labels", "cau_labels", "labels", "bow_reps")


def run(b, i):
    loss = model(*(b[k] for k in KEYS), i % 41, seq_lengths=lengths[i % 4])
    optim.zero_grad()
    loss.backward()
    optim.step()
    return loss


def leg(name, n, get, readback=None):
    for i in range(3):
        run(get(i), i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = 0.0
    for i in range(n):
        h0 = time.perf_counter()
        loss = run(get(i), i)
        if readback is not None:
            readback(i, loss)
        th += time.perf_counter() - h0
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-58s %7.3f ms/step   (host loop %.3f ms/step)" % (name, 1e3 * dt / n, 1e3 * t_host / n), flush=True)
    return dt / n


# packed single-copy path: one page-locked block per batch, one H2D on a copy stream, issued one step ahead
def make_packed():
    blocks = []
    for b in host:
        parts, lay, o = [], {}, 0
        for k in KEYS:
            t = b[k].contiguous().view(-1).view(torch.uint8)
            lay[k] = (o, t.numel(), b[k].dtype, tuple(b[k].shape))
            pad = (-t.numel()) % 256
            parts += [t, torch.zeros(pad, dtype=torch.uint8)]
            o += t.numel() + pad
        blocks.append((torch.cat(parts).pin_memory(), lay))
    return blocks


blocks = make_packed()
nbytes = blocks[0][0].numel()
slots = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(3)]
copy_stream = torch.cuda.Stream(device=dev)
copied = [torch.cuda.Event() for _ in range(3)]
released = [None, None, None]
state = {"next": 0}


def issue(i):
    s = i % 3
    if released[s] is not None:
        released[s].synchronize()            # (host wait; never blocks in practice: the slot was released two steps ago)
    with torch.cuda.stream(copy_stream):
        slots[s].copy_(blocks[i % 4][0], non_blocking=True)
        copied[s].record(copy_stream)


def get_packed(i):
    if state["next"] <= i:
        issue(i)
        state["next"] = i + 1
    if state["next"] <= i + 1:
        issue(i + 1)                          # next batch's copy flies under this step
        state["next"] = i + 2
    s = i % 3
    torch.cuda.current_stream().wait_event(copied[s])
    lay = blocks[i % 4][1]
    out = {k: slots[s][o:o + n].view(dt).view(shp) for k, (o, n, dt, shp) in lay.items()}
    ev = torch.cuda.Event()
    released[s] = ev
    return out


def mark_release(i, loss):
    released[i % 3].record()


def rb_item(i, loss):
    if i % 10 == 9:
        float(loss.detach())


pin_loss = torch.zeros(64, dtype=torch.float32).pin_memory()
rb_events = []


def rb_pinned(i, loss, acc={"run": None}):
    acc["run"] = loss.detach().clone() if acc["run"] is None else acc["run"] + loss.detach()
    if i % 10 == 9:
        slot = (i // 10) % 64
        pin_loss[slot:slot + 1].copy_(acc["run"].reshape(1), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        rb_events.append((ev, slot))
        acc["run"] = None
    while rb_events and rb_events[0][0].query():       # print the ones that have arrived (no wait)
        rb_events.pop(0)


def both(f, g):
    def h(i, loss):
        f(i, loss)
        g(i, loss)
    return h


N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for rnd in range(2):
    print("--- round %d" % rnd)
    leg("resident inputs", N, lambda i: resident[i % 4])
    leg("7 x .to() on the compute stream", N, lambda i: {k: v.to(dev, non_blocking=True) for k, v in pinned[i % 4].items()})
    state["next"] = 0
    leg("one packed copy, copy stream, one step ahead", N, get_packed, mark_release)
    leg("resident + .item() every 10 steps", N, lambda i: resident[i % 4], rb_item)
    leg("resident + pinned read-back every 10 steps (no drain)", N, lambda i: resident[i % 4], rb_pinned)
    leg("7 x .to() + .item() every 10  (round-3 leg)", N, lambda i: {k: v.to(dev, non_blocking=True) for k, v in pinned[i % 4].items()}, rb_item)
    state["next"] = 0
    leg("packed copy + pinned read-back  (candidate)", N, get_packed, both(mark_release, rb_pinned))

# host enqueue time alone: how long does the Python loop take per step when nothing waits for the GPU?
torch.cuda.synchronize()
for n in (1, 2, 4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        run(resident[i % 4], i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host enqueue of %d step(s): %.3f ms per step; GPU done %.3f ms after" % (n, 1e3 * (t1 - t0) / n, 1e3 * (t2 - t1)))
print("cpus:", len(os.sched_getaffinity(0)), "torch threads:", torch.get_num_threads())
