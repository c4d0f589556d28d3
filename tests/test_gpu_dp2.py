"""Two real data-parallel ranks on the one GPU (gloo for the collectives, both processes on cuda:0): DataParallel over two
shards of a batch must reproduce the single-process step on the whole batch -- loss, averaged gradients and updated
weights -- with dropout ON (the masks hash the global row index) and the global-batch MMD / pos_weight / shared noise
(SURVEY 8(e)).  RCCL itself refuses two ranks on one device; its single-rank path is covered by tests/test_gpu_dp.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


KEYS = ["encoder.encoder.layer.0.attention.self.query.weight", "encoder.encoder.layer.1.output.dense.weight",
        "encoder.encoder.layer.1.output.LayerNorm.weight", "encoder.embeddings.position_embeddings.weight",
        "encoder.pooler.dense.weight", "decoder.weight", "pair_classifier.weight", "emotion_classifier.bias"]


def _build():
    from carel_vae_amd import drl_classifier as M
    from oracle import carel_oracle as O
    cfg, opt = O.EncoderConfig(layers=2, vocab_size=1000), O.Opt(pair_bow_dim=513, dropout=0.3)
    z = np.load(os.path.join(HERE, "golden", "zh_ragged.npz"), allow_pickle=False)
    batch = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    mcfg = M.encoder_config("zh", vocab_size=cfg.vocab_size, layers=cfg.layers)          # encoder dropout 0.1 on
    model = M.DrlClassifier(M.make_opt(**vars(opt)), mcfg)
    model.load_state_dict(O.init_params(cfg, opt, seed=int(z["meta"][5])))
    model.to("cuda").train()
    eps = (torch.from_numpy(z["eps_e_0"]), torch.from_numpy(z["eps_c_0"]))
    return M, model, batch, eps, opt


def _step(M, model, batch, eps, opt, lo, hi, fuse=False):
    b = {k: v[lo:hi].cuda() for k, v in batch.items()}
    optim = M.FusedAdam(model, lr=1e-3, fuse_into_backward=fuse)
    model.set_noise(*eps)
    loss = model(b["input_ids"], b["attention_masks"], b["token_type_ids"], b["emo_labels"], b["cau_labels"], b["labels"], b["bow_reps"], 3)
    optim.zero_grad()
    loss.backward()
    named = dict(model.named_parameters())
    grads = {k: named[k].grad.detach().float().cpu().clone() for k in KEYS}
    optim.step()
    torch.cuda.synchronize()
    weights = {k: named[k].detach().float().cpu().clone() for k in KEYS}
    terms = {k: float(v) for k, v in model.last_terms().items()}
    return float(loss.detach()), terms, grads, weights


def _worker(rank, world, port, q, mode="plain"):
    try:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from carel_vae_amd.dp import DataParallel
        M, model, batch, eps, opt = _build()
        # "fused_bf16_wire" (ADVICE r03): the gradient buckets travel as bf16 (rounded once, summed by the collective, widened back into the
        # fp32 gradient buffer) AND each layer's fused Adam update runs inside backward() on the auxiliary stream behind its own bucket's
        # all-reduce -- dp.backward_done with the real FusedAdam hook, embedding bucket in pieces
        fused = mode == "fused_bf16_wire"
        DataParallel(model, wire_dtype=torch.bfloat16) if fused else DataParallel(model)
        B = batch["input_ids"].shape[0]
        n = B // world
        loss, terms, grads, weights = _step(M, model, batch, eps, opt, rank * n, (rank + 1) * n, fuse=fused)
        # plain numpy through the queue (torch tensors would travel as shared-memory handles that die with this process)
        q.put((rank, (loss, terms, {k: v.numpy() for k, v in grads.items()}, {k: v.numpy() for k, v in weights.items()})))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:           # surface the failure instead of hanging the parent
        import traceback
        q.put((rank, "ERROR: " + traceback.format_exc()))


@pytest.mark.parametrize("world,mode", [(2, "plain"), (4, "plain"), (2, "fused_bf16_wire")])
def test_two_rank_data_parallel_equals_single_process_on_the_whole_batch(world, mode):
    """mode "fused_bf16_wire": DataParallel(wire_dtype=bfloat16) + FusedAdam(fuse_into_backward=True) against the plain single-process
    step -- the bf16 wire perturbs every averaged gradient by up to 2^-9 relative per element (stated tolerance: 6e-3 of the tensor's norm
    instead of 2e-3), the ranks still hold bit-identical gradients and weights, and the weights agree with the unfused single-process
    update to the same lr-level bound."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
    for r in res.values():
        assert not isinstance(r, str), r
    M, model, batch, eps, opt = _build()
    loss, terms, grads, weights = _step(M, model, batch, eps, opt, 0, batch["input_ids"].shape[0])
    # the MMD statistic and the pair term's pos_weight are those of the WHOLE batch on every rank
    for r in range(world):
        assert abs(res[r][1]["mmd"] - terms["mmd"]) <= 1e-5 * max(abs(terms["mmd"]), 1e-3), (r, res[r][1]["mmd"], terms["mmd"])
    # batch-mean terms: the average over the shards is the whole-batch value
    for k in ("emo", "cau", "pair", "kl_e", "kl_c", "rec"):
        avg = sum(res[r][1][k] for r in range(world)) / world
        assert abs(avg - terms[k]) <= 2e-4 * max(abs(terms[k]), 1e-3), (k, avg, terms[k])
    # averaged gradients and updated weights are identical on both ranks and equal to the single-process ones
    for k in KEYS:
        g0, g1, g = torch.from_numpy(res[0][2][k]), torch.from_numpy(res[1][2][k]), grads[k]
        assert torch.equal(g0, g1), k
        for r in range(2, world):
            assert torch.equal(g0, torch.from_numpy(res[r][2][k])), (k, r)
        den = float(g.norm()) + 1e-12
        assert float((g0 - g).norm()) / den < (6e-3 if mode == "fused_bf16_wire" else 2e-3), (k, float((g0 - g).norm()) / den)
        w0, w1 = torch.from_numpy(res[0][3][k]), torch.from_numpy(res[1][3][k])
        assert torch.equal(w0, w1), k
        assert float((w0 - weights[k]).abs().max()) <= 2.2e-3, k      # Adam: |update| <= lr, sign flips at ~0 gradients


# ------------------------------------------------------------------------------------------------------------------
# config 4: the English three-space adversarial model under data parallelism (global pos_weight, shared noise, global
# dropout indices; discriminator gradients travel in the tail bucket)
# ------------------------------------------------------------------------------------------------------------------
EN_KEYS = ["encoder.encoder.layer.0.attention.self.query.weight", "encoder.encoder.layer.1.output.dense.weight",
           "encoder.embeddings.position_embeddings.weight", "encoder.pooler.dense.weight", "decoder.weight", "content_classifier.weight",
           "pair_classifier.weight", "content_disc.weight", "ec_disc.weight", "emotion_disc.weight", "cause_disc.bias"]


def _build_en():
    from carel_vae_amd import drl_classifier as M
    from carel_vae_amd import drl_classifier_en as ME
    from oracle import carel_oracle as O
    from oracle import carel_oracle_en as OE
    cfg = O.EncoderConfig(layers=2, vocab_size=900, max_pos=514, type_vocab=1, ln_eps=1e-5, variant="roberta", pad_id=1)
    opt = OE.OptEn(pair_bow_dim=211, dropout=0.3)
    z = np.load(os.path.join(HERE, "golden", "en_adv_small.npz"), allow_pickle=False)
    batch = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    mcfg = M.encoder_config("en", vocab_size=cfg.vocab_size, layers=cfg.layers)          # encoder dropout 0.1 on
    model = ME.DrlClassifier(ME.make_opt(**{k: v for k, v in vars(opt).items() if k in ME.DEFAULT_OPT}), mcfg)
    model.load_state_dict(OE.init_params(cfg, opt, seed=int(z["meta"][5])))
    model.to("cuda").train()
    eps = (torch.from_numpy(z["eps_con_0"]), torch.from_numpy(z["eps_e_0"]), torch.from_numpy(z["eps_c_0"]))
    return model, batch, eps


def _step_en(model, batch, eps, lo, hi):
    b = {k: v[lo:hi].cuda() for k, v in batch.items()}
    opts = model.make_fused_optimizers(adv_lr=1e-3, vae_lr=1e-3)
    model.set_noise(*eps)
    losses = model(b["input_ids"], b["attention_masks"], b["token_type_ids"], b["emo_labels"], b["cau_labels"], b["labels"], b["bow_reps"], 3)
    cd_e, cd_c, ed, ecd, cad, ced, vae = losses
    opts[0].zero_grad(); (cd_e + cd_c).backward(retain_graph=True)        # noqa: E702   drl_classifier_en.py:919-939
    opts[1].zero_grad(); ed.backward(retain_graph=True)                  # noqa: E702
    opts[3].zero_grad(); ecd.backward(retain_graph=True)                 # noqa: E702
    opts[2].zero_grad(); cad.backward(retain_graph=True)                 # noqa: E702
    opts[4].zero_grad(); ced.backward(retain_graph=True)                 # noqa: E702
    opts[5].zero_grad(); vae.backward()                                  # noqa: E702
    named = dict(model.named_parameters())
    grads = {k: named[k].grad.detach().float().cpu().clone() for k in EN_KEYS}
    for o in opts:
        o.step()
    torch.cuda.synchronize()
    weights = {k: named[k].detach().float().cpu().clone() for k in EN_KEYS}
    terms = {k: float(v) for k, v in model.last_terms().items()}
    return terms, grads, weights


def _worker_en(rank, world, port, q):
    try:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from carel_vae_amd.dp import DataParallel
        model, batch, eps = _build_en()
        DataParallel(model)
        n = batch["input_ids"].shape[0] // world
        terms, grads, weights = _step_en(model, batch, eps, rank * n, (rank + 1) * n)
        q.put((rank, (terms, {k: v.numpy() for k, v in grads.items()}, {k: v.numpy() for k, v in weights.items()})))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, "ERROR: " + traceback.format_exc()))


def test_english_adversarial_two_ranks_equal_single_process():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_en, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
    for r in res.values():
        assert not isinstance(r, str), r
    model, batch, eps = _build_en()
    terms, grads, weights = _step_en(model, batch, eps, 0, batch["input_ids"].shape[0])
    for k, v in terms.items():                     # every term is a batch mean (the pair term with the GLOBAL pos_weight)
        avg = sum(res[r][0][k] for r in range(world)) / world
        assert abs(avg - v) <= 3e-4 * max(abs(v), 1e-3), (k, avg, v)
    for k in EN_KEYS:
        g0, g1, g = torch.from_numpy(res[0][1][k]), torch.from_numpy(res[1][1][k]), grads[k]
        assert torch.equal(g0, g1), k
        den = float(g.norm()) + 1e-12
        assert float((g0 - g).norm()) / den < 2e-3, (k, float((g0 - g).norm()) / den)
        w0, w1 = torch.from_numpy(res[0][2][k]), torch.from_numpy(res[1][2][k])
        assert torch.equal(w0, w1), k
        step = 1e-2 if k.split(".")[0].endswith("disc") else 1e-3          # RMSprop: lr / sqrt(1 - alpha); Adam: lr
        assert float((w0 - weights[k]).abs().max()) <= 2.2 * step, k


def test_bench_launches_two_ranks_by_itself_on_one_gpu():
    """`python bench.py --gpus 2` with no launcher in the environment (it used to run ONE GPU and report n_gpus 1):
    the parent starts the rank processes as a child before touching the GPU and relays rank 0's line; rehearsed on this one-GPU
    box with both ranks on cuda:0 and gloo carrying the collectives (CAREL_REHEARSE_ONE_GPU=1)."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, CAREL_REHEARSE_ONE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-ecpe"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["config"]["parallelism"] == "dp2" and d["value"] > 0
