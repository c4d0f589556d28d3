"""The C-ABI shared library builds for gfx950 without a GPU, loads, and exports every symbol that
include/carel_hip.h declares (no compute is launched here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from carel_vae_amd import build
    return build.build(verbose=False)


def header_functions(name="carel_hip.h"):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:int|int32_t|int64_t|void\*|const char\*)\s+(carel_\w+)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_header_declares_the_expected_surface():
    names = header_functions()
    for must in ("carel_gemm_bf16", "carel_attention_fwd", "carel_attention_bwd", "carel_rbf_mmd_fwd", "carel_rbf_mmd_bwd",
                 "carel_encoder_forward", "carel_encoder_backward_layer", "carel_tail_losses", "carel_adam_step", "carel_last_error"):
        assert must in names


def test_every_declared_symbol_is_exported_and_bound(lib_path):
    lib = ctypes.CDLL(lib_path)
    from carel_vae_amd import _lib
    names = header_functions()
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    assert set(_lib.SIGNATURES) == set(names), set(_lib.SIGNATURES) ^ set(names)
    assert _lib.load().carel_abi_version() == _lib.ABI_VERSION


def test_product_library_has_no_tuning_hooks_and_the_experiments_build_has_them(lib_path):
    """the product library exports no mutable tuning state -- carel_gemm_set_variant and the kernels that were built,
    measured and not adopted (row-band GEMM + LayerNorm, three-group GEMM, pair split-K) exist in libcarel_hip_exp.so only
    (-DCAREL_EXPERIMENTS, include/carel_hip_experiments.h), which exports the whole product surface as well."""
    from carel_vae_amd import _lib, build
    prod = ctypes.CDLL(lib_path)
    exp_names = header_functions("carel_hip_experiments.h")
    assert set(exp_names) == set(_lib.EXP_SIGNATURES) and "carel_gemm_set_variant" in exp_names
    for n in exp_names:
        assert not hasattr(prod, n), "the product library exports the experiments-only symbol " + n
    import subprocess
    syms = subprocess.run(["nm", "-D", "--defined-only", lib_path], capture_output=True, text=True).stdout
    for frag in ("gemm_tri", "gemm_rowln", "set_variant", "pair_enable", "gemm_pp_launch_pair"):
        assert frag not in syms, frag
    exp = ctypes.CDLL(build.lib_path(build.EXP_TAG))
    for n in header_functions() + exp_names:
        assert hasattr(exp, n), "missing export in the experiments library: " + n
    assert os.path.getsize(lib_path) < os.path.getsize(build.lib_path(build.EXP_TAG))


def test_argument_validation_without_a_gpu(lib_path):
    """Error paths return before any HIP call: null / bad arguments give negative codes and a message."""
    from carel_vae_amd import _lib
    lib = _lib.load()
    assert lib.carel_gemm_bf16(None, None) == -1
    assert b"null" in lib.carel_last_error()
    a = _lib.GemmArgs()
    a.M, a.N, a.K, a.splits = 100, 128, 64, 1
    assert lib.carel_gemm_bf16(ctypes.byref(a), None) == -2
    m = _lib.MmdArgs()
    assert lib.carel_rbf_mmd_fwd(ctypes.byref(m), None) == -1
    at = _lib.AttnArgs()
    at.heads, at.head_dim, at.seq_len, at.batch = 12, 64, 100, 1
    assert lib.carel_attention_fwd(ctypes.byref(at), None) == -2
    assert lib.carel_encoder_act_bytes(64, 128, 12, 0) > 2_000_000_000
    assert lib.carel_encoder_act_bytes(64, 128, 12, 1) < lib.carel_encoder_act_bytes(64, 128, 12, 0) // 6
    assert lib.carel_tail_workspace_floats(64, 24, 23771) > 0


def test_layernorm_backward_block_counts_and_the_scratch_that_holds_them(lib_path):
    """carel_layernorm_bwd_blocks is a pure host function: 4 / 8 / 16 rows per block (<= 2048 / <= 4096 / more rows), NOT monotonic in the row
    count; include/carel_hip.h promises that no smaller count needs more than max(512, ceil(rows / 16)) blocks once rows > 2048 -- what the
    encoder's scratch (sized for batch x seq_len rows, used by packed batches of fewer rows) relies on."""
    from carel_vae_amd import _lib
    lib = _lib.load()
    f = lib.carel_layernorm_bwd_blocks
    assert [f(r) for r in (1, 4, 5, 2048, 2049, 4096, 4097, 8192)] == [1, 1, 2, 512, 257, 512, 257, 512]
    worst = 0
    for rows in range(1, 9001, 7):
        worst = max(worst, f(rows))
        bound = (rows + 3) // 4 if rows <= 2048 else max(512, (rows + 15) // 16)
        assert worst <= bound, (rows, worst, bound)
    # the scratch block grows with the bound: a packed batch inside a (64, 128) scratch never overruns its partial buffers
    assert lib.carel_encoder_scratch_bytes(36, 128) >= lib.carel_encoder_scratch_bytes(16, 128)


def test_product_path_refuses_cpu_tensors(lib_path):
    """No CPU fallback: the module raises instead of computing when tensors are not on the GPU."""
    import torch
    from carel_vae_amd import drl_classifier as M
    from carel_vae_amd._lib import CarelError
    model = M.DrlClassifier(M.make_opt(pair_bow_dim=50), M.encoder_config("zh", vocab_size=100, layers=1))
    z = torch.zeros((2, 128), dtype=torch.long)
    with pytest.raises(CarelError):
        model(z, z, z, torch.zeros(2, 1, dtype=torch.long), torch.zeros(2, 1), torch.zeros(2, 1), torch.zeros(2, 50), 0)
    with pytest.raises(CarelError):
        M.MMDStatistic(4, 4)(torch.zeros(4, 24), torch.zeros(4, 24), [0.1])
    # state_dict carries the reference's key names (checkpoint interchange)
    keys = set(model.state_dict().keys())
    for k in ("encoder.embeddings.word_embeddings.weight", "encoder.encoder.layer.0.attention.self.query.weight",
              "encoder.encoder.layer.0.output.LayerNorm.bias", "encoder.pooler.dense.weight", "emotion_mu.weight",
              "cause_log_var.bias", "emotion_classifier.weight", "pair_classifier.bias", "decoder.weight"):
        assert k in keys
    # get_params(): reference order and membership (:292-295) -- the four latent heads are absent
    ids = {id(p) for p in model.get_params()}
    assert id(model.emotion_mu.weight) not in ids and id(model.cause_log_var.bias) not in ids
    assert id(model.decoder.weight) in ids and id(model.encoder.pooler.dense.bias) in ids
    assert model.get_params()[0] is model.encoder.embeddings.word_embeddings.weight
