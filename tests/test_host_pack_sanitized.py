"""AddressSanitizer + UndefinedBehaviorSanitizer over the library's host C path (GPU sanitizers are not available on
this pool): carel_host_pack_batch -- the batch assembly PrefetchLoader calls without the GIL -- is compiled BY ITSELF from
carel_vae_amd/csrc/host_pack.hip with g++ -fsanitize=address,undefined -DCAREL_HOST_ONLY and driven in a child process (the sanitizer
runtime has to be loaded before Python's allocator) on exactly-sized numpy buffers: a dense batch, token packing with lengths of 0 and
beyond the sequence length, and out-of-range sample indices.  Any out-of-bounds access, misaligned store or overflow aborts the child."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "carel_vae_amd", "csrc", "host_pack.hip")

CHILD = r'''
import ctypes as C, sys
import numpy as np
sys.path.insert(0, %(root)r)
from carel_vae_amd._lib import HostPackArgs      # the structure definition only (no GPU library is loaded)
lib = C.CDLL(%(so)r)
lib.carel_host_pack_batch.restype = C.c_int
lib.carel_host_pack_batch.argtypes = [C.POINTER(HostPackArgs)]
lib.carel_last_error.restype = C.c_char_p
rs = np.random.RandomState(0)
n, S, M, B, Bp = 37, 128, 12, 8, 8
ids = rs.randint(1, 21128, size=(n, S)).astype(np.int64)
lens = rs.randint(0, S + 40, size=n).astype(np.int32)          # some 0, some beyond S (clamped by the packer)
att = (np.arange(S)[None, :] < np.minimum(lens, S)[:, None]).astype(np.int64)
tt = np.zeros((n, S), np.int64)
lab, cau = rs.rand(n).astype(np.float32), rs.rand(n).astype(np.float32)
emo = rs.randint(0, 6, size=n).astype(np.int64)
cols, vals = rs.randint(-1, 23771, size=(n, M)).astype(np.int32), rs.rand(n, M).astype(np.float32)
def run(idx, lengths, emo_arr, emo_is_float, batch_padded=Bp):
    idx = np.asarray(idx, np.int64)
    B = len(idx)
    off, cur = {}, 0
    def take(name, words, align8=False):
        nonlocal cur
        if align8 and cur %% 2: cur += 1
        off[name] = cur; cur += words
    take("input_ids", 2 * B * S, True); take("attention_masks", 2 * B * S, True); take("token_type_ids", 2 * B * S, True)
    take("labels", B); take("cau_labels", B); take("emo_labels", B if emo_is_float else 2 * B, not emo_is_float)
    take("trip", 3 * B * M)
    take("cu", batch_padded + 1); take("tok", (B * S + 127) // 128 * 128)
    dst = np.full(cur, -7, np.int32)                           # EXACTLY the bytes the layout needs: one word more is an overflow
    a = HostPackArgs()
    a.input_ids, a.attention_masks, a.token_type_ids = ids.ctypes.data, att.ctypes.data, tt.ctypes.data
    a.labels, a.cau_labels, a.emo_labels = lab.ctypes.data, cau.ctypes.data, emo_arr.ctypes.data
    a.bow_cols, a.bow_vals, a.idx, a.dst = cols.ctypes.data, vals.ctypes.data, idx.ctypes.data, dst.ctypes.data
    a.n_samples, a.batch, a.seq_len, a.bow_entries, a.emo_is_float = n, B, S, M, emo_is_float
    a.off_input_ids, a.off_attention_masks, a.off_token_type_ids = off["input_ids"], off["attention_masks"], off["token_type_ids"]
    a.off_labels, a.off_cau_labels, a.off_emo_labels, a.off_trip = off["labels"], off["cau_labels"], off["emo_labels"], off["trip"]
    a.lengths = lengths.ctypes.data if lengths is not None else None
    a.batch_padded, a.off_cu, a.off_tok = batch_padded, off["cu"], off["tok"]
    rc = lib.carel_host_pack_batch(C.byref(a))
    return rc, a, dst, off
# (a) dense batch, int64 emotion labels
idx = rs.permutation(n)[:B]
rc, a, dst, off = run(idx, None, emo, 0)
assert rc == 0, lib.carel_last_error()
got = dst[off["input_ids"]:off["input_ids"] + 2 * B * S].view(np.int64).reshape(B, S)
assert (got == ids[idx]).all()
assert (dst[off["labels"]:off["labels"] + B].view(np.float32) == lab[idx]).all()
assert (dst[off["emo_labels"]:off["emo_labels"] + 2 * B].view(np.int64) == emo[idx]).all()
assert (dst[off["cu"]:] == -7).all()                          # the packing regions stay untouched without lengths
# (b) token packing: lengths 0 and > S included; float emotion labels
emo_f = emo.astype(np.float32)
rc, a, dst, off = run(idx, lens, emo_f, 1)
assert rc == 0, lib.carel_last_error()
ln = np.clip(lens[idx], 0, S)
cu = dst[off["cu"]:off["cu"] + Bp + 1]
assert (cu[:B] == np.concatenate(([0], np.cumsum(ln)[:-1]))).all() and cu[B] == ln.sum()
assert a.t_eff == ln.sum() and a.t_pad == (ln.sum() + 127) // 128 * 128
tok = dst[off["tok"]:off["tok"] + a.t_pad]
want = np.concatenate([b * S + np.arange(l) for b, l in enumerate(ln)] + [np.full(a.t_pad - a.t_eff, -1)])
assert (tok == want).all()
# (c) every sample at full length: the token map fills its region to the last word
full = np.full(n, S, np.int32)
rc, a, dst, off = run(idx, full, emo, 0)
assert rc == 0 and a.t_eff == B * S and a.t_pad == B * S
# (d) out-of-range indices are refused before anything is read
for bad in ([0, 1, n], [-1, 2, 3]):
    rc, a, dst, off = run(bad, lens, emo, 0, batch_padded=3)
    assert rc != 0 and b"out of range" in lib.carel_last_error()
# (e) bad sizes / null pointers
a = HostPackArgs()
assert lib.carel_host_pack_batch(C.byref(a)) != 0
print("sanitized host pack ok")
'''


def _asan_runtime():
    for cc in ("g++", "gcc"):
        if shutil.which(cc):
            p = subprocess.run([cc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
            if p and os.path.sep in p and os.path.exists(p):
                return os.path.realpath(p)
    return None


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_pack_batch_under_asan_and_ubsan(tmp_path):
    rt = _asan_runtime()
    if rt is None:
        pytest.skip("libasan runtime not found")
    so = str(tmp_path / "libhostpack_san.so")
    cmd = ["g++", "-x", "c++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           "-DCAREL_HOST_ONLY", "-shared", "-fPIC", "-o", so, SRC]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    child = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, so=so)], capture_output=True, text=True, env=env, timeout=300)
    assert child.returncode == 0 and "sanitized host pack ok" in child.stdout, child.stdout[-2000:] + child.stderr[-4000:]
    # negative control: the same call with a staging block ONE WORD too short must be caught (the sanitizer really is watching)
    short = (CHILD % dict(root=ROOT, so=so)).replace("dst = np.full(cur, -7, np.int32)", "dst = np.full(cur - 1, -7, np.int32)")
    bad = subprocess.run([sys.executable, "-c", short], capture_output=True, text=True, env=env, timeout=300)
    assert bad.returncode != 0 and "AddressSanitizer" in bad.stderr, bad.stderr[-2000:]
