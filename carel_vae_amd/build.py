"""Builds the shared libraries (hipcc, gfx950 only) in-tree next to this file.

    python -m carel_vae_amd.build [--force]

  libcarel_hip.so       the PRODUCT library: plain C ABI (include/carel_hip.h), no dependency on torch, no tuning hooks, no
                        mutable process-wide state beyond what the header lists.
  libcarel_hip_exp.so   the EXPERIMENTS library (-DCAREL_EXPERIMENTS): the same sources plus carel_gemm_set_variant and the kernels
                        that were built, measured and not adopted (include/carel_hip_experiments.h).  Loaded only by the tests and
                        tools that flip a hook (carel_vae_amd._lib.experiments()); nothing on the product path touches it.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
# CAREL_BUILD_TAG=<tag>: a one-off build (ablation flags via CAREL_EXTRA_FLAGS) goes to libcarel_hip_<tag>.so with its own object
# directory and never touches the two standing libraries; load it with CAREL_HIP_LIB=<path> (carel_vae_amd/_lib.py).
TAG = os.environ.get("CAREL_BUILD_TAG", "")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
BASE_FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function", "-Wno-pass-failed", "-I../include"]
# kernel arguments preloaded into SGPRs at wave launch (the first 16 dwords of scalar / pointer arguments; by-value structs are not:
# gemm_pp_kernel repeats the fields its prologue needs as leading scalars).  Measured on the ping-pong GEMM alone: -0.4 us per launch.
BASE_FLAGS += ["-mllvm", "-amdgpu-kernarg-preload-count=16"]
EXP_TAG = "exp"


def lib_path(tag=""):
    return os.path.join(HERE, "libcarel_hip%s.so" % ("_" + tag if tag else ""))


LIB = lib_path(TAG)


def _flags(tag):
    extra = os.environ.get("CAREL_EXTRA_FLAGS", "").split() if tag == TAG and TAG else []
    # (the include path is given relative to the package directory -- hipcc runs there -- so that a tree moved or rsynced to another
    # path keeps its flag hash: ADVICE r03)
    return BASE_FLAGS + (["-DCAREL_EXPERIMENTS"] if tag == EXP_TAG or "-DCAREL_EXPERIMENTS" in extra else []) + extra


def _objdir(tag):
    return os.path.join(CSRC, ".obj_" + tag) if tag else CSRC


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hs += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include")) if f.endswith(".h")]
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force, tag):
    obj = os.path.join(_objdir(tag), src[:-4] + ".o")
    srcp = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) >= max(os.path.getmtime(srcp), _headers_mtime())):
        return obj, False
    cmd = [HIPCC] + _flags(tag) + ["-c", srcp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=HERE)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def _flag_hash(tag):
    return hashlib.sha1(" ".join(_flags(tag)).encode()).hexdigest()


def _stale_flags(tag):
    """The object cache is keyed on the compile flags too (an ablation build with CAREL_EXTRA_FLAGS must not survive into a normal
    build: ADVICE r02): a stamp file holds the flags of the objects on disk.  When they differ, the old objects are DELETED and the new
    stamp is written only after every source compiled (a failed forced rebuild must not leave a matching stamp behind)."""
    d = _objdir(tag)
    stamp = os.path.join(d, ".flags_stamp" + ("_" + tag if tag else ""))
    try:
        old = open(stamp).read().strip()
    except OSError:
        old = None
    if old == _flag_hash(tag):
        return False, stamp
    had = False
    for f in os.listdir(d):
        if f.endswith(".o"):
            os.remove(os.path.join(d, f))
            had = True
    if os.path.exists(stamp):
        os.remove(stamp)
    return had or old is not None, stamp


def _build_one(tag, force, verbose):
    d = _objdir(tag)
    os.makedirs(d, exist_ok=True)
    stale, stamp = _stale_flags(tag)
    force = force or stale
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force, tag), srcs))
    open(stamp, "w").write(_flag_hash(tag) + "\n")
    objs = [o for o, _ in res]
    lib = lib_path(tag)
    if any(c for _, c in res) or not os.path.exists(lib) or force:
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", lib)
    elif verbose:
        print("up to date:", lib)
    return lib


def build(force=False, verbose=True, experiments=True):
    """-> path of the product library (or of the tagged one-off build).  experiments: also build libcarel_hip_exp.so."""
    if TAG:
        return _build_one(TAG, force, verbose)
    lib = _build_one("", force, verbose)
    if experiments:
        _build_one(EXP_TAG, force, verbose)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv)
