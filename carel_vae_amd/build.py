"""Builds libcarel_hip.so (hipcc, gfx950 only) in-tree next to this file.

    python -m carel_vae_amd.build [--force]

The shared library has a plain C ABI (include/carel_hip.h) and no dependency on torch.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
# CAREL_BUILD_TAG=<tag>: an experiment build (ablation flags via CAREL_EXTRA_FLAGS) goes to libcarel_hip_<tag>.so with its own object
# directory and never touches the product library; load it with CAREL_HIP_LIB=<path> (carel_vae_amd/_lib.py).
TAG = os.environ.get("CAREL_BUILD_TAG", "")
LIB = os.path.join(HERE, "libcarel_hip%s.so" % ("_" + TAG if TAG else ""))
OBJDIR = os.path.join(CSRC, ".obj_" + TAG) if TAG else CSRC
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(ROOT, "include")] + os.environ.get("CAREL_EXTRA_FLAGS", "").split()
# kernel arguments preloaded into SGPRs at wave launch (the first 16 dwords of scalar / pointer arguments; by-value structs are not:
# gemm_pp_kernel repeats the fields its prologue needs as leading scalars).  Measured on the ping-pong GEMM alone: -0.4 us per launch.
FLAGS += ["-mllvm", "-amdgpu-kernarg-preload-count=16"]
FILE_FLAGS = {}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hs.append(os.path.join(ROOT, "include", "carel_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force):
    obj = os.path.join(OBJDIR, src[:-4] + ".o")
    srcp = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) >= max(os.path.getmtime(srcp), _headers_mtime())):
        return obj, False
    cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(src, []) + ["-c", srcp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def _flags_changed():
    """The object cache is keyed on the compile flags too (an ablation build with CAREL_EXTRA_FLAGS must not survive into a normal
    build: ADVICE r02): a stamp file holds the flags of the objects on disk."""
    import hashlib
    stamp = os.path.join(OBJDIR, ".flags_stamp")
    h = hashlib.sha1(" ".join(FLAGS + [k + ":" + " ".join(v) for k, v in sorted(FILE_FLAGS.items())]).encode()).hexdigest()
    try:
        old = open(stamp).read().strip()
    except OSError:
        old = None
    if old != h:
        open(stamp, "w").write(h + "\n")
        return old is not None or any(f.endswith(".o") for f in os.listdir(OBJDIR))
    return False


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    force = _flags_changed() or force
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or not os.path.exists(LIB) or force:
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
