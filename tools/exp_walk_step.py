#!/usr/bin/env python3
"""bench.py with a given XCD walk (argv[1] = 20 N-fastest chunks | 24 M-fastest bands); remaining args go to bench.py."""
import os, sys, runpy
os.environ.setdefault("CAREL_USE_EXPERIMENTS", "1")      # tuning hooks live in libcarel_hip_exp.so only (carel_vae_amd/_lib.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carel_vae_amd import _lib as L
walk = int(sys.argv[1])
L.check(L.load().carel_gemm_set_variant(walk))
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
