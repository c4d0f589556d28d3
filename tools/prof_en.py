#!/usr/bin/env python3
"""Runs the English adversarial leg of bench.py alone (for `rocprofv3 --kernel-trace --stats -- python3 tools/prof_en.py`)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    torch.cuda.set_device(0)
    from carel_vae_amd import _lib as L
    L.check(L.load().carel_init(0), "carel_init")
    print(bench.english_leg(torch.device("cuda", 0), 64, int(sys.argv[1]) if len(sys.argv) > 1 else 10), file=sys.stderr)
