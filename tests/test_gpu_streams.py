"""Stream logic under load: the same model with and without the side streams (weight-gradient stream, two forward
chains, per-layer Adam on the auxiliary stream), forward/backward passes back to back on dense and packed batches of
several sizes -- every encoder-layer gradient bit-identical in every pass (tools/stress_streams.py)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_side_streams_bit_identical_over_many_passes():
    spec = importlib.util.spec_from_file_location("stress_streams", os.path.join(ROOT, "tools", "stress_streams.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(12, verbose=False, layers=4) == 0
