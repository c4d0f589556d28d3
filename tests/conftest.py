import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "experiments: runs on libcarel_hip_exp.so (the build with carel_gemm_set_variant and the kernels that were "
                                       "not adopted); every other test runs on the product library, which has no tuning hooks")


@pytest.fixture(autouse=True)
def _experiments_library(request):
    """Tests marked `experiments` see the EXPERIMENTS build as the active library for their duration (carel_vae_amd._lib.experiments)."""
    if request.node.get_closest_marker("experiments") is None:
        yield
        return
    from carel_vae_amd import _lib
    with _lib.experiments():
        yield


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
