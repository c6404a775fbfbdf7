import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _bf16_tests_reach_the_bf16_kernels(request):
    """Small products run on the exact-fp32 small-M kernel in either arithmetic mode (csrc/gemm_skinny.hip), and test
    batches are small: tests of the bf16 mode (name contains 'bf16') switch that routing off so that they exercise the
    bf16 kernels, and restore the default afterwards."""
    if "bf16" not in request.node.name or "gpu" not in request.node.keywords:
        yield
        return
    from _dvae_pkg import dvae_amd
    lib = dvae_amd._lib.load()
    lib.dvae_debug_set_gemm_skinny_mmac(0)
    yield
    lib.dvae_debug_set_gemm_skinny_mmac(-1)
