import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "needs_reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir("/root/reference/src/models/components")
    skip_ref = pytest.mark.skip(reason="/root/reference not present (GPU box)")
    try:
        import torch
        have_gpu = torch.cuda.is_available() and os.path.exists(os.path.join(ROOT, "bio-diffusion_amd", "libgcdm_hip.so"))
    except Exception:
        have_gpu = False
    skip_gpu = pytest.mark.skip(reason="needs an MI355X and the built libgcdm_hip.so (run on the GPU box: pytest -m gpu)")
    for item in items:
        if "needs_reference" in item.keywords and not have_ref:
            item.add_marker(skip_ref)
        if "gpu" in item.keywords and not have_gpu:
            item.add_marker(skip_gpu)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
