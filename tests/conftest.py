import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The in-tree .so files are git-ignored build products: build them if this checkout has none yet
    # (hipcc cross-compiles gfx950 without a GPU).  The package itself never builds or falls back.
    if not (os.path.exists(os.path.join(ROOT, "mr_rl_amd", "libmrsim.so")) and
            os.path.exists(os.path.join(ROOT, "oracle", "libmrsim_oracle.so")) and
            os.path.exists(os.path.join(ROOT, "examples", "abi_demo"))):
        import __graft_entry__
        __graft_entry__.build()


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
