import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "rwm-pt-pytorch_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    # the oracle (test infrastructure) is plain C: build it on demand
    from oracle import oracle

    oracle.lib()
    # the engine library normally travels pre-built; on a fresh checkout build it (hipcc cross-compiles, ~1 min)
    import ptrwm_hip

    if not os.path.exists(ptrwm_hip.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no ROCm GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def device():
    import torch

    return torch.device("cuda:0")
