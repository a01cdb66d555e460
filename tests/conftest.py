import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
_TOOLS = os.path.join(ROOT, "tools")          # tools/synth_convnets.py: synthetic classifier containers (not product code)
if _TOOLS not in sys.path:
    sys.path.insert(1, _TOOLS)


# NativeConvNet's three named routes onto PyTorch operators (train() mode, CPU module, un-lowerable trace) are errors in the suite;
# the one test that covers those routes turns this off for itself
os.environ.setdefault("AUDIOPURE_STRICT", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # and their announcement is an error too: a lowering that degrades to MIOpen can never pass with a warning nobody reads
    config.addinivalue_line("filterwarnings", "error:.*PyTorch operators.*:RuntimeWarning")


@pytest.fixture(autouse=True)
def _hand_big_blocks_back(request):
    """GPU tests at BASELINE's full sizes hold workspaces of 150-165 GB each (B = 512 in the bf16 modes).  After such a test, collect
    what reference cycles kept alive and hand cached blocks back to the driver, so that the next full-size test does not depend on
    which test ran before it (a small tensor carved out of a cached 155 GB block pins the whole block)."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    import gc
    import torch
    if torch.cuda.is_available():
        gc.collect()
        if torch.cuda.memory_reserved() - torch.cuda.memory_allocated() > (16 << 30):
            torch.cuda.empty_cache()


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "golden_v1.npz")
    return np.load(path)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
