import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ppea-depth_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# A GPU box exposes all host cores (128+) but grants a 16-core share: torch's default thread count then oversubscribes
# every CPU-side phase of the suite (model construction, synthetic weights, the oracle).
torch.set_num_threads(min(torch.get_num_threads(), 16))
os.environ.setdefault("OMP_NUM_THREADS", "16")        # inherited by spawned worker processes


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s on CPU")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        # numeric arrays as tensors; string arrays (state_dict key lists) stay numpy
        return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiub" else z[k]) for k in z.files}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.fixture(scope="session")
def device():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")
