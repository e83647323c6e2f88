import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        return cache[name]
    return load


@pytest.fixture(scope="session", autouse=True)
def _bounded_cpu_threads():
    """The CPU oracle runs on torch-CPU; on a shared many-core host the default (one thread per logical core, 256 on
    the GPU boxes) oversubscribes the machine and can stall for minutes.  Cap it at the per-GPU CPU share."""
    import torch
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    torch.set_num_threads(max(1, min(n, 16)))
    yield
