import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle runs on torch CPU: a one-GPU box's share is 16 cores whatever the host shows, and torch's default of one thread
    # per visible core oversubscribes them (measured: the oracle ~5x slower)
    try:
        import torch

        n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        torch.set_num_threads(max(1, min(16, n)))
    except Exception:
        pass


def has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_required():
    if not has_gpu():
        pytest.fail("GPU test selected but no HIP device is visible (the engine has no CPU fallback)")
    return True
