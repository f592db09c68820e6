import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from boxsegliver_amd.utils import hostcpu  # noqa: E402  (no torch inside)

# Before any test module imports torch: size the CPU thread pools (this process and the children the tests start) by the CPUs
# the cgroup grants, not by the host's count -- a one-GPU box shows 256 CPUs and grants 16, and the oracle (most of the GPU
# suite's wall time) ran 5.8x slower on 128 throttled threads (boxsegliver_amd/utils/hostcpu.py).
hostcpu.size_thread_pools()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        torch.set_num_threads(min(torch.get_num_threads(), hostcpu.usable_cpus()))      # torch loaded before this conftest
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
