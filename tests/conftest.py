import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multimodal-aspect-category-sentiment-analysis_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLD = os.path.join(ROOT, "tests", "golden")


# CPU oracles run on this many threads: a one-GPU box of the pool owns 16 cores of a much larger host, and torch's default
# (one thread per visible core) oversubscribes them ~10x (tests/oracle_thread_scan.py)
torch.set_num_threads(min(16, os.cpu_count() or 1))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def dev():
    return torch.device("cuda:0")
