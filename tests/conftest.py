import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "slow: a GPU test that needs a minute or more of one-thread oracle time (full-size bit comparisons); part of -m gpu")
    config.addinivalue_line("markers", "native_threshold: (test_gpu_shard_group_sums) the library's own size threshold for the group sums, no override")


def _have_gpu() -> bool:
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (/dev/kfd absent)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
