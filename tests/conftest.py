import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# a product call whose fp16 operators leave their range is an ERROR in the test suite (default outside: RuntimeWarning)
os.environ.setdefault("FLOAT_AMD_RANGE", "raise")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    import torch
    # device_count() does not create a HIP context in this process (is_available() does): tests/test_00_ranks8_gpu.py needs the
    # GPU's eight compute process slots for its eight ranks
    has_gpu = torch.cuda.device_count() > 0
    skip_gpu = pytest.mark.skip(reason="no GPU visible")
    skip_ref = pytest.mark.skip(reason="/root/reference not present")
    has_ref = os.path.isdir("/root/reference/src/nodes")
    for item in items:
        if "gpu" in item.keywords and not has_gpu:
            item.add_marker(skip_gpu)
        if "reference" in item.keywords and not has_ref:
            item.add_marker(skip_ref)
