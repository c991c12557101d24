import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    return os.path.exists("/dev/kfd")


def pytest_sessionstart(session):
    """GPU box: let PyTorch bring up ITS HIP runtime before libnfai_hip.so is mapped.  The tests that drive pipeline stages
    (tests/test_gpu_pipeline.py) use torch for device buffers and streams in the same process as the ctypes binding; PyTorch
    ships its own libamdhip64 and reports "No HIP GPUs are available" when it initialises after another copy of the runtime
    has opened the device (bench.py and nfai_amd.pipeline initialise torch first for the same reason)."""
    if not _has_gpu():
        return
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception as e:  # the ctypes-only tests do not need torch
        print(f"conftest: torch GPU init skipped: {e}", file=sys.stderr)


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a GPU: skip loudly instead of failing inside HIP.
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU (/dev/kfd absent)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
