import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (GPU tests run via gpurun)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    return oracle_lib.load_oracle()


@pytest.fixture(scope="session")
def reflibs():
    """The genuine reference libraries (oracle/_ref) or None when not built / not present."""
    from tests import oracle_lib
    return oracle_lib.load_ref()


@pytest.fixture(scope="session")
def gpu():
    import torch

    from rplidar_ros2_driver_amd import RplGpu
    h = RplGpu(device=0, max_samples_per_scan=32768, max_batch=4096)
    # one real stream shared by torch and the library: tensor fills / copies issued by a test
    # and the library's kernels are then ordered (the handle's own stream would race torch's)
    stream = torch.cuda.Stream(device=0)
    torch.cuda.set_stream(stream)
    h.set_stream(stream.cuda_stream)
    yield h
    torch.cuda.synchronize()
    h.close()
