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
    config.addinivalue_line("markers", "gpu2: needs TWO MI355X in one node (real RCCL ranks; skipped below two devices)")


def _gpu_present() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_present():
        import torch
        if torch.cuda.device_count() < 2:
            skip2 = pytest.mark.skip(reason="needs two GPUs in one node")
            for item in items:
                if "gpu2" in item.keywords:
                    item.add_marker(skip2)
        return
    skip = pytest.mark.skip(reason="no GPU in this container (GPU tests run via gpurun)")
    for item in items:
        if "gpu" in item.keywords or "gpu2" in item.keywords:
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


_STREAM = None


def _shared_stream():
    """One real stream shared by torch and EVERY handle of the session: tensor fills / copies
    issued by a test and the library's kernels are then ordered (a handle's own stream would
    race torch's, and so would a second handle on a stream of its own)."""
    global _STREAM
    import torch
    if _STREAM is None:
        _STREAM = torch.cuda.Stream(device=0)
        torch.cuda.set_stream(_STREAM)
    return _STREAM


@pytest.fixture(scope="session")
def gpu():
    import torch

    from rplidar_ros2_driver_amd import RplGpu
    h = RplGpu(device=0, max_samples_per_scan=32768, max_batch=4096)
    h.set_stream(_shared_stream().cuda_stream)
    yield h
    torch.cuda.synchronize()
    h.close()


# The voxel path has two block-aggregation forms that must give the same clouds: plain and
# two-class (RPLGPU_VOXEL_AGG_TWO_CLASS, normally picked from the batch's own statistics).
# `gpu_mode` runs a test once per form, each on its own handle.  (Rounds 4-5 also ran the two-kernel
# and pipelined forms here; they measured 1.36 x slower and left the library in round 6 —
# tools/dev/patches/voxel_lab_r05.diff has them.)
_MODES = {
    "default": ({}, 0),
    "two_class": ({}, 2),
}


@pytest.fixture(scope="session", params=list(_MODES))
def gpu_mode(request):
    import torch

    from rplidar_ros2_driver_amd import RplGpu
    env, agg = _MODES[request.param]
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        h = RplGpu(device=0, max_samples_per_scan=32768, max_batch=4096)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    h.set_voxel_aggregation(agg)
    h.set_stream(_shared_stream().cuda_stream)
    h.mode_name = request.param
    yield h
    torch.cuda.synchronize()
    h.close()
