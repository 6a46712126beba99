"""rplidar_ros2_driver_amd — MI355X (gfx950) scan-preprocessing path for RPLIDAR-class lidars.

Only what the hot path needs lives here:

* ``csrc/``      hand-written HIP kernels + the extern "C" ABI (``include/rplgpu.h``)
* ``lib/``       the built ``librplgpu.so`` (git-ignored, built by ``__graft_entry__.build()``)
* ``abi.py``     thin ctypes binding over that ABI (``include/rplgpu.h``, ``include/rplgpu_msg.h``;
                 no torch types cross it)
* ``synth.py``   deterministic synthetic raw-scan generators (bench / tests input)
* ``capsules.py`` deterministic synthetic recorded answer streams (encoder for the decode stage)
* ``sharding.py`` scan-index sharding + all-gather of the filtered clouds (RCCL / gloo)
* ``host/``      C++ mirror of the reference call sites (publish_scan / grab_scan_data)

There is no CPU fallback anywhere in this package: if the HIP library or a gfx950
device is missing, the entry points raise.
"""
from .abi import (  # noqa: F401
    NODE_DTYPE,
    Params,
    ScanMeta,
    Stamp,
    RplGpu,
    RplGpuError,
    load_library,
    library_path,
)

__all__ = [
    "NODE_DTYPE",
    "Params",
    "ScanMeta",
    "Stamp",
    "RplGpu",
    "RplGpuError",
    "load_library",
    "library_path",
]
