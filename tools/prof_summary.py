"""Summarise a tools/prof.sh output directory: per-kernel average duration from the
kernel trace and per-kernel average PMC counter values (per dispatch)."""
import csv
import glob
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = name.replace("(anonymous namespace)::", "").split("(")[0]
    if name.startswith("void "):
        name = name[5:]
    return name.replace("rpl::", "")


for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats (rocprofv3 --kernel-trace --stats):", f.split(out)[-1])
    for row in csv.DictReader(open(f)):
        print("  %-44s calls=%-4s avg_us=%10.1f min_us=%10.1f max_us=%10.1f pct=%s" % (
            short(row["Name"])[:44], row["Calls"], float(row["AverageNs"]) / 1e3,
            float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3, row["Percentage"]))

agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("== PMC (average per dispatch)")
for k, d in agg.items():
    if not k.startswith("k_"):
        continue
    print(" ", k)
    for c, v in sorted(d.items()):
        print("    %-24s n=%-3d avg=%16.1f" % (c, len(v), sum(v) / len(v)))

# HBM traffic per launch, as MI355X_MICROARCH.md prescribes: separate --pmc passes;
# bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 — FETCH_SIZE is in KiB and reports half of a
# wide coalesced streaming read on gfx950 (TCC_EA0_RDREQ x 64 B for 128-B requests).
import json
import os
import hashlib
# the state of the kernel sources the counters were measured on (bench.py refuses a traffic figure
# whose sources changed since): the kernel's own file AND what sets its launch geometry, its store
# sizing and its device helpers — rpl_device.hpp, rpl_launch.hpp, rplgpu_api.hip (round 4 keyed on
# the kernel file alone: a change of the launcher could change the traffic unnoticed)
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SRC = {"k_cloud_voxel": "rpl_voxel.hip", "k_voxel_runs": "rpl_voxel.hip", "k_voxel_cells": "rpl_voxel.hip",
        "k_ascend_stream": "rpl_kernels.hip", "k_ascend": "rpl_kernels.hip", "k_laserscan_a": "rpl_laserscan.hip",
        "k_ror_mask": "rpl_ror.hip", "k_decode": "rpl_decode.hip"}
_COMMON = ["rpl_device.hpp", "rpl_launch.hpp", "rplgpu_api.hip"]


def _sha(name):
    files = ["rplidar_ros2_driver_amd/csrc/" + f for f in [_SRC.get(name, "rpl_voxel.hip")] + _COMMON]
    h = hashlib.sha256()
    try:
        for rel in files:
            h.update(open(os.path.join(_ROOT, rel), "rb").read())
        return files, h.hexdigest()
    except OSError:
        return files, None


traffic = {}
for k, d in agg.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        f = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"])
        w = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        name = k.split("<")[0]
        traffic[name] = {
            "hbm_bytes_per_launch": int((2.0 * f + w) * 1024),
            "fetch_size_kib_raw": f, "write_size_kib": w,
            "rdreq_x128B": int(sum(d.get("TCC_EA0_RDREQ_sum", [0])) / max(len(d.get("TCC_EA0_RDREQ_sum", [1])), 1) * 128),
            "scans": int(os.environ.get("RPL_PROF_SCANS", "4096")),
            "samples_per_scan": int(os.environ.get("RPL_PROF_SAMPLES", "32000")),
            "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py (tools/prof.sh), FETCH x2 gfx950 correction",
            "source_files": _sha(name)[0], "source_sha256": _sha(name)[1],
        }
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print("== traffic.json:", json.dumps({k: v["hbm_bytes_per_launch"] for k, v in traffic.items()}))
