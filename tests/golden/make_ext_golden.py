"""Generate tests/golden/ext_golden.npz: outputs of the SECOND WRITER of the extension spec
(oracle/ext_second_writer.py — numpy / scipy, written from SURVEY.md §8(a-ext)'s text) for every
case of tests/cases.py under the parameter sets of EXT_PARAM_SETS.  The C++ extension oracle
(oracle/oracle.cpp) and, on the GPU box, the kernels are compared against this file, i.e. against
vectors neither of them produced.

    python tests/golden/make_ext_golden.py

Small cases ship their full outputs (points, cells, counts); the large ones ship SHA-256 digests of
(cells, counts, intensity bits) and their x / y columns rounded to 1e-6 m are not needed: the test
recomputes the second writer for them where scipy / numpy are available (always, here and on the
GPU box) and uses the digests as a regression pin of the second writer itself."""
import hashlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from oracle import ext_second_writer as sw  # noqa: E402
from tests.cases import CASES  # noqa: E402

OUT = Path(__file__).resolve().parent

# (tag, keyword arguments of ext_second_writer.cloud_pipeline)
EXT_PARAM_SETS = [
    ("cloud", dict(clip_enable=True, range_max=40.0)),
    ("cloud_inv_new", dict(clip_enable=True, range_max=40.0, inverted=True, is_new_protocol=True)),
    ("cloud_q48", dict(clip_enable=True, range_max=12.0, range_min=0.5, q_min=48)),
    ("voxel", dict(clip_enable=True, range_max=40.0, voxel_enable=True, voxel_leaf=0.05)),
    ("voxel_inv", dict(clip_enable=True, range_max=40.0, voxel_enable=True, voxel_leaf=0.05, inverted=True)),
    ("voxel_leaf10_q48", dict(clip_enable=True, range_max=40.0, q_min=48, voxel_enable=True, voxel_leaf=0.10)),
    ("ror_voxel", dict(clip_enable=True, range_max=40.0, ror_enable=True, ror_radius=0.10,
                       ror_min_neighbors=2, voxel_enable=True, voxel_leaf=0.05)),
    ("ror_cloud", dict(clip_enable=True, range_max=40.0, ror_enable=True, ror_radius=0.05,
                       ror_min_neighbors=3)),
]
ROR_MAX_N = 8192          # the O(n^2) C++ oracle side of the comparison stays in seconds
FULL_MAX_POINTS = 1200    # cases with at most this many output points ship in full


def digest(pts, cells, counts):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(pts[:, 3]).view(np.uint32).tobytes())
    h.update(np.ascontiguousarray(pts[:, 2]).view(np.uint32).tobytes())
    if cells is not None:
        h.update(np.ascontiguousarray(cells, np.int32).tobytes())
        h.update(np.ascontiguousarray(counts, np.uint32).tobytes())
    # x / y at 1e-6 m: the tolerance of the spec
    h.update(np.round(pts[:, :2].astype(np.float64) * 1e6).astype(np.int64).tobytes())
    return np.frombuffer(h.digest(), np.uint8).copy()


def main():
    out = {}
    for name, nodes in CASES.items():
        for tag, kw in EXT_PARAM_SETS:
            if kw.get("ror_enable") and len(nodes) > ROR_MAX_N:
                continue
            pts, cells, counts = sw.cloud_pipeline(nodes, **kw)
            key = f"{name}__{tag}"
            out[key + "__n"] = np.uint32(len(pts))
            out[key + "__sha"] = digest(pts, cells, counts)
            if len(pts) <= FULL_MAX_POINTS:
                out[key + "__pts"] = pts
                if cells is not None:
                    out[key + "__cells"] = cells
                    out[key + "__counts"] = counts
    np.savez_compressed(OUT / "ext_golden.npz", **out)
    print(f"wrote {OUT / 'ext_golden.npz'}: {len(out)} arrays")


if __name__ == "__main__":
    main()
