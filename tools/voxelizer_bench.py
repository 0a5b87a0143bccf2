"""Times the device voxelizer (lfa_voxelize_mesh: upload + triangle marking + exterior flood fill) beside the CPU checker
on the same mesh: python tools/voxelizer_bench.py [subdivisions] [radius_cells] [--cpu]
Prints one JSON line (committed under profiles/ per round)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import libfluid_amd as lfa  # noqa: E402
from libfluid_amd import scenes  # noqa: E402

sub = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 6
rad = float(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 120.0
c = rad + 8.3
pos, idx = scenes.icosphere((c, c + 0.4, c - 0.2), rad, sub)
lfa.Voxels.from_mesh(pos[:3], np.array([0, 1, 2], dtype=np.uint64)).close()  # context + module load
best = 1e30
for _ in range(3):
    t0 = time.perf_counter()
    v = lfa.Voxels.from_mesh(pos, idx, 1.0, (0.0, 0.0, 0.0))
    best = min(best, time.perf_counter() - t0)
    types = v.types()
    v.close()
n = types.size
out = {"mesh": f"icosphere, {len(idx) // 3} triangles, radius {rad} cells", "voxels": list(types.shape[::-1]),
       "surface": int((types == 2).sum()), "interior": int((types == 0).sum()),
       "device_ms": 1e3 * best, "device_Mvoxels_per_s": n / best * 1e-6}
if "--cpu" in sys.argv:
    from oracle import loader as orc
    kind = "ref" if orc.have_ref() else "oracle"
    t0 = time.perf_counter()
    _, _, want = orc.voxelize(pos, idx, 1.0, (0.0, 0.0, 0.0), kind=kind)
    cpu = time.perf_counter() - t0
    out.update({"cpu_kind": "reference" if kind == "ref" else "port", "cpu_ms": 1e3 * cpu, "identical": bool(np.array_equal(want, types))})
print(json.dumps(out))
