"""Times the device mesher (sampling + marching cubes) on a dam-break block meshed on a 2x finer grid, beside the CPU
checker on a bounded sub-block: python tools/mesher_bench.py [block_cells] [--cpu]. Prints one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import libfluid_amd as lfa  # noqa: E402
from libfluid_amd import scenes  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 96
pts = scenes.seed_block((1, 1, 1), (1 + nb, 1 + nb, 1 + nb))["pos"]
size = (2 * nb + 8,) * 3
kw = dict(size=size, grid_offset=(0.0, 0.0, 0.0), cell_size=0.5, particle_extent=1.0, cell_radius=3)
m = lfa.Mesher(**kw)
m.sample(pts[:1000], 0.5)
best_s = best_m = 1e30
for _ in range(3):
    t0 = time.perf_counter()
    m.sample(pts, 0.5)
    t1 = time.perf_counter()
    pos, idx = m.marching_cubes()
    t2 = time.perf_counter()
    best_s, best_m = min(best_s, t1 - t0), min(best_m, t2 - t1)
npts = (size[0] + 1) ** 3
out = {"workload": f"{len(pts)} particles ({nb}^3 cells x 8), sampling grid {size[0]}^3 cells, cell_radius 3",
       "sample_ms_incl_upload": 1e3 * best_s, "marching_cubes_ms_incl_download": 1e3 * best_m,
       "grid_points_per_s": npts / best_s, "vertices": int(len(pos)), "triangles": int(len(idx) // 3)}
if "--cpu" in sys.argv:
    from oracle import loader as orc
    kind = "ref" if orc.have_ref() else "oracle"
    nc = 24
    sub = scenes.seed_block((1, 1, 1), (1 + nc, 1 + nc, 1 + nc))["pos"]
    skw = dict(kw, size=(2 * nc + 8,) * 3)
    t0 = time.perf_counter()
    wp, wi = orc.mesher_mesh(sub, kind=kind, r=0.5, **skw)
    cpu = time.perf_counter() - t0
    ms = lfa.Mesher(**skw)
    gp, gi = ms.generate_mesh(sub, 0.5)
    out.update({"cpu_kind": "reference (OpenMP sampling)" if kind == "ref" else "port (serial)", "cpu_sample": f"{nc}^3-cell block, {(2 * nc + 9) ** 3} grid points",
                "cpu_ms": 1e3 * cpu, "cpu_grid_points_per_s": (2 * nc + 9) ** 3 / cpu,
                "identical": bool(np.array_equal(wp, gp, equal_nan=True) and np.array_equal(wi, gi))})
print(json.dumps(out))
