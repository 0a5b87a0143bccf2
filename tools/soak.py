"""Soak run: N device-resident full time steps of a dam break (CFL-limited dt like simulation::time_step()); prints the
iteration-count range, the step time range and checks that nothing was lost or turned NaN.
usage: python tools/soak.py [config] [steps]"""
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import libfluid_amd as lfa
from libfluid_amd import scenes

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cfg = scenes.CONFIGS[name]
s = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"])
s.seed_block(*cfg["block"])
n0 = s.counts()["particles"]
its, ms, rcs, tsim = [], [], [], 0.0
for k in range(steps):
    dt = min(3.0 * s.cfl(), 0.033)
    t0 = time.perf_counter()
    res, it, rc = s.time_step(dt)
    s.synchronize()
    ms.append(1e3 * (time.perf_counter() - t0))
    its.append(it)
    rcs.append(rc)
    tsim += dt
p = s.download_particles()
ok = bool(np.isfinite(p["pos"]).all() and np.isfinite(p["vel"]).all())
size = np.asarray(cfg["size"], dtype=np.float64)
inside = bool((p["pos"] >= 0).all() and (p["pos"] <= size).all())
print(json.dumps(dict(config=name, steps=steps, simulated_seconds=round(tsim, 3), particles=[n0, len(p)], finite=ok, inside=inside,
                      iterations=[int(min(its)), int(np.median(its)), int(max(its))], not_converged=int(sum(r != 0 for r in rcs)),
                      ms_per_step=[round(min(ms), 2), round(float(np.median(ms)), 2), round(max(ms), 2)],
                      y_extent=[float(p["pos"][:, 1].min()), float(p["pos"][:, 1].max())],
                      x_extent=[float(p["pos"][:, 0].min()), float(p["pos"][:, 0].max())])))
