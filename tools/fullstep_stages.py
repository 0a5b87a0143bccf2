"""Device time of consecutive full time steps of C4 (CFL-limited dt) after a lead-in: per-stage medians (lfa_get_step_timings)."""
import statistics
import sys
sys.path.insert(0, ".")
import libfluid_amd as lfa
from libfluid_amd import scenes
cfg = scenes.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C4"]
lead, n = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (20, 10)
s = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"])
s.seed_block(*cfg["block"])
s.enable_timing(True)
import time
for k in range(lead):
    s.time_step(min(3.0 * s.cfl(), 0.033))
for mode in (1, 0, 1, 0):
    s.set_step_overlap(mode)
    rows = []
    s.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        s.time_step(min(3.0 * s.cfl(), 0.033))
        rows.append(s.step_timings())
    s.synchronize()
    print("overlap", mode, "wall ms/step", round(1e3 * (time.perf_counter() - t0) / n, 3),
          {k: round(statistics.median(r[k] for r in rows), 3) for k in rows[0]})
