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
rows = []
for k in range(lead + n):
    dt = min(3.0 * s.cfl(), 0.033)
    s.time_step(dt)
    if k >= lead:
        rows.append(s.step_timings())
print({k: round(statistics.median(r[k] for r in rows), 3) for k in rows[0]})
