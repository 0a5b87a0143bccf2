import sys, statistics
sys.path.insert(0, ".")
import libfluid_amd as lfa
from libfluid_amd import scenes
cfg = scenes.CONFIGS["C3"]
s = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"]); s.seed_block(*cfg["block"])
for k in range(550): s.time_step(min(3.0 * s.cfl(), 0.033))
s.enable_timing(True); s.set_step_overlap(False)
rows = []
for k in range(20):
    s.time_step(min(3.0 * s.cfl(), 0.033)); rows.append(s.step_timings())
print({k: round(statistics.median(r[k] for r in rows), 3) for k in ("correct_cell_index","correct_tiled_kernel","correct_collide","time_step")}, s.correction_stats())
