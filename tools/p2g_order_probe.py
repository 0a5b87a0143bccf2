"""P2G scatter time in the first steps after seeding (particles arrive cell-sorted, 8 per cell): with the library's first-binning
shuffle and, for the library named by LFA_LIB_PATH built with -DLFA_NO_FIRST_SHUFFLE=1, in cell-sorted order."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libfluid_amd as lfa
from libfluid_amd import scenes
cfg = scenes.CONFIGS["C4"]
s = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"])
s.seed_block(*cfg["block"])
s.set_step_overlap(False)
s.enable_timing(True)
out = []
for k in range(6):
    s.time_step(min(3.0 * s.cfl(), 0.033))
    t = s.step_timings()
    out.append((round(t["p2g_scatter_kernel"], 3), round(t["p2g"], 3), round(t["g2p"], 3)))
print(os.environ.get("LFA_LIB_PATH", "default"), out)
s.close()
