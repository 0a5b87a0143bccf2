"""How many half tiles of the position correction take the slow fallback over the first steps of a configuration's dam break."""
import sys
sys.path.insert(0, ".")
import libfluid_amd as lfa
from libfluid_amd import scenes
cfg = scenes.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C4"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
s = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"])
s.seed_block(*cfg["block"])
worst = (0, 0, 0)
for k in range(n):
    s.time_step(min(3.0 * s.cfl(), 0.033))
    f, t = s.correction_stats()
    if f > worst[0]: worst = (f, t, k)
    if k % 20 == 19: print("step", k + 1, "flagged", f, "of", t, flush=True)
print("worst", worst)
