"""Device time of consecutive full time steps of C4 (CFL-limited dt): advect+collide, correct+collide, whole step."""
import sys
sys.path.insert(0, ".")
import libfluid_amd as lfa
from libfluid_amd import scenes
cfg = scenes.CONFIGS["C4"]
s = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"])
s.seed_block(*cfg["block"])
s.enable_timing(True)
for k in range(6):
    dt = min(3.0 * s.cfl(), 0.033)
    s.time_step(dt)
    print(round(dt, 4), {k: round(v, 2) for k, v in s.step_timings().items()})
