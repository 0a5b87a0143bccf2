"""Times lfa_correct_collide alone on the C4 scene (after two full steps so the particles are not in seeding order)."""
import sys, time
sys.path.insert(0, ".")
import libfluid_amd as lfa
from libfluid_amd import scenes
cfg = scenes.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C4"]
s = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"])
s.seed_block(*cfg["block"])
s.enable_timing(True)
for _ in range(3):
    s.time_step(0.02)
    print({k: round(v, 3) for k, v in s.step_timings().items()})
