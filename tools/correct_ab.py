"""Position correction alone: stage times of lfa_time_step's correction on C4 (moving dam) and on the late C3 sheet (step 550),
for the library named by LFA_LIB_PATH.  python tools/correct_ab.py [C4|C3late]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libfluid_amd as lfa
from libfluid_amd import scenes
import statistics

which = sys.argv[1] if len(sys.argv) > 1 else "C4"  # C4 | C3late | CONFIG@STEP (e.g. C4@650)
cfg = scenes.CONFIGS["C3" if which == "C3late" else which.split("@")[0]]
s = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"])
s.seed_block(*cfg["block"])
s.set_step_overlap(True)  # (the lead-in as the default step runs it)
lead = 550 if which == "C3late" else (int(which.split("@")[1]) if "@" in which else 30)
for _ in range(lead):
    s.time_step(min(3.0 * s.cfl(), 0.033))
s.set_step_overlap(False)
s.enable_timing(True)
rec = {}
for _ in range(20):
    s.time_step(min(3.0 * s.cfl(), 0.033))
    for k, v in s.step_timings().items():
        rec.setdefault(k, []).append(v)
print(which, os.environ.get("LFA_LIB_PATH", "default"), json.dumps({k: round(statistics.median(v), 4) for k, v in rec.items() if "correct" in k or k == "time_step"}))
s.close()
