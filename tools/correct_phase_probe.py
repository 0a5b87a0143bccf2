"""Where a work item of the LDS-tiled position correction spends its time (GPU box; needs the CORR_PROFILE variant:
python libfluid_amd/build.py variant corr_prof particles.hip -DCORR_PROFILE, LFA_LIB_PATH=libfluid_amd/variants/corr_prof.so).
Wall-clock ticks (100 MHz) of thread 0 per phase, summed over the work items of 10 steps."""
import ctypes as C
import sys
sys.path.insert(0, ".")
import libfluid_amd as lfa
from libfluid_amd import scenes
which = sys.argv[1] if len(sys.argv) > 1 else "C4"
cfg = scenes.CONFIGS["C3" if which == "C3late" else which]
s = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"])
s.seed_block(*cfg["block"])
s.set_step_overlap(False)
for _ in range(550 if which == "C3late" else 30):
    s.time_step(min(3.0 * s.cfl(), 0.033))
lib = lfa.load_library()
out = (C.c_uint64 * 8)()
s.synchronize()
lib.lfa_debug_corr_prof(out, 1)
for _ in range(10):
    s.time_step(min(3.0 * s.cfl(), 0.033))
s.synchronize()
lib.lfa_debug_corr_prof(out, 0)
v = list(out)
items = max(v[7], 1)
us = lambda t: t / items / 100.0  # 100 MHz constant clock
print(f"{which}: {items // 10} work items per step; per item: counts+scans {us(v[0]):.2f} us, staging {us(v[1]):.2f} us, thread 0's particles {us(v[2]):.2f} us (of which the candidate walks {us(v[4]):.2f}), "
      f"tail wait {us(v[3]):.2f} us; staged {v[5] / items:.0f}, own {v[6] / items:.0f} particles per item")
