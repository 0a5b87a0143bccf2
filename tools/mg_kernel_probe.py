import sys; sys.path.insert(0, ".")
import libfluid_amd as lfa
from libfluid_amd import scenes
cfg = scenes.CONFIGS["C4"]
sim = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"])
sim.seed_block(*cfg["block"]); sim.enable_timing(True)
for _ in range(2): res, it, rc = sim.step_hot(0.033)
t = sim.timings()
print(it, "iterations", t["pcg_loop"], "ms", 1e3 * t["pcg_iteration_mean"], "us/it")
for k in ("pcg_a", "mg_axpy_presmooth", "mg_down0", "mg_coarse", "mg_up0"):
    print(f"  {k:20s} {1e3 * sim.bench_kernel(k, 20):7.1f} us")
