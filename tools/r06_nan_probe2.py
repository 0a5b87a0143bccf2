"""Round 6: where does the C4 run with timing enabled (bench.py's loop) turn NaN? (GPU box)"""
import sys
sys.path.insert(0, ".")
import libfluid_amd as lfa
from libfluid_amd import scenes
cfg = scenes.CONFIGS["C4"]
timing, kernels, serial = [a == "1" for a in sys.argv[1:4]]
s = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"])
s.seed_block(*cfg["block"])
s.enable_timing(timing)
k = 0
try:
    for k in range(320):
        if k == 40 and serial:
            s.set_step_overlap(False)
        if k == 60 and serial:
            s.set_step_overlap(True)
        if k == 60 and kernels:
            s.bench_stream(1 << 30, 10)
            for name in ("pcg_a", "mg_axpy_presmooth", "mg_down0", "mg_coarse", "mg_up0"):
                s.bench_kernel(name, 20)
        dt = min(3.0 * s.cfl(), 0.033)
        res, it, rc = s.time_step(dt)
        if timing:
            s.step_timings()
    print(sys.argv[1:], "ok to step", k, it)
except Exception as e:  # noqa: BLE001
    print(sys.argv[1:], "FAILED at step", k, "dt", dt, e, s.solver_stats())
