"""Round 6: `bench.py --config C4 --late 300` fails with 'NaN in the PCG residual at iteration 1' on the first time_step after the
kernel-timing section. Which call leaves the state that does it? (GPU box)"""
import sys
sys.path.insert(0, ".")
import libfluid_amd as lfa
from libfluid_amd import scenes
cfg_name = sys.argv[1] if len(sys.argv) > 1 else "C4"
cfg = scenes.CONFIGS[cfg_name]


def fresh(n=30):
    s = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"])
    s.seed_block(*cfg["block"])
    s.enable_timing(True)
    for _ in range(n):
        s.time_step(min(3.0 * s.cfl(), 0.033))
    return s


def try_step(s, label):
    try:
        for k in range(3):
            res, it, rc = s.time_step(min(3.0 * s.cfl(), 0.033))
        print(label, "-> ok", it, rc, flush=True)
        return s
    except Exception as e:  # noqa: BLE001
        print(label, "-> FAILED", e, flush=True)
        s.close()
        return fresh()


s = fresh()
s = try_step(s, "nothing")
s.bench_stream(1 << 30, 10)
s = try_step(s, "bench_stream")
for name in ("pcg_a", "mg_axpy_presmooth", "mg_down0", "mg_coarse", "mg_up0"):
    for reps in (1, 20):
        s.bench_kernel(name, reps)
        s = try_step(s, f"bench_kernel {name} x{reps}")
s.set_step_overlap(False)
s = try_step(s, "overlap off")
s.set_step_overlap(True)
s = try_step(s, "overlap on")
