"""Does a handle step more slowly after pcg_scrub (the hierarchy freed and rebuilt)? bench.py's late window follows lfa_bench_kernel.
(GPU box) python tools/r06_scrub_probe.py C4 300"""
import sys, time
sys.path.insert(0, ".")
import libfluid_amd as lfa
from libfluid_amd import scenes
name, n = sys.argv[1], int(sys.argv[2])
cfg = scenes.CONFIGS[name]
s = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"])
s.seed_block(*cfg["block"])
s.enable_timing(True)
for _ in range(n):
    s.time_step(min(3.0 * s.cfl(), 0.033))


def window(k=20):
    s.synchronize(); t0 = time.perf_counter(); it = 0
    for _ in range(k):
        it += s.time_step(min(3.0 * s.cfl(), 0.033))[1]
    s.synchronize()
    return round(1e3 * (time.perf_counter() - t0) / k, 3), it / k


print(name, "before", window(), window())
s.bench_kernel("pcg_a", 20)
print(name, "after one bench_kernel (scrub)", window(), window())
for k in ("pcg_a", "mg_axpy_presmooth", "mg_down0", "mg_coarse", "mg_up0"):
    s.bench_kernel(k, 20)
s.bench_stream(1 << 30, 10)
print(name, "after bench.py's sequence", window(), window())
