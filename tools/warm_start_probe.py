"""PCG iterations and step time of the moving dam with and without lfa_params.pcg_warm_start (the solve starts from the previous
step's pressure instead of p = 0).  python tools/warm_start_probe.py C4 [steps]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libfluid_amd as lfa
from libfluid_amd import scenes

name = sys.argv[1] if len(sys.argv) > 1 else "C4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
cfg = scenes.CONFIGS[name]
out = {}
for warm in (0, 1):
    s = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"], pcg_warm_start=warm)
    s.seed_block(*cfg["block"])
    its = []
    for k in range(steps):
        if k == 20:
            s.synchronize(); t0 = time.perf_counter()
        r, it, rc = s.time_step(min(3.0 * s.cfl(), 0.033))
        assert rc == 0 and r < 1e-6
        its.append(int(it))
    s.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / (steps - 20)
    out["warm" if warm else "cold"] = {"ms_per_step": ms, "iterations": its}
    s.close()
print(json.dumps(out))
