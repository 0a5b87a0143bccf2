"""Late-phase cost picture (GPU box): runs a config to step N with the default (overlapped) step, then times `late_steps` steps with
the stages back to back, then the isolated PCG kernels on that state. Prints one JSON object.
usage: python tools/late_probe.py C3 550 [late_steps]"""
import json
import sys
import time
sys.path.insert(0, ".")
import numpy as np
import libfluid_amd as lfa
from libfluid_amd import scenes

name, n_steps = sys.argv[1], int(sys.argv[2])
late_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cfg = scenes.CONFIGS[name]
s = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"])
s.seed_block(*cfg["block"])
s.enable_timing(True)
med = lambda v: float(np.median(v))


def window(n, overlap):
    s.set_step_overlap(overlap)
    s.synchronize()
    t0 = time.perf_counter()
    st, its = [], 0
    for _ in range(n):
        _, it, _ = s.time_step(min(3.0 * s.cfl(), 0.033))
        its += it
        st.append(s.step_timings())
    s.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    return {"ms_per_step": round(ms, 3), "iterations": its / n,
            "stage_ms_median": {k: round(med([q[k] for q in st]), 4) for k in st[0] if k not in ("pcg_iterations", "overlapped")}}


out = {"config": name}
for k in range(20):
    s.time_step(min(3.0 * s.cfl(), 0.033))
out["early_overlapped"] = window(20, True)
out["early_serial"] = window(20, False)
out["early_counts"] = dict(s.counts(), mg_level_tiles=s.mg_level_tiles())
s.set_step_overlap(True)
for k in range(60, n_steps):
    s.time_step(min(3.0 * s.cfl(), 0.033))
out["late_overlapped"] = window(late_steps, True)
out["late_serial"] = window(late_steps, False)
out["late_counts"] = dict(s.counts(), mg_level_tiles=s.mg_level_tiles(), correction=list(s.correction_stats_ex()))
out["late_kernels_isolated_ms"] = {k: round(s.bench_kernel(k, 20), 4) for k in ("pcg_a", "mg_axpy_presmooth", "mg_down0", "mg_coarse", "mg_up0")}
out["ratio_overlapped"] = round(out["late_overlapped"]["ms_per_step"] / out["early_overlapped"]["ms_per_step"], 3)
print(json.dumps(out))
