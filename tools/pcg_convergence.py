"""Iteration counts of the pressure solve for BASELINE configs: tiled vs exact MIC(0), f32 vs f64 vectors."""
import sys, time, json
sys.path.insert(0, ".")
import numpy as np
import libfluid_amd as lfa
from libfluid_amd import scenes

for name in [a for a in sys.argv[1:] if not a.startswith("--")]:
    cfg = scenes.CONFIGS[name]
    modes = ((lfa.PRECOND_MULTILEVEL, lfa.PCG_F32), (lfa.PRECOND_MULTILEVEL, lfa.PCG_F64), (lfa.PRECOND_MIC0_TILED, lfa.PCG_F32))
    if "--all" in sys.argv:
        modes += ((lfa.PRECOND_MIC0_EXACT, lfa.PCG_F64), (lfa.PRECOND_MIC0_TILED, lfa.PCG_F64), (lfa.PRECOND_MIC0_EXACT, lfa.PCG_F32))
    for precond, dtype in modes:
        s = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"], precond=precond, pcg_dtype=dtype,
                    max_iterations=3000)
        s.seed_block(*cfg["block"])
        s.hash(); s.p2g(); s.add_gravity(0.033)
        t0 = time.perf_counter()
        res = lfa.C.c_double(0); it = lfa.C.c_uint64(0)
        rc = s.lib.lfa_pcg_solve(s.h, 0.033, lfa.C.byref(res), lfa.C.byref(it))
        s.synchronize()
        dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        rc = s.lib.lfa_pcg_solve(s.h, 0.033, lfa.C.byref(res), lfa.C.byref(it))
        s.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps(dict(config=name, precond=["tiled", "exact", "multilevel"][precond], dtype=["f32", "f64"][dtype], rc=rc,
                              iters=it.value, residual=res.value, solve_ms=dt * 1e3)), flush=True)
        s.close()
