#!/bin/bash
# A/B of the min-waves-per-SIMD instantiations of the two finest-level streaming kernels (GPU box): isolated kernel times at C4.
cd "${GRAFT_REPO_ROOT:-.}"
for A in 1 4 5 6; do for U in 1 4 5 6; do
  if [ $A != 1 ] && [ $U != 1 ] && [ $A != $U ]; then continue; fi
  LFA_MG_MW_A=$A LFA_MG_MW_U=$U python3 - <<P
import os, sys
sys.path.insert(0, ".")
import libfluid_amd as lfa
from libfluid_amd import scenes
cfg = scenes.CONFIGS["C4"]
s = lfa.Sim(cfg["size"], method=cfg["method"])
s.seed_block(*cfg["block"])
s.enable_timing(True)
for _ in range(12):
    s.time_step(min(3.0 * s.cfl(), 0.033))
its, ms = 0, 0.0
for _ in range(8):
    _, it, _ = s.step_hot(0.033); t = s.timings(); its += it; ms += t["pcg_loop"]
print("MW_A", os.environ["LFA_MG_MW_A"], "MW_U", os.environ["LFA_MG_MW_U"], "axpy_presmooth %.4f" % s.bench_kernel("mg_axpy_presmooth", 20), "up0 %.4f" % s.bench_kernel("mg_up0", 20), "pcg_iter %.4f" % (ms / its), "it", its / 8)
s.close()
P
done; done
