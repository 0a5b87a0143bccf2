"""What ONE rank of the fixed C4 domain on N z-slabs has to compute, measured: the rank's slab as a domain of its own
(512 x 512 x 256/N cells... the block's z extent split N ways, walls where the neighbour ranks would be), its share of the block's
particles, the default step. Kernel times of this run are what DESIGN.md's multi-GPU projection is built from - NOT a multi-GPU
measurement: no exchange happens, and the replicated coarse levels of the real decomposition span the whole 512^3 domain (their
cost is taken from the single-GPU C4 run). (GPU box)  python tools/slab_rank_alone.py 8"""
import json
import sys
import time
sys.path.insert(0, ".")
import numpy as np
import libfluid_amd as lfa
from libfluid_amd import scenes

n_ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = scenes.CONFIGS["C4"]
(blo, bhi) = cfg["block"]
nz = (bhi[2] - blo[2]) // n_ranks  # the fluid's z extent split evenly (lfa.balanced_layer_bounds)
size = (cfg["size"][0], cfg["size"][1], nz)
s = lfa.Sim(size, method=cfg["method"], blending=cfg["blending"])
s.seed_block(blo, (bhi[0], bhi[1], nz))
s.enable_timing(True)
for _ in range(20):
    s.time_step(min(3.0 * s.cfl(), 0.033))
med = lambda v: float(np.median(v))
out = {"ranks": n_ranks, "slab_grid": list(size), "particles": s.counts()["particles"]}
for overlap in (True, False):
    s.set_step_overlap(overlap)
    s.synchronize(); t0 = time.perf_counter(); st = []; its = 0
    for _ in range(20):
        _, it, _ = s.time_step(min(3.0 * s.cfl(), 0.033)); its += it; st.append(s.step_timings())
    s.synchronize()
    out["overlapped" if overlap else "serial"] = {"ms_per_step": round(1e3 * (time.perf_counter() - t0) / 20, 3), "iterations": its / 20,
                                                 "stage_ms_median": {k: round(med([q[k] for q in st]), 4) for k in st[0] if k not in ("pcg_iterations", "overlapped")}}
out["counts"] = dict(s.counts(), mg_level_tiles=s.mg_level_tiles())
out["kernels_isolated_ms"] = {k: round(s.bench_kernel(k, 20), 4) for k in ("pcg_a", "mg_axpy_presmooth", "mg_down0", "mg_coarse", "mg_up0")}
print(json.dumps(out))
