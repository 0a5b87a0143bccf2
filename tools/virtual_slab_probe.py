"""Weak-scaling probe on ONE GPU: N virtual slabs, each the size of the single-GPU workload, against the single domain.
Timing is meaningless here (the slabs share one GPU); what it shows is the iteration count with the rank-local coarse
correction and that the slab protocol holds at scale."""
import json
import sys
import threading
import time

sys.path.insert(0, ".")
import numpy as np
import libfluid_amd as lfa
from libfluid_amd import scenes

name, n = sys.argv[1], int(sys.argv[2])
cfg = scenes.CONFIGS[name]
size, (blo, bhi) = list(cfg["size"]), [list(x) for x in cfg["block"]]
size[2] *= n
bhi[2] *= n
ntz = (size[2] + 7) // 8
bounds = lfa.balanced_layer_bounds(ntz, n, blo[2] // 8, (bhi[2] + 7) // 8)
hub = lfa.LocalHub(n)
sims = []
for r in range(n):
    s = lfa.Sim(size, method=cfg["method"], blending=cfg["blending"])
    if n > 1:
        s.init_local_slab(hub.h, r, bounds)
    s.seed_block(blo, bhi)
    sims.append(s)
out = [None] * n


def work(r):
    t0 = time.perf_counter()
    res = [sims[r].step_hot(0.033) for _ in range(2)]
    out[r] = (res, time.perf_counter() - t0, sims[r].counts())


th = [threading.Thread(target=work, args=(r,)) for r in range(n)]
[t.start() for t in th]
[t.join() for t in th]
print(json.dumps(dict(config=name, slabs=n, bounds=bounds, size=size,
                      iters=[[x[1] for x in o[0]] for o in out], rc=[[x[2] for x in o[0]] for o in out],
                      particles=[o[2]["particles"] for o in out], unknowns=[o[2]["unknowns"] for o in out],
                      seconds=[round(o[1], 3) for o in out])))
