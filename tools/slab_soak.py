import sys
sys.path.insert(0, ".")
import numpy as np
import libfluid_amd as lfa
from tests import test_gpu_slabs as T
size, block = (64, 64, 128), ((0, 0, 20), (64, 40, 100))
kw = dict(precond=lfa.PRECOND_MULTIGRID, pcg_dtype=lfa.PCG_F32)
p1, _, _ = T.run_time_steps(size, block, lfa.APIC, 25, **kw)
pn, before, after = T.run_time_steps(size, block, lfa.APIC, 25, bounds=[0, 4, 8, 12, 16], **kw)
print(len(p1), len(pn), before, after, np.isfinite(pn["pos"]).all(), np.isfinite(pn["vel"]).all())
print("com single", p1["pos"].mean(axis=0), "slabs", pn["pos"].mean(axis=0), "max |dpos|", np.abs(p1["pos"] - pn["pos"]).max())
print("ke", 0.5 * (p1["vel"] ** 2).sum(axis=1).mean(), 0.5 * (pn["vel"] ** 2).sum(axis=1).mean())
