"""One rank of a multi-process z-slab run over the shared-memory transport (lfa_dist_init_shm); started by
tests/test_gpu_slabs.py, one process per rank, all on the GPU of the box. Writes this rank's result to `out`."""
import json
import sys

import numpy as np

import libfluid_amd as lfa
from tests import util


def main():
    spec = json.loads(sys.argv[1])
    rank, bounds = spec["rank"], spec["bounds"]
    n = len(bounds) - 1
    s = lfa.Sim(spec["size"], method=spec["method"], blending=0.95, precond=spec["precond"], pcg_dtype=spec["pcg_dtype"])
    s.init_shm_slab(spec["name"], rank, n, bounds)
    s.seed_block(*spec["block"])
    iters = []
    for _ in range(spec["hot_steps"]):
        res, it, rc = s.step_hot(util.DT)
        assert rc == 0
        iters.append(it)
    lo, hi = s.slab()
    cells = s.cells()
    before = s.num_particles
    for _ in range(spec["full_steps"]):
        res, it, rc = s.time_step(util.DT)
        assert rc == 0
    parts = s.download_particles()
    ids = s.particle_ids()
    calls = s.solver_stats()
    s.close()
    np.savez(spec["out"], cells=cells, parts=parts, ids=ids, iters=np.array(iters), slab=np.array([lo, hi]), before=before,
             transport_calls=calls["transport_calls_per_iteration"],
             waits_given_up=calls["device_waits_given_up"])


if __name__ == "__main__":
    main()
