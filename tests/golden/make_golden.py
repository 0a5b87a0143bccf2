"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref/libref.so = lukedan/libfluid's own sources compiled
in place by oracle/Makefile). Run in the build container only: `python tests/golden/make_golden.py`.

The reference has no tests or fixtures of its own, so these files -- inputs are regenerated from
libfluid_amd/scenes.py with fixed seeds, outputs are what the reference computed -- are the golden vectors that pin
oracle/oracle.c (tests/test_oracle.py) and, through it, the HIP path (tests/test_gpu_parity.py):
  <case>.npz         every stage of two hot-path passes (tests/util.py:staged_cpu_run)
  fullstep_flip.npz  particles after three full simulation::time_step(dt) calls (tests/test_host_class.py)
  next_stages.npz    positions after advect+collide and after correct+collide (tests/test_next_rows.py)
A fixture is data only: stage outputs as fp64/integer arrays.
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import util  # noqa: E402
from oracle import loader as orc  # noqa: E402


def make_fullstep():
    from tests.test_host_class import fullstep_inputs
    c, parts, solid = fullstep_inputs()
    s = orc.CpuSim(c["size"], method=c["method"], blending=c["blend"], kind="ref")
    s.set_solid_cells(solid)
    s.set_particles(parts)
    iters = []
    for _ in range(c["steps"]):
        res, it = C.c_double(0), C.c_uint64(0)
        s.L.time_step(s.h, c["dt"], C.byref(res), C.byref(it))  # simulation::time_step, src/simulation.cpp:43-125
        iters.append(it.value)
    out = s.particles()
    ids = np.rint(out["cx"][:, 0]).astype(np.int64)
    assert np.array_equal(np.sort(ids), np.arange(len(parts)))
    out = out[np.argsort(ids)]
    np.savez_compressed(util.golden_path("fullstep_flip"), pos=out["pos"], vel=out["vel"], raw=out["raw"],
                        iters=np.array(iters))
    print("fullstep_flip", iters, os.path.getsize(util.golden_path("fullstep_flip")) // 1024, "KiB")


def make_next_stages():
    """Particle stages around the hot path: advect -> collide -> hash -> correct -> collide."""
    from tests.test_next_rows import next_inputs, run_next_cpu
    rec = run_next_cpu("ref")
    np.savez_compressed(util.golden_path("next_stages"), **rec)
    print("next_stages", os.path.getsize(util.golden_path("next_stages")) // 1024, "KiB")


if __name__ == "__main__":
    orc.build()
    if not orc.have_ref():
        sys.exit("oracle/_ref/libref.so is not built: /root/reference is needed to generate the golden vectors")
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    for name in util.CASES:
        if only and name not in only:
            continue
        rec = util.staged_cpu_run(name, "ref")
        np.savez_compressed(util.golden_path(name), **rec)
        print(name, {k: int(rec[k]) for k in rec if k.startswith("iters")},
              os.path.getsize(util.golden_path(name)) // 1024, "KiB")
    if not only or "fullstep_flip" in only:
        make_fullstep()
    if not only or "next_stages" in only:
        make_next_stages()
