"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref/libref.so = lukedan/libfluid's own sources compiled
in place by oracle/Makefile). Run in the build container only: `python tests/golden/make_golden.py`.

The reference has no tests or fixtures of its own, so these files -- inputs are regenerated from
libfluid_amd/scenes.py with fixed seeds, outputs are what the reference computed, stage by stage -- are the golden
vectors that pin oracle/oracle.c (tests/test_oracle.py) and, through it, the HIP path (tests/test_gpu_parity.py).
A fixture is data only: stage outputs as fp64/integer arrays.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import util  # noqa: E402
from oracle import loader as orc  # noqa: E402

if __name__ == "__main__":
    orc.build()
    if not orc.have_ref():
        sys.exit("oracle/_ref/libref.so is not built: /root/reference is needed to generate the golden vectors")
    for name in util.CASES:
        rec = util.staged_cpu_run(name, "ref")
        np.savez_compressed(util.golden_path(name), **rec)
        print(name, {k: int(rec[k]) for k in rec if k.startswith("iters")},
              os.path.getsize(util.golden_path(name)) // 1024, "KiB")
