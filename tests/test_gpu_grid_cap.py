"""The per-iteration kernels of the pressure solve walk their tiles with a software pipeline whose look-ahead state (table rows,
the tile ids of a wave's next 64 slots) is refilled as the wave goes. With the tuned grid a wave sees a handful of tiles; here the
grid is cut to 1 and 3 workgroups (LFA_PCG_GRID_CAP, read once per process: separate processes), so a wave walks hundreds of tiles
and passes the 64-slot refill several times. Same solves as with the default grid up to the grouping of the partial sums."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import json, sys, zlib
import numpy as np
sys.path.insert(0, %r)
import libfluid_amd as lfa
s = lfa.Sim((96, 96, 96), method=lfa.APIC)
s.seed_block((0, 0, 0), (48, 96, 96))
out = []
for _ in range(4):
    s.time_step(min(3.0 * s.cfl(), 0.033))
    p = np.asarray(s.pressure(), dtype=np.float64)
    st = s.solver_stats()
    out.append({"it": int(st["iterations"]), "pmax": float(np.abs(p).max()), "psum": float(p.sum()), "n": int(np.count_nonzero(p))})
parts = s.download_particles(write_positions=True)
pos = np.asarray(parts["pos"], dtype=np.float64)
print(json.dumps({"steps": out, "pos_sum": float(pos.sum()), "np": int(pos.shape[0])}))
s.close()
""" % ROOT


def run(cap):
    env = dict(os.environ)
    env.pop("LFA_PCG_GRID_CAP", None)
    if cap is not None:
        env["LFA_PCG_GRID_CAP"] = str(cap)
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("cap", [1, 3])
def test_a_wave_that_walks_hundreds_of_tiles_solves_what_the_tuned_grid_solves(cap):
    ref, got = run(None), run(cap)
    assert got["np"] == ref["np"]
    for a, b in zip(ref["steps"], got["steps"]):
        assert a["n"] == b["n"] and a["n"] > 100000
        assert abs(a["it"] - b["it"]) <= 1, (a, b)
        assert b["pmax"] == pytest.approx(a["pmax"], rel=2e-5)
        assert b["psum"] == pytest.approx(a["psum"], rel=2e-4)
    assert got["pos_sum"] == pytest.approx(ref["pos_sum"], rel=1e-6)
