"""Device against the fp64 CPU path (oracle/_ref = the reference compiled in place when it travelled, else the plain-C oracle) on
ONE pass of the hot path at a BASELINE configuration's FULL size, from a real dam-break state: `lead_in` whole time steps on the
device set the dam in motion, the downloaded 152-byte records go to both sides. Prints one JSON record (profiles/r05_*_parity.json):
iterations, pressure max-norm and pointwise error, face- and particle-velocity errors, CPU seconds.
  python tools/fullsize_parity.py C4 [lead_in]          (C4: ~35 GB of host memory, several minutes of CPU time)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libfluid_amd as lfa  # noqa: E402
from libfluid_amd import scenes  # noqa: E402
from oracle import loader as orc  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C3"
    lead_in = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    cfg = scenes.CONFIGS[name]
    s = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"])
    s.seed_block(*cfg["block"])
    for _ in range(lead_in):
        res, it, rc = s.time_step(min(3.0 * s.cfl(), 0.033))
        assert rc == 0
    dt = min(3.0 * s.cfl(), 0.033)
    parts = s.download_particles(into=np.zeros(s.num_particles, dtype=lfa.PARTICLE_DTYPE), write_positions=True)
    s.close()
    kind = "ref" if orc.have_ref() else "oracle"
    o = orc.CpuSim(cfg["size"], method=cfg["method"], blending=cfg["blending"], kind=kind)
    o.set_particles(parts)
    t0 = time.perf_counter()
    po, reso, ito = o.hot_step(dt)
    cpu_s = time.perf_counter() - t0
    g = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"])
    g.upload_particles(parts)
    t0 = time.perf_counter()
    res, it, rc = g.step_hot(dt)
    gpu_s = time.perf_counter() - t0
    pg = g.pressure()
    pmax = float(np.abs(po).max())
    rec = {"config": name, "grid": list(cfg["size"]), "particles": int(len(parts)), "unknowns": int(len(po)), "lead_in_steps": lead_in,
           "dt": dt, "max_particle_speed": float(np.sqrt((parts["vel"] ** 2).sum(axis=1).max())),
           "cpu": {"kind": "reference" if kind == "ref" else "port", "iterations": int(ito), "residual": float(reso), "seconds": cpu_s},
           "device": {"iterations": int(it), "residual": float(res), "rc": int(rc), "seconds_incl_first_launches": gpu_s,
                      "config": "default: fp32 state, multigrid-preconditioned CG, fp64 scalars"},
           "fluid_cells_identical": bool(np.array_equal(g.fluid_cells(), o.fluid_cells())),
           "pressure_max": pmax,
           "pressure_max_rel_err": float(np.abs(pg - po).max() / pmax),
           "pressure_pointwise_rel_err_floor_1e-4": float((np.abs(pg - po) / (np.abs(po) + 1e-4 * pmax)).max())}
    oc, gc = o.cells(), g.cells()
    rec["cell_types_identical"] = bool(np.array_equal(oc["type"], gc["type"]))
    rec["face_velocity_max"] = float(np.abs(oc["vel"]).max())
    rec["face_velocity_max_rel_err"] = float(np.abs(gc["vel"] - oc["vel"]).max() / np.abs(oc["vel"]).max())
    del oc, gc
    got, want = g.download_particles(into=parts.copy()), o.particles()
    gi = np.lexsort((got["pos"][:, 2], got["pos"][:, 1], got["pos"][:, 0]))
    wi = np.lexsort((want["pos"][:, 2], want["pos"][:, 1], want["pos"][:, 0]))
    rec["raw_cell_indices_identical"] = bool(np.array_equal(got["raw"][gi], want["raw"][wi]))
    vmax = float(np.abs(want["vel"]).max())
    rec["particle_velocity_max_rel_err"] = float(np.abs(got["vel"][gi] - want["vel"][wi]).max() / vmax)
    if cfg["method"] == scenes.APIC:
        cg = np.concatenate([got["cx"], got["cy"], got["cz"]], axis=1)[gi]
        cw = np.concatenate([want["cx"], want["cy"], want["cz"]], axis=1)[wi]
        rec["particle_c_max_rel_err"] = float(np.abs(cg - cw).max() / np.abs(cw).max())
    rec["bars"] = {"pressure_max_rel_err": 1e-4, "pressure_pointwise": 1e-3, "velocities": 1e-4, "particle_c": 2e-4}
    g.close(); o.close()
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
