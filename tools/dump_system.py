"""Dumps the pressure system of a moving-dam state for the CPU model of the solver (tools/mg_line_study.py): cell types, the
unknowns' raw indices, the right-hand side, the time step and the device's iteration count. (GPU box)
usage: python tools/dump_system.py C3 40 550  -> gpurun_out/system_C3_<step>.npz"""
import sys
sys.path.insert(0, ".")
import numpy as np
import libfluid_amd as lfa
from libfluid_amd import scenes

name = sys.argv[1]
marks = sorted(int(a) for a in sys.argv[2:])
cfg = scenes.CONFIGS[name]
s = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"])
s.seed_block(*cfg["block"])
k = 0
for m in marks:
    while k < m:
        _, it_prev, _ = s.time_step(min(3.0 * s.cfl(), 0.033)); k += 1
    dt = min(3.0 * s.cfl(), 0.033)
    parts = s.download_particles(write_positions=True)
    q = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"])
    q.upload_particles(parts)
    q.hash(); q.p2g(); q.add_gravity(dt); q.build_system(dt)
    b = q.b().astype(np.float32)
    fc = q.fluid_cells().astype(np.int32)
    types = q.cells()["type"].astype(np.uint8)
    _, res, it, rc = q.solve(dt)
    np.savez_compressed(f"gpurun_out/system_{name}_{m}.npz", size=np.asarray(cfg["size"]), fluid_cells=fc, b=b, types=types, dt=dt,
                        device_iterations=it, density=1.0, cell_size=1.0)
    print(name, "step", m, "unknowns", len(fc), "dt", dt, "device iterations", it, "(in the run:", it_prev, ")", "|b|max", float(np.abs(b).max()), flush=True)
    q.close()
