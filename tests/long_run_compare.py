"""Long-run comparison (not a pytest module): N full time steps of a dam break on the device and in the CPU checker
(the real reference when oracle/_ref is built, else the plain-C restatement); prints bulk statistics of both per step.
Trajectories of single particles diverge chaotically after a few dozen steps, bulk quantities (centre of mass, front
position, height) must keep agreeing. usage: python tests/long_run_compare.py [steps] [n] [apic|flip|pic]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import libfluid_amd as lfa  # noqa: E402
from libfluid_amd import scenes  # noqa: E402
from oracle import loader as orc  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
method = sys.argv[3] if len(sys.argv) > 3 else "apic"  # apic | flip (blend 0.95) | pic
M_ORC = {"apic": orc.APIC, "flip": orc.FLIP, "pic": orc.PIC}[method]
M_LFA = {"apic": lfa.APIC, "flip": lfa.FLIP_BLEND, "pic": lfa.PIC}[method]
dt = 0.005
size, block = (n, n, n), ((0, 0, 0), (n // 2, n // 2, n // 2))
parts = scenes.seed_block(*block)
parts["cx"][:, 0] = np.arange(len(parts))
kind = "ref" if os.path.exists(orc.REF_SO) else "oracle"
cpu = orc.CpuSim(size, method=M_ORC, blending=0.95, kind=kind)
cpu.set_particles(parts)
gpu = lfa.Sim(size, method=M_LFA, blending=0.95, pcg_dtype=lfa.PCG_F32)
gpu.upload_particles(parts)
# the checker's own sensitivity: the same run from positions perturbed by 1e-6 cells (below the device's fp32 resolution of
# a velocity, far above fp64 rounding) - the yardstick for what "agreement" can mean on a chaotic splash
pert = parts.copy()
pert["pos"] += np.random.default_rng(1).uniform(-1e-6, 1e-6, size=pert["pos"].shape)
pert["old_pos"] = pert["pos"]
cpu2 = orc.CpuSim(size, method=M_ORC, blending=0.95, kind=kind)
cpu2.set_particles(pert)


def stats(pos, vel):
    return [float(pos[:, 0].mean()), float(pos[:, 1].mean()), float(pos[:, 0].max()), float(pos[:, 1].max()),
            float(0.5 * (vel ** 2).sum(axis=1).mean())]


rows = []
for k in range(steps):
    res, it = C.c_double(0), C.c_uint64(0)
    cpu.L.time_step(cpu.h, dt, C.byref(res), C.byref(it))
    res2, it2 = C.c_double(0), C.c_uint64(0)
    cpu2.L.time_step(cpu2.h, dt, C.byref(res2), C.byref(it2))
    _, git, rc = gpu.time_step(dt)
    assert rc == 0
    if k % 10 == 9 or k == steps - 1:
        a = cpu.particles()
        b = gpu.download_particles(into=parts.copy(), write_positions=True)
        a2 = cpu2.particles()
        rows.append(dict(step=k + 1, cpu_iters=int(it.value), gpu_iters=int(git), cpu=stats(a["pos"], a["vel"]), gpu=stats(b["pos"], b["vel"]),
                         cpu_perturbed=stats(a2["pos"], a2["vel"])))
        print(json.dumps(rows[-1]), flush=True)
dev = max(max(abs(x - y) for x, y in zip(r["cpu"][:4], r["gpu"][:4])) for r in rows)
dev2 = max(max(abs(x - y) for x, y in zip(r["cpu"][:4], r["cpu_perturbed"][:4])) for r in rows)
print(json.dumps(dict(kind=kind, method=method, particles=len(parts), steps=steps, dt=dt, max_abs_dev_of_com_and_extents_cells=dev,
                      same_for_the_checker_perturbed_by_1e_6_cells=dev2,
                      final_ke_cpu=rows[-1]["cpu"][4], final_ke_gpu=rows[-1]["gpu"][4])))
