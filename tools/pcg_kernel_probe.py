"""Times the PCG kernels (fused and unfused) on one configuration: python tools/pcg_kernel_probe.py [C4] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libfluid_amd as lfa  # noqa: E402
from libfluid_amd import scenes  # noqa: E402

cfg_name = sys.argv[1] if len(sys.argv) > 1 else "C4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = scenes.CONFIGS[cfg_name]
for fused in (1, 0):
    precond = {"tiled": lfa.PRECOND_MIC0_TILED, "multilevel": lfa.PRECOND_MULTILEVEL, "multigrid": lfa.PRECOND_MULTIGRID}[os.environ.get("PROBE_PRECOND", "multilevel")]
    sim = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"], pcg_fused=fused, precond=precond,
                  max_iterations=int(os.environ.get("PROBE_MAXIT", "200")))
    sim.seed_block(*cfg["block"])
    sim.enable_timing(True)
    for _ in range(2):
        res, it, rc = sim.step_hot(0.033)
    t = sim.timings()
    n = sim.counts()["unknowns"]
    print(f"{cfg_name} fused={fused}: {it} iterations, pcg_loop {t['pcg_loop']:.3f} ms, {1e3 * t['pcg_iteration_mean']:.1f} us/iteration, "
          f"n={n}")
    if precond == lfa.PRECOND_MULTIGRID:
        sim.close()
        break
    names = (("pcg_a", "pcg_b") if fused else ()) + ("spmv_dot", "axpy_max", "mic_apply_dot", "update_s", "mic_fine", "coarse_levels")
    for k in names:
        ms = sim.bench_kernel(k, reps)
        print(f"   {k:16s} {1e3 * ms:8.1f} us")
    sim.close()
