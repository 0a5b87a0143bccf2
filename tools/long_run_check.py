"""Many full device steps of a configuration's dam break: invariants (no particle lost, everything finite and inside the domain,
the PCG converges, the correction never needs its fallback) and the step time as the flow develops."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import libfluid_amd as lfa
from libfluid_amd import scenes
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
cfg = scenes.CONFIGS[name]
s = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"])
s.seed_block(*cfg["block"])
npart = s.counts()["particles"]
t_sim, worst_it, worst_flag, not_conv = 0.0, 0, 0, 0
t0 = time.perf_counter()
for k in range(n):
    dt = min(3.0 * s.cfl(), 0.033)
    res, it, rc = s.time_step(dt)
    t_sim += dt
    worst_it = max(worst_it, it); not_conv += int(rc != 0)
    worst_flag = max(worst_flag, s.correction_stats()[0])
    if k % 100 == 99:
        s.synchronize()
        print(f"step {k + 1}: t = {t_sim:.3f} s, {1e3 * (time.perf_counter() - t0) / (k + 1):.2f} ms/step so far, last solve {it} iterations", flush=True)
host = np.zeros(npart, dtype=lfa.PARTICLE_DTYPE)
out = s.download_particles(into=host, write_positions=True)
size = np.asarray(cfg["size"], float)
print(name, "steps", n, "particles", s.counts()["particles"], "of", npart, "finite", bool(np.isfinite(out["pos"]).all() and np.isfinite(out["vel"]).all()),
      "inside", bool((out["pos"] >= 0).all() and (out["pos"] <= size).all()), "max iterations", worst_it, "not converged", not_conv,
      "fallback half tiles (max)", worst_flag, "max |v|", float(np.abs(out["vel"]).max()))
ids = np.sort(s.particle_ids()); print("ids intact", bool(ids[0] == 0 and ids[-1] == npart - 1 and np.all(np.diff(ids) == 1)))
