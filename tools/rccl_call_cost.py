"""What a transport call costs at least (GPU box, one GPU): the PCG loop of C3 / C4 as a single domain, and on a ONE-rank RCCL slab
decomposition with the slab mode of the multigrid hierarchy forced (LFA_MG_DIST_SINGLE=1): every all-reduce of the protocol then goes
through ncclAllReduce with one rank (no link is crossed, exchanges with no neighbour send nothing) - the difference per iteration
divided by the all-reduces per iteration is the enqueue + kernel cost of one RCCL call, the floor of a real one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libfluid_amd as lfa
from libfluid_amd import scenes

for cfg_name in sys.argv[1:] or ["C3", "C4"]:
    cfg = scenes.CONFIGS[cfg_name]
    out = {}
    for mode in ("single", "rccl1"):
        if mode == "rccl1":
            os.environ["LFA_MG_DIST_SINGLE"] = "1"
        else:
            os.environ.pop("LFA_MG_DIST_SINGLE", None)
        s = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"])
        if mode == "rccl1":
            s.init_rccl_slab(0, 1, lfa.rccl_unique_id(), [0, (cfg["size"][2] + 7) // 8])
        s.seed_block(*cfg["block"])
        s.enable_timing(True)
        for _ in range(3):
            s.step_hot(0.01)
        its, ms = 0, 0.0
        for _ in range(6):
            _, it, _ = s.step_hot(0.01)
            t = s.timings()
            its += it
            ms += t["pcg_loop"]
        out[mode] = (ms / its, its / 6, s.solver_stats())
        s.close()
    a, b = out["single"], out["rccl1"]
    calls = b[2]["transport_calls_per_iteration"]
    print(cfg_name, "single %.4f ms/iteration (%.1f it)" % a[:2], "one-rank RCCL slabs %.4f ms/iteration (%.1f it)" % b[:2],
          "transport calls per iteration", calls, "=> %.1f us per call" % (1e3 * (b[0] - a[0]) / max(calls, 1)), b[2])
