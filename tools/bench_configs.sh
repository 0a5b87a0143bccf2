#!/bin/bash
# The other BASELINE configurations through the same bench command (C4 is the default line): JSON lines under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
TAG=${1:-r05}
python3 bench.py --config C2 --steps 30 --warmup 20 --no-cpu-baseline > gpurun_out/${TAG}_c2_bench.json 2>/tmp/c2.err || tail -3 /tmp/c2.err
python3 bench.py --config C3 --steps 30 --warmup 20 --no-cpu-baseline > gpurun_out/${TAG}_c3_bench.json 2>/tmp/c3.err || tail -3 /tmp/c3.err
python3 bench.py --config C5 --steps 20 --warmup 20 --no-cpu-baseline --obstacle --mesh > gpurun_out/${TAG}_c5_bench.json 2>/tmp/c5.err || tail -3 /tmp/c5.err
python3 bench.py --precond multilevel --steps 10 --warmup 20 --no-cpu-baseline --no-hot-path > gpurun_out/${TAG}_c4_multilevel_bench.json 2>/tmp/ml.err || tail -3 /tmp/ml.err
# BASELINE configs[1]: the P2G variants side by side (global atomics vs LDS-binned), same workload
python3 bench.py --config C2 --p2g atomic --steps 30 --warmup 20 --no-cpu-baseline --no-mic0-record > gpurun_out/${TAG}_c2_p2g_atomic_bench.json 2>/tmp/c2a.err || tail -3 /tmp/c2a.err
python3 bench.py --p2g atomic --steps 20 --warmup 20 --no-cpu-baseline --no-mic0-record --no-hot-path > gpurun_out/${TAG}_c4_p2g_atomic_bench.json 2>/tmp/c4a.err || tail -3 /tmp/c4a.err
# late in the run: the sheet has spread over many partly filled tiles
python3 bench.py --config C3 --late 550 --late-steps 20 --steps 30 --warmup 20 --no-cpu-baseline --no-hot-path --no-mic0-record --no-kernel-timing > gpurun_out/${TAG}_c3_late_bench.json 2>/tmp/c3l.err || tail -3 /tmp/c3l.err
TAG=$TAG python3 - <<'P'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/" + os.environ["TAG"] + "_c*_bench.json")):
    try:
        b = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, "%.3g p-steps/s" % b["value"], "%.2f ms" % b["ms_per_step"], "it", b["pcg"]["iterations_per_step"], b.get("mesher"), b.get("voxelizer"), (b.get("pcg_mic0") or {}).get("iters_per_sec"),
          "p2g_scatter_ms", (b.get("stage_ms_median") or {}).get("p2g_scatter_kernel"), "late", (b.get("late_phase") or {}).get("ratio_to_timed_region"))
P
