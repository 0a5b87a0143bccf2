#!/bin/bash
# The other BASELINE configurations through the same bench command (C4 is the default line): JSON lines under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
TAG=${1:-r02}
python3 bench.py --config C2 --steps 30 --warmup 20 --no-cpu-baseline > gpurun_out/${TAG}_c2_bench.json 2>/tmp/c2.err || tail -3 /tmp/c2.err
python3 bench.py --config C3 --steps 30 --warmup 20 --no-cpu-baseline > gpurun_out/${TAG}_c3_bench.json 2>/tmp/c3.err || tail -3 /tmp/c3.err
python3 bench.py --config C5 --steps 20 --warmup 20 --no-cpu-baseline --obstacle --mesh > gpurun_out/${TAG}_c5_bench.json 2>/tmp/c5.err || tail -3 /tmp/c5.err
python3 bench.py --precond multilevel --steps 10 --warmup 20 --no-cpu-baseline --no-hot-path > gpurun_out/${TAG}_c4_multilevel_bench.json 2>/tmp/ml.err || tail -3 /tmp/ml.err
python3 - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02_c*_bench.json")):
    try:
        b = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, "%.3g p-steps/s" % b["value"], "%.2f ms" % b["ms_per_step"], "it", b["pcg"]["iterations_per_step"], b.get("mesher"), b.get("voxelizer"), (b.get("pcg_mic0") or {}).get("iters_per_sec"))
P
