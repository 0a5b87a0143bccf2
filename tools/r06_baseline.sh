#!/bin/bash
# Round-6 baseline on the box: default bench line, the late windows the judge's targets are quoted on, sparsity.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
T=${TAG:-base}
L="--no-cpu-baseline --no-hot-path --no-mic0-record"
timeout 600 python3 bench.py > gpurun_out/r06_${T}_c4.json 2> gpurun_out/r06_${T}_c4.err
timeout 600 python3 bench.py --config C3 --steps 20 --warmup 20 --late 550 --late-steps 20 $L > gpurun_out/r06_${T}_c3_late.json 2> gpurun_out/r06_${T}_c3_late.err
timeout 600 python3 bench.py --config C4 --steps 20 --warmup 20 --late 300 --late-steps 20 $L > gpurun_out/r06_${T}_c4_late.json 2> gpurun_out/r06_${T}_c4_late.err
python3 - $T <<'P'
import json, sys
t = sys.argv[1]
for f in ("c4", "c3_late", "c4_late"):
    try:
        o = json.loads([l for l in open(f"gpurun_out/r06_{t}_{f}.json") if l.startswith("{")][-1])
    except Exception as e:
        print(f, "failed", e); continue
    print(f, "ms/step", round(o["ms_per_step"], 3), "value", o["value"], "it", o.get("pcg", {}).get("iterations_per_step"))
    print("  stages", {k: round(v, 3) for k, v in o.get("stage_ms_median", {}).items()})
    if "late_phase" in o:
        l = o["late_phase"]
        print("  late", round(l["ms_per_step"], 3), "ratio", round(l["ms_per_step"] / o["ms_per_step"], 3), {k: round(v, 3) for k, v in l.get("stage_ms_median", {}).items()})
        print("  late extra", {k: v for k, v in l.items() if k not in ("stage_ms_median",)})
P
