#!/bin/bash
# tools/ab_late.sh VAR -- A/B of one environment switch in the C3 late window (step 550)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
VAR=$1
for V in default set default set; do
  if [ $V = set ]; then export $VAR=1; else unset $VAR; fi
  timeout 300 python3 bench.py --config C3 --steps 5 --warmup 5 --late 550 --late-steps 20 --no-cpu-baseline --no-hot-path --no-mic0-record 2>/dev/null | grep "^{" > gpurun_out/abl.json
  python3 - <<P
import json
o=json.load(open("gpurun_out/abl.json")); l=o["late_phase"]; sm=l["stage_ms_median"]
print("C3 late ${VAR} ${V}", round(l["ms_per_step"],3), {k:round(v,3) for k,v in sm.items() if k.startswith("correct")}, "fallback+rest", round(sm["correct_collide"]-sm["correct_tiled_kernel"]-sm["correct_cell_index"],3), l["correction_fallback_half_tiles"])
P
done
