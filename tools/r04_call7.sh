cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
L="--no-cpu-baseline --no-hot-path --no-mic0-record"
for C in C4 C2; do
  python3 bench.py --config $C --steps 30 --warmup 20 $L 2>/dev/null | grep "^{" > gpurun_out/r04_c_${C}.json
  python3 - <<P
import json
o=json.load(open("gpurun_out/r04_c_${C}.json")); sm=o["stage_ms_median"]
print("$C ms/step %.3f" % o["ms_per_step"], {k: round(v, 3) for k, v in sm.items()})
P
done
bash tools/hip_api_counts.sh C4 r04 | tail -3
bash tools/hip_api_counts.sh C2 r04 | tail -3
