cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
L="--no-cpu-baseline --no-mic0-record"
for C in C4 C2 C4 C2; do
  python3 bench.py --config $C --steps 30 --warmup 20 $L 2>/dev/null | grep "^{" > /tmp/b.json
  python3 - <<P
import json
o=json.load(open("/tmp/b.json")); sm=o["stage_ms_median"]
print("$C ms/step %.3f" % o["ms_per_step"], {k: round(v,3) for k,v in sm.items()})
print("   roofline", o["roofline"]["achieved"], o["roofline"]["frac"], o.get("hot_path",{}) if isinstance(o.get("hot_path"),dict) else "")
P
done
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r04_call10_pytest.log 2>&1; grep -v "^W2026\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" gpurun_out/r04_call10_pytest.log | tail -25
