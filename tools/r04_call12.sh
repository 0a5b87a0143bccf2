cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_next_rows.py tests/test_configs.py -m gpu -x -q -k "correction or dam or C3 or C2" 2>&1 | grep -v "^W2026\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|^RCCL" | tail -15
L="--no-cpu-baseline --no-mic0-record --no-hot-path"
for V in big nobig; do
  if [ $V = nobig ]; then export LFA_CORR_NO_BIG=1; else unset LFA_CORR_NO_BIG; fi
  python3 bench.py --config C3 --steps 30 --warmup 20 --late 550 --late-steps 20 $L 2>/dev/null | grep "^{" > /tmp/b.json
  python3 - <<P
import json
o=json.load(open("/tmp/b.json")); sm=o["stage_ms_median"]; lp=o["late_phase"]; ls=lp["stage_ms_median"]
print("$V C3 ms/step %.3f" % o["ms_per_step"], "late %.3f ratio %.3f" % (lp["ms_per_step"], lp["ratio_to_timed_region"]), lp["correction_fallback_half_tiles"])
print("   late stages", {k: round(v,3) for k,v in ls.items()})
P
done
unset LFA_CORR_NO_BIG
python3 bench.py --config C4 --steps 30 --warmup 20 $L 2>/dev/null | grep "^{" > /tmp/b.json
python3 - <<P
import json
o=json.load(open("/tmp/b.json")); sm=o["stage_ms_median"]
print("C4 ms/step %.3f" % o["ms_per_step"], {k: round(v,3) for k,v in sm.items()})
P
