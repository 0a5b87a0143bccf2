#!/bin/bash
# tools/late_fallback_prof.sh -- kernel trace of the C3 late window (step 550: ~0.7 % of the half tiles take the correction's fallback kernel)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf /tmp/lf_prof
rocprofv3 --kernel-trace --stats -d /tmp/lf_prof -o lf -- python3 bench.py --config C3 --steps 5 --warmup 5 --late 550 --late-steps 20 --no-cpu-baseline --no-hot-path --no-mic0-record --no-serial-stages > /tmp/lf.log 2>&1
f=$(find /tmp/lf_prof -name "*kernel_stats.csv" | head -1)
head -1 $f; grep -E "k_correct|k_build_fine" $f
python3 - <<P
import csv, glob
f = glob.glob("/tmp/lf_prof/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_correct_fl" in r["Kernel_Name"] or "k_correct_collide" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("fallback kernel launches", len(d), "last 20 (us):", [round(x) for x in d[-20:]])
P
