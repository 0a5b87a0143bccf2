#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / duration of the finest-level solver kernels for the library named by LFA_LIB_PATH (or the default one)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for G in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/rrab
  rocprofv3 --pmc $G --kernel-trace -d /tmp/rrab -- python3 tools/fullstep_stages.py C4 20 2 > /tmp/rrab.log 2>&1
  DB=$(find /tmp/rrab -name "*results.db" | head -1)
  for K in k_mg_residual_restrict "k_mg_prolong_postsmooth<float, true" k_mg_axpy_presmooth; do
  python3 - "$DB" "$K" <<'P'
import sqlite3, sys, statistics
db = sqlite3.connect(sys.argv[1]); pat = sys.argv[2]
rows = db.execute("select dispatch_id, counter_name, sum(counter_value), avg(duration) from pmc_events where name like ? group by dispatch_id, counter_name", (f"%{pat}%",)).fetchall()
by = {}
for d, c, v, dur in rows:
    if dur > 15000: by.setdefault(c, []).append((v, dur))
for c, l in by.items():
    print(pat, c, "dispatches", len(l), "median", statistics.median(x[0] for x in l), "median_ns", statistics.median(x[1] for x in l))
P
  done
done
