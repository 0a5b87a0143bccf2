#!/bin/bash
# PMC counters of the finest-level residual kernel (the dispatches of k_mg_residual_restrict longer than 15 us) at C4.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
OUT=gpurun_out/r05_rr0_pmc.txt; : > $OUT
i=0
for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  rm -rf /tmp/rr_$i
  rocprofv3 --pmc $G --kernel-trace -d /tmp/rr_$i -- python3 tools/fullstep_stages.py C4 20 2 > /tmp/rr_$i.log 2>&1
  DB=$(find /tmp/rr_$i -name "*results.db" | head -1)
  for K in k_mg_residual_restrict "k_mg_prolong_postsmooth<float, true" k_mg_axpy_presmooth k_pcg_a; do
  python3 - "$DB" "$K" >> $OUT <<'P'
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
  i=$((i+1))
done
cat $OUT
