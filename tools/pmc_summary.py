"""Per-kernel value of a rocprofv3 --pmc counter (FETCH_SIZE / WRITE_SIZE: KiB per dispatch; SQ_*: raw counts) from a results.db.

The value is the MEDIAN over the kernel's dispatches (each dispatch: sum over the counter's instances): the first launch of
a kernel is often not a steady-state one (the first binning after seeding applies a rank permutation and writes 9x the
bytes of every later one; a mean over 8 launches reported 1.41x write amplification that no steady-state launch has)."""
import re
import sqlite3
import statistics
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, counter_name, dispatch_id, sum(counter_value), avg(duration) from pmc_events "
                  "group by name, counter_name, dispatch_id").fetchall()
groups = {}
for name, ctr, _, val, dur in rows:
    nm = name.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z_0-9]+(<[^>]*>)?)", nm)
    groups.setdefault((m.group(1) if m else nm[:40], ctr), []).append((val, dur))
print("kernel,counter,dispatches,median_value_per_dispatch,median_duration_ns,max_value_per_dispatch")
for (nm, ctr), v in sorted(groups.items(), key=lambda kv: -statistics.median(x[0] for x in kv[1]) * len(kv[1])):
    vals, durs = [x[0] for x in v], [x[1] for x in v]
    print(f'"{nm}",{ctr},{len(v)},{statistics.median(vals):.1f},{statistics.median(durs):.0f},{max(vals):.1f}')
