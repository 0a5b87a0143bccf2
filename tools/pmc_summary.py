"""Per-kernel mean of a rocprofv3 --pmc counter (FETCH_SIZE / WRITE_SIZE, in KiB per dispatch) from a results.db."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, counter_name, count(*), avg(counter_value), avg(duration) from pmc_events "
                  "group by name, counter_name order by sum(counter_value) desc").fetchall()
print("kernel,counter,dispatches,mean_KiB_per_dispatch,mean_duration_ns")
for name, ctr, n, val, dur in rows:
    nm = name.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z_0-9]+(<[^>]*>)?)", nm)
    print(f'"{m.group(1) if m else nm[:40]}",{ctr},{n},{val:.1f},{dur:.0f}')
