"""Per-dispatch values of a rocprofv3 --pmc counter for one kernel (summed over the counter's instances), from a results.db."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
rows = db.execute("select dispatch_id, counter_name, sum(counter_value), avg(duration) from pmc_events where name like ? "
                  "group by dispatch_id, counter_name order by dispatch_id", (f"%{pat}%",)).fetchall()
for r in rows:
    print(r)
