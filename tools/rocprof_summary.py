"""Turns a rocprofv3 --kernel-trace --stats results.db into the per-kernel summary committed under profiles/."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
print("kernel,calls,total_us,avg_us,percent")
for name, calls, tot, avg, pct in rows:
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z_0-9]+(<[^>]*>)?)", n)
    print(f'"{m.group(1) if m else n[:40]}",{calls},{tot:.1f},{avg:.3f},{pct:.2f}')
