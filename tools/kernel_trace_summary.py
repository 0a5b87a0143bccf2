"""Per-kernel summary of a rocprofv3 --kernel-trace results.db with MEDIAN and p95 beside the mean (the first launch of a
kernel is often not a steady-state one, and a mean hides it): kernel, calls, total_us, mean_us, median_us, p95_us, max_us, percent.
usage: kernel_trace_summary.py results.db [skip_first_n_dispatches_per_kernel]"""
import re
import sqlite3
import statistics
import sys

db = sqlite3.connect(sys.argv[1])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
views = [r[0] for r in db.execute("select name from sqlite_master where type in ('view', 'table')")]
src = "kernels" if "kernels" in views else None
if src is None:
    sys.exit("no `kernels` view in this results.db; objects: " + ", ".join(views))
cols = [r[1] for r in db.execute(f"pragma table_info({src})")]
if "duration" in cols:
    q = f"select name, duration from {src} order by start"
else:
    q = f"select name, end - start from {src} order by start"
groups = {}
for name, dur in db.execute(q):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z_0-9]+(<[^>]*>)?)", n)
    groups.setdefault(m.group(1) if m else n[:40], []).append(dur / 1e3)
total = sum(sum(v) for v in groups.values())
print("kernel,calls,total_us,mean_us,median_us,p95_us,max_us,percent")
for k, v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
    w = v[skip:] if len(v) > skip else v
    s = sorted(w)
    print(f'"{k}",{len(v)},{sum(v):.1f},{statistics.mean(w):.3f},{statistics.median(w):.3f},'
          f'{s[min(len(s) - 1, int(0.95 * len(s)))]:.3f},{s[-1]:.3f},{100.0 * sum(v) / total:.2f}')
