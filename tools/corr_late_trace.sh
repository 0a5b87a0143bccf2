#!/bin/bash
# Kernel trace of the position correction late in the C3 run (last 20 dispatches of every kernel whose name holds "correct" / "fine").
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
W=${1:-C3late}
rm -rf /tmp/kt && rocprofv3 --kernel-trace -d /tmp/kt -- python3 tools/correct_ab.py $W > /tmp/kt.out 2> /tmp/kt.log
python3 - "$(find /tmp/kt -name '*results.db' | head -1)" <<'P'
import sqlite3, sys, statistics, re
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
q = "select name, duration from kernels order by start" if "duration" in cols else "select name, end - start from kernels order by start"
g = {}
for name, dur in db.execute(q):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z_0-9]+(<[^>]*>)?)", n)
    g.setdefault(m.group(1) if m else n[:40], []).append(dur / 1e3)
for k, v in g.items():
    if "correct" in k or "fine" in k:
        w = v[-20:]
        print(f"{k}: last 20 dispatches median {statistics.median(w):.1f} us, min {min(w):.1f}, max {max(w):.1f}")
P
tail -1 /tmp/kt.out
