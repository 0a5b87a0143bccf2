cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
L="--no-cpu-baseline --no-hot-path --no-mic0-record --no-kernel-timing"
C=${1:-C3}
python3 bench.py --config $C --steps 20 --warmup 20 $L > /tmp/a.json 2>/tmp/s1.err || tail -3 /tmp/s1.err
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-slabs --config $C --steps 20 --warmup 20 $L > /tmp/b.json 2>/tmp/s2.err || tail -5 /tmp/s2.err
python3 - <<'P'
import json
last = lambda f: json.loads([l for l in open(f) if l.startswith("{")][-1])
a = last("/tmp/a.json"); b = last("/tmp/b.json")
print("ms/step", a["ms_per_step"], b["ms_per_step"])
for k in a["stage_ms_median"]:
    print("%-24s %8.3f %8.3f  %+.3f" % (k, a["stage_ms_median"][k], b["stage_ms_median"].get(k, float('nan')), b["stage_ms_median"].get(k, float('nan')) - a["stage_ms_median"][k]))
P
