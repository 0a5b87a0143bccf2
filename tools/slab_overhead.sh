#!/bin/bash
# The z-slab code path with ONE rank (RCCL communicator, halo calls, migration) against the single-domain step (GPU box).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
L="--no-cpu-baseline --no-hot-path --no-mic0-record --no-kernel-timing --no-serial-stages"
for C in ${CONFIGS:-C3 C4}; do
  python3 bench.py --config $C --steps 20 --warmup 20 $L > gpurun_out/slab1_${C}_single.json 2>/tmp/s1.err || tail -3 /tmp/s1.err
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-slabs --config $C --steps 20 --warmup 20 $L > gpurun_out/slab1_${C}_slabs.json 2>/tmp/s2.err || tail -5 /tmp/s2.err
  python3 - $C <<'P'
import json, sys
c = sys.argv[1]
last = lambda f: json.loads([l for l in open(f) if l.startswith("{")][-1])  # (librccl prints a banner to stdout first)
a = last(f"gpurun_out/slab1_{c}_single.json"); b = last(f"gpurun_out/slab1_{c}_slabs.json")
print(c, "single %.3f ms" % a["ms_per_step"], "one-rank slabs %.3f ms" % b["ms_per_step"], "ratio %.3f" % (b["ms_per_step"] / a["ms_per_step"]),
      "it", a["pcg"]["iterations_per_step"], b["pcg"]["iterations_per_step"], b["pcg"]["solver_stats_last_solve"])
P
done
