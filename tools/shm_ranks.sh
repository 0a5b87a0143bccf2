#!/bin/bash
# tools/shm_ranks.sh -- bench.py --gpus N as the driver launches it, on a box with ONE GPU: the ranks share the device and the slab
# messages go through the shared-memory transport (lfa_dist_init_shm). A FUNCTIONAL record of the N-process path (rendezvous, slab
# bounds, migration, global CFL, max-over-ranks timing); the ranks time-share one GPU, so the rates say nothing about scaling.
mkdir -p gpurun_out
for n in 2 4; do
  for mode in "" "--strong"; do
    tag="n${n}${mode:+_strong}"
    timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) \
      bench.py --gpus $n --steps 10 --warmup 3 --no-serial-stages $mode > gpurun_out/shm_$tag.log 2>&1
    grep '^{' gpurun_out/shm_$tag.log | tail -1 > gpurun_out/r03_shm_$tag.json
    python - <<PY
import json
try:
    o = json.load(open("gpurun_out/r03_shm_$tag.json"))
    print("$tag", o["value"], o["ms_per_step"], o["transport"], o["config"]["parallelism"], o.get("solver_stats_last_solve"))
except Exception as e:
    print("$tag failed", e)
    print(open("gpurun_out/shm_$tag.log").read()[-1500:])
PY
  done
done
