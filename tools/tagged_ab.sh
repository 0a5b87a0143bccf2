#!/bin/bash
# A/B on the GPU box: k_mg_coarse's tagged hand-off (default, fp32) against the ready-flag form (LFA_MG_NO_TAGGED=1): PCG loop per config.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
L="--no-cpu-baseline --no-hot-path --no-mic0-record --no-kernel-timing"
for C in C2 C3 C4; do
  for V in top notop flags top notop flags; do
    unset LFA_MG_NO_TAGGED LFA_MG_NO_TOP
    if [ $V = flags ]; then export LFA_MG_NO_TAGGED=1; fi
    if [ $V = notop ]; then export LFA_MG_NO_TOP=1; fi
    python3 bench.py --config $C --steps 30 --warmup 20 $L 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); sm=d['stage_ms_median']
print('$C $V step %.3f ms  pcg_loop %.3f  iteration %.4f  it/step %.2f' % (d['ms_per_step'], sm['pcg_loop'], sm['pcg_iteration_mean'], d['pcg']['iterations_per_step']))"
  done
done
