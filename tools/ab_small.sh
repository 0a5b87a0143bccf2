#!/bin/bash
# A/B on the GPU box: the whole PCG solve in one launch (k_pcg_small, default for small systems) against the multi-launch loop.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
L="--no-cpu-baseline --no-hot-path --no-mic0-record --no-kernel-timing"
for C in ${CONFIGS:-C2}; do
  for V in small multi; do
    if [ $V = multi ]; then export LFA_PCG_NO_SMALL=1; else unset LFA_PCG_NO_SMALL; fi
    python3 bench.py --config $C --steps 30 --warmup 20 $L 2>/dev/null | python3 -c "
import json,sys
b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); sm=b['stage_ms_median']
print('$C $V', 'ms/step %.3f' % b['ms_per_step'], 'serial %.3f' % b.get('ms_per_step_serial_stages',0), 'it', b['pcg']['iterations_per_step'], 'pcg_iter_ms %.4f' % sm['pcg_iteration_mean'], 'pcg_loop %.3f' % sm['pcg_loop'], b['pcg']['solver_stats_last_solve'])"
  done
done
