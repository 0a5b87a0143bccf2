#!/bin/bash
# A/B of variant libraries on the GPU box: tools/lib_ab.sh "C3 C4" default libfluid_amd/variants/x.so ...   (each twice, interleaved)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
CFGS=$1; shift
L="--no-cpu-baseline --no-hot-path --no-mic0-record --no-kernel-timing"
for C in $CFGS; do
  for R in 1 2; do
    for V in "$@"; do
      unset LFA_LIB_PATH
      if [ $V != default ]; then export LFA_LIB_PATH=$V; fi
      python3 bench.py --config $C --steps 30 --warmup 20 $L 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); sm=d['stage_ms_median']
print('$C $V step %.3f ms  pcg_loop %.3f  iteration %.4f  it/step %.2f  p2g %.3f  g2p %.3f  advect %.3f  bin %.3f' % (d['ms_per_step'], sm['pcg_loop'], sm['pcg_iteration_mean'], d['pcg']['iterations_per_step'], sm['p2g'], sm['g2p'], sm['advect_collide'], sm['bin']))"
    done
  done
done
