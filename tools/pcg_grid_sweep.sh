#!/bin/bash
# usage: tools/pcg_grid_sweep.sh [CFG] GA:GB [GA:GB ...]   -- times the fused PCG kernels for several launch widths
CFG=${1:-C4}; shift
[ $# -eq 0 ] && set -- 2048:2048 1024:2048 1024:1024 1280:1280 768:2048
for pair in "$@"; do
  echo "== GA=${pair%%:*} GB=${pair##*:}"
  LFA_PCG_GA=${pair%%:*} LFA_PCG_GB=${pair##*:} python tools/pcg_kernel_probe.py $CFG 2>&1 | grep -v amdgpu.ids | head -3
done
