cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for V in single two single two; do
  if [ $V = two ]; then export LFA_DIST_TWO_REDUCTIONS=1; else unset LFA_DIST_TWO_REDUCTIONS; fi
  echo "== $V reduction(s)"
  bash tools/slab_overhead.sh 2>&1 | grep "^C[0-9]"
done
