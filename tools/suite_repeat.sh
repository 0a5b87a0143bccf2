#!/bin/bash
# tools/suite_repeat.sh N -- the GPU suite N times on one box; the log of every run that fails is kept (flaky-test hunt).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for i in $(seq 1 ${1:-5}); do
  timeout 850 python -m pytest tests/ -q -m gpu -rf -x > /tmp/suite.log 2>&1
  grep -E "passed|failed" /tmp/suite.log | tail -1
  if grep -q "^FAILED" /tmp/suite.log; then cp /tmp/suite.log gpurun_out/suite_fail_$i.log; grep "^FAILED" /tmp/suite.log; fi
done
