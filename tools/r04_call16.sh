cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_slabs.py tests/test_next_rows.py -m gpu -x -q > gpurun_out/r04_call16_pytest.log 2>&1; grep -v "^W2026\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|^RCCL" gpurun_out/r04_call16_pytest.log | tail -25
bash tools/r04_call15.sh C3
bash tools/slab_overhead.sh 2>&1 | grep "^C[0-9]"
