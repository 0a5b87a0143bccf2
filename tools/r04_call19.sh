cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_slabs.py -m gpu -x -q -k "shared_memory or bench" > gpurun_out/r04_call19_pytest.log 2>&1; grep -v "^W2026\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|^RCCL" gpurun_out/r04_call19_pytest.log | tail -25
