cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_handles.py -m gpu -x -q 2>&1 | tail -15
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 600 python bench.py > gpurun_out/r04_b_bench.json 2> gpurun_out/r04_b_bench.err; tail -2 gpurun_out/r04_b_bench.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_b_bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["scaling"], {k:round(v,3) for k,v in d["stage_ms_median"].items()})
PY
