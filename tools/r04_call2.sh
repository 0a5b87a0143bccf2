set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_pytest_gpu.txt 2>&1; tail -15 gpurun_out/r04_pytest_gpu.txt
timeout 600 python bench.py > gpurun_out/r04_cells_bench.json 2> gpurun_out/r04_cells_bench.err
LFA_P2G_CELLS=0 timeout 600 python bench.py > gpurun_out/r04_sorted_oldp2g_bench.json 2>/dev/null
LFA_BIN_CELLSORT=0 timeout 600 python bench.py > gpurun_out/r04_unsorted_bench.json 2>/dev/null
python - <<'PY'
import json
for f in ("r04_cells_bench","r04_sorted_oldp2g_bench","r04_unsorted_bench"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], {k:round(v,3) for k,v in d["stage_ms_median"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
