cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r04_final_pytest.log 2>&1; grep -v "^W2026\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|^RCCL" gpurun_out/r04_final_pytest.log | tail -6
python3 bench.py --late 300 --late-steps 20 > gpurun_out/r04_c4_bench.json 2>/tmp/b.err || tail -3 /tmp/b.err
python3 -c "
import json; o=json.load(open('gpurun_out/r04_c4_bench.json')); print(o['ms_per_step'], o['value'], o['roofline']['frac'], o['roofline']['traffic_source'], o['roofline_groups']['p2g_plus_pcg']['frac'])"
