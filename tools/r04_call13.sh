cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
L="--no-cpu-baseline --no-hot-path --no-mic0-record --no-kernel-timing"
for CAP in 2048 1024 1536 768 2048 1024; do
  for C in C4; do
  LFA_PCG_GRID_CAP=$CAP python3 bench.py --config $C --steps 20 --warmup 20 $L 2>/dev/null | grep "^{" > /tmp/b.json
  python3 - <<P
import json
o=json.load(open("/tmp/b.json")); sm=o["stage_ms_median"]
print("cap $CAP $C ms/step %.3f" % o["ms_per_step"], "pcg_loop %.3f iter %.4f its %.2f" % (sm["pcg_loop"], sm["pcg_iteration_mean"], o["pcg"]["iterations_per_step"]))
P
  done
done
# PMC of the finest-level kernels (what binds them: waiting or issuing)
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD"
G2="SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
bash tools/pmc_kernel.sh r04pcg "k_pcg_a\|k_mg_axpy_presmooth\|k_mg_residual_restrict\|k_mg_prolong_postsmooth\|k_mg_coarse" "$G1" "$G2" -- python3 tools/fullstep_stages.py C4 20 2 > gpurun_out/r04_pcg_pmc.txt 2>&1
head -60 gpurun_out/r04_pcg_pmc.txt
