#!/bin/bash
# PMC pass of the PCG loop's kernels at C4 on the moving dam (stages back to back): instruction mix and waits.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-r03}
{
echo "# rocprofv3 --pmc <group> --kernel-trace -- python3 tools/fullstep_stages.py C4 20 2; medians per dispatch (tools/pmc_summary.py)."
echo "# SQ_* counters are summed over the device; SQ_BUSY_CYCLES / kernel cycles = 32 shader engines."
bash tools/pmc_kernel.sh pcg "k_mg_axpy_presmooth\|k_mg_prolong_postsmooth<float, true>\|k_pcg_a<float, false\|k_mg_residual_restrict<\|k_mg_coarse" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU" -- python3 tools/fullstep_stages.py C4 20 2
} > gpurun_out/${TAG}_pcg_pmc.txt 2>&1
cat gpurun_out/${TAG}_pcg_pmc.txt | cut -c1-170
