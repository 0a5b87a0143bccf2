#!/bin/bash
# PMC pass of the G2P kernel at C4 on the moving dam (stages back to back): what bounds it?
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-r03}
{
echo "# rocprofv3 --pmc <group> --kernel-trace -- python3 tools/fullstep_stages.py C4 20 2; medians per dispatch (tools/pmc_summary.py)."
bash tools/pmc_kernel.sh g2p "k_g2p<" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" -- python3 tools/fullstep_stages.py C4 20 2
} > gpurun_out/${TAG}_g2p_pmc.txt 2>&1
cat gpurun_out/${TAG}_g2p_pmc.txt | cut -c1-150
