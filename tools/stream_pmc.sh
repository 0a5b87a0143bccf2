#!/bin/bash
# PMC passes of the streaming particle kernels at C4 (advect + collide + count, tile scatter, fine index, finalize): VALU load and HBM traffic
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-r03}
{
echo "# rocprofv3 --pmc <group> --kernel-trace -- python3 tools/fullstep_stages.py C4 12 2; medians per dispatch (tools/pmc_summary.py)."
bash tools/pmc_kernel.sh stream "k_advect_collide_count\|k_tile_scatter\|k_build_fine_index\|k_p2g_finalize" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" -- python3 tools/fullstep_stages.py C4 12 2
} > gpurun_out/${TAG}_stream_pmc.txt 2>&1
cut -c1-150 gpurun_out/${TAG}_stream_pmc.txt
