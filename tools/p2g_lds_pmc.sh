cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
CTRS="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS"
{
echo "# LDS PMC pass of the P2G scatter kernel k_p2g_binned<true, false> (APIC) at C4 on the moving dam (steps 20-22 of the dam break),"
echo "# rocprofv3 --pmc $CTRS --kernel-trace -- python3 tools/fullstep_stages.py C4 20 3; medians per dispatch (tools/pmc_summary.py)."
echo "# BEFORE = LFA_P2G_NO_ROT=1 (every lane visits the eight nodes of a component in the same order: round 1's kernel)"
LFA_P2G_NO_ROT=1 bash tools/pmc_kernel.sh p2g_before "k_p2g_binned" "$CTRS" -- python3 tools/fullstep_stages.py C4 20 3
echo "# AFTER = default (node order rotated by the lane number)"
bash tools/pmc_kernel.sh p2g_after "k_p2g_binned" "$CTRS" -- python3 tools/fullstep_stages.py C4 20 3
} > gpurun_out/r02_p2g_lds_pmc.txt 2>&1
cat gpurun_out/r02_p2g_lds_pmc.txt
