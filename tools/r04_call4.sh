cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
rocprofv3 -L 2>/dev/null | grep -o "Name:[[:space:]]*[A-Za-z0-9_]*" | sed 's/Name:[[:space:]]*//' | sort -u > gpurun_out/r04_counter_names.txt
wc -l gpurun_out/r04_counter_names.txt
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD"
G2="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"
G3="FETCH_SIZE"
G4="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
bash tools/pmc_kernel.sh r04cells "k_p2g_cells\|k_cell_sort\|k_tile_scatter_idx" "$G1" "$G2" "$G3" "$G4" -- python3 tools/fullstep_stages.py C4 20 2 > gpurun_out/r04_cells_pmc.txt 2>&1
LFA_BIN_CELLSORT=0 bash tools/pmc_kernel.sh r04old "k_p2g_binned\|k_tile_scatter" "$G1" "$G2" "$G3" "$G4" -- python3 tools/fullstep_stages.py C4 20 2 > gpurun_out/r04_old_pmc.txt 2>&1
cat gpurun_out/r04_cells_pmc.txt gpurun_out/r04_old_pmc.txt
tail -3 /tmp/pmc_r04cells_1.log
