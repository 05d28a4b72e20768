#!/bin/bash
# Round 6: the PMC passes of round 4's call r4t (+ four groups of r4aq on the combine) on the FINAL round-6 sources -- what holds
# k_spmv_range and k_slab_combine on R-MAT 24 (VERDICT r05 Missing 4: the counter evidence for the headline's dominant kernel was a
# round old).  One counter group per pass, counters + kernel trace only.  -> gpurun_out/r06_pmc_raw.txt, r06_pmc_combine_raw.txt;
# then here: python scripts/experiments/pmc_range_report.py gpurun_out/r06_pmc_raw.txt gpurun_out/r06_pmc_combine_raw.txt r06 > profiles/r06_pmc_range.txt
mkdir -p gpurun_out
B="--no-sub-configs --no-side-figures"
{
for K in k_spmv_range "k_slab_combine<double, 16"; do
export KFILTER="$K"
echo "#### $K"
PMC="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum" bash scripts/gpu_pmc1.sh lat $B
PMC="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_WRITE_REQ_sum" bash scripts/gpu_pmc1.sh tlb $B
PMC="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" bash scripts/gpu_pmc1.sh l2 $B
PMC="TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TOTAL_ACCESSES_sum" bash scripts/gpu_pmc1.sh l1 $B
PMC="GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" bash scripts/gpu_pmc1.sh ta $B
PMC="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" bash scripts/gpu_pmc1.sh sq $B
rm -rf gpurun_out/pmc1_*
done
} 2>&1 | grep -v "^$" > gpurun_out/r06_pmc_raw.txt
{
export KFILTER="k_slab_combine<double, 16"
PMC="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" bash scripts/gpu_pmc1.sh lds $B
PMC="SPI_RA_LDS_CU_FULL_CSN SPI_RA_TGLIM_CU_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN" bash scripts/gpu_pmc1.sh spi $B
PMC="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU" bash scripts/gpu_pmc1.sh ins $B
rm -rf gpurun_out/pmc1_*
} 2>&1 | grep -v "^$" > gpurun_out/r06_pmc_combine_raw.txt
cat gpurun_out/r06_pmc_raw.txt gpurun_out/r06_pmc_combine_raw.txt
