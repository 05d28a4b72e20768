#!/bin/bash
# round 4, call t: PMC passes on the final k_spmv_range and k_slab_combine (what holds the L1: request latency, TLB, FIFOs, L2)
mkdir -p gpurun_out
{
for K in k_spmv_range "k_slab_combine<double, 16"; do
export KFILTER="$K"
echo "#### $K"
PMC="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum" bash scripts/gpu_pmc1.sh lat --no-sub-configs --no-side-figures
PMC="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_WRITE_REQ_sum" bash scripts/gpu_pmc1.sh tlb --no-sub-configs --no-side-figures
PMC="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" bash scripts/gpu_pmc1.sh l2 --no-sub-configs --no-side-figures
PMC="TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TOTAL_ACCESSES_sum" bash scripts/gpu_pmc1.sh l1 --no-sub-configs --no-side-figures
PMC="GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" bash scripts/gpu_pmc1.sh ta --no-sub-configs --no-side-figures
PMC="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" bash scripts/gpu_pmc1.sh sq --no-sub-configs --no-side-figures
rm -rf gpurun_out/pmc1_*
done
} 2>&1 | grep -v "^$" > gpurun_out/r04_pmc_raw.txt
cat gpurun_out/r04_pmc_raw.txt
