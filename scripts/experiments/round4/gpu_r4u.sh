#!/bin/bash
# round 4, call u: PMC passes on k_slab_combine<double, 16> of R-MAT 24 only
mkdir -p gpurun_out
export KFILTER="k_slab_combine<double, 16"
{
echo "#### $KFILTER"
PMC="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_READ_sum" bash scripts/gpu_pmc1.sh lat --no-sub-configs
PMC="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" bash scripts/gpu_pmc1.sh l2 --no-sub-configs
PMC="GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" bash scripts/gpu_pmc1.sh ta --no-sub-configs
PMC="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" bash scripts/gpu_pmc1.sh sq --no-sub-configs
rm -rf gpurun_out/pmc1_*
} 2>&1 | grep -v "^$" > gpurun_out/r04_pmc_combine_raw.txt
cat gpurun_out/r04_pmc_combine_raw.txt
