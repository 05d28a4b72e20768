#!/bin/bash
# round 5: what holds the plain-path tile kernels on the SuiteSparse-shaped stand-ins?  rocprofv3 --pmc passes (one counter group per
# pass, counters + kernel trace only) of `bench.py --workload W --slabs 0 --tile-walk off|force`, averaged over the COLD-protocol
# launches (the last 40 dispatches of the kernel in a bench run with --steps 20).  usage: pmc_plain.sh "<workloads>" "<variants>"
mkdir -p gpurun_out
WL=${1:-"nd24k webbase"}
VAR=${2:-"off force"}
{
for W in $WL; do
for V in $VAR; do
export KFILTER="k_spmv"
export KLAST=40
echo "#### $W tile-walk=$V (cold launches)"
A="--workload $W --slabs 0 --tile-walk $V --no-sub-configs --no-side-figures"
[ "$W" = nd24k ] && A="$A --sigma 16"   # (the walking kernel exists for sigma <= 16: both kernels at 16, the fp32 auto rule's choice)
PMC="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum" bash scripts/gpu_pmc1.sh lat $A
PMC="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" bash scripts/gpu_pmc1.sh l2 $A
PMC="TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TOTAL_ACCESSES_sum" bash scripts/gpu_pmc1.sh l1 $A
PMC="GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" bash scripts/gpu_pmc1.sh ta $A
PMC="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" bash scripts/gpu_pmc1.sh sq $A
PMC="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" bash scripts/gpu_pmc1.sh inst $A
PMC="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES" bash scripts/gpu_pmc1.sh act $A
PMC="SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_WAVES SQ_INSTS_VMEM_WR" bash scripts/gpu_pmc1.sh misc $A
rm -rf gpurun_out/pmc1_*
done
done
} 2>&1 | grep -v "^$" > gpurun_out/r05_pmc_plain_raw.txt
cat gpurun_out/r05_pmc_plain_raw.txt
