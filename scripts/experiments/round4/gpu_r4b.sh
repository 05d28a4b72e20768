#!/bin/bash
# round 4, call b: what holds the L1 (TCP) of k_spmv_range -- request latency, TLB, FIFOs (one PMC pass each)
export KFILTER=k_spmv_range
PMC="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum" bash scripts/gpu_pmc1.sh lat
PMC="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" bash scripts/gpu_pmc1.sh tlb
PMC="TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum TCP_TD_TCP_STALL_CYCLES_sum" bash scripts/gpu_pmc1.sh fifo
PMC="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" bash scripts/gpu_pmc1.sh l2
PMC="TCP_TOTAL_CACHE_ACCESSES_sum TCP_CACHE_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" bash scripts/gpu_pmc1.sh l1
PMC="GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" bash scripts/gpu_pmc1.sh ta
