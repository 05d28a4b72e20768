#!/bin/bash
# round 4, call aq: what holds k_slab_combine (R-MAT 24)?  LDS conflicts, the dispatcher's resource stalls, instruction classes
mkdir -p gpurun_out
{
export KFILTER="k_slab_combine<double, 16"
PMC="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" bash scripts/gpu_pmc1.sh lds --no-sub-configs
PMC="SPI_RA_LDS_CU_FULL_CSN SPI_RA_TGLIM_CU_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN" bash scripts/gpu_pmc1.sh spi --no-sub-configs
PMC="SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN SPI_CSN_BUSY SPI_CSN_WAVE" bash scripts/gpu_pmc1.sh spi2 --no-sub-configs
PMC="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD" bash scripts/gpu_pmc1.sh cls --no-sub-configs
PMC="SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_WAVES SQ_IFETCH" bash scripts/gpu_pmc1.sh lvl --no-sub-configs
PMC="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" bash scripts/gpu_pmc1.sh sq --no-sub-configs
PMC="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU" bash scripts/gpu_pmc1.sh ins --no-sub-configs
rm -rf gpurun_out/pmc1_*
} 2>&1 | grep -v "^$" > gpurun_out/r04_pmc_combine_raw.txt
cat gpurun_out/r04_pmc_combine_raw.txt
