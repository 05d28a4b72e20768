#!/bin/bash
# round 4, call bj: one sequential stream per wavefront (6-KB tile records: column words, then values) instead of two arrays -- does
# the HBM side care how many concurrent streams the 2 048 wavefronts read?
cd scripts/probes
echo "## two arrays"; timeout 120 ./lds_dma_streams_w0 268435456 28,0 | grep -v "LDS-DMA\|consumers"
echo "## one array of tile records"; timeout 120 ./lds_dma_streams_il 268435456 28,0
echo "## two arrays"; timeout 120 ./lds_dma_streams_w0 268435456 28 | grep -v "LDS-DMA\|consumers"
echo "## one array of tile records"; timeout 120 ./lds_dma_streams_il 268435456 28
