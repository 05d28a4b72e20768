#!/bin/bash
# round 4, call bq: TIMING ONLY -- the tile's cold lanes issued as ceil(cold / 64) FULL gather instructions instead of 8 quarter-full ones
# (what a cross-lane compaction would issue; its own cost is not in here)
cd scripts/probes
for r in 1 2; do
echo "## 8 gather instructions, 28 % of the lanes cold"; timeout 120 ./lds_dma_streams_wv 268435456 28 | grep "registers + gathers\|no streams"
echo "## compacted gather instructions"; timeout 120 ./lds_dma_streams_cg 268435456 28 | grep "registers + gathers\|no streams"
done
