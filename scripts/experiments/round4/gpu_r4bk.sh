#!/bin/bash
# round 4, call bk: how much does the size of the region the cold gathers fall into matter (per XCD: 3 600 KB = the product's, down to L1-sized)?
cd scripts/probes
for kb in 3600 1024 256 64 16; do timeout 120 ./lds_dma_streams_w0 268435456 28 $kb | grep "^##\|registers + gathers\|no streams"; done
