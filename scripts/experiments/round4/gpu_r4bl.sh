#!/bin/bash
# round 4, call bl: the value stream as four 16-byte loads per lane instead of eight 8-byte ones (half the vector-memory instructions, same lines)
cd scripts/probes
for r in 1 2; do
echo "## 8 x 8 bytes"; timeout 120 ./lds_dma_streams_w0 268435456 28 | grep "registers + gathers\|streams only"
echo "## 4 x 16 bytes"; timeout 120 ./lds_dma_streams_wv 268435456 28 | grep "registers + gathers\|streams only"
done
