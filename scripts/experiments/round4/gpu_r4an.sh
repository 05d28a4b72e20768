#!/bin/bash
# round 4, call an: the stream / gather probe again with its modes repaired (round 4's first version computed the column codes of the
# "no streams" and "streams only" modes with 64-bit hashes -- 388 us of arithmetic -- and the compiler dropped the unused stream loads),
# plus wavefront specialisation: producer wavefronts stream into LDS buffers, consumer wavefronts gather and compute
cd scripts/probes
timeout 120 ./lds_dma_streams 268435456 28,0
timeout 120 ./lds_dma_streams_w0 268435456 28,0
