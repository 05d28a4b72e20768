#!/usr/bin/env python3
"""profiles/rNN_pmc_range.txt from the raw counter averages of scripts/experiments/round4/gpu_r4t.sh (range kernel + combine) and
gpu_r4aq.sh (more groups on the combine): the raw blocks plus a per-CU reading.

    python scripts/experiments/pmc_range_report.py gpurun_out/r04_pmc_raw.txt gpurun_out/r04_pmc_combine_raw.txt > profiles/r04_pmc_range.txt
"""
import re
import sys

raw = open(sys.argv[1]).read()
extra = open(sys.argv[2]).read()
ROUND = sys.argv[3] if len(sys.argv) > 3 else "r04"
if ROUND == "r04":
    SOURCES = ("FINAL round-4 sources: 8 wavefronts per CU, 16 384-slot table, permuted x,\n## bit flags in the column codes -- no "
               "descriptor load --, DPP prefix for y_offset, values in 16-byte pieces")
    HOW = "scripts/experiments/round4/gpu_r4t.sh"
    EARLIER = """{EARLIER}"""
else:
    SOURCES = (f"FINAL {ROUND} sources: round 4's kernel + round 5's dealing of 8.5 % more tiles to the first four\n## wavefronts of a "
               "workgroup and the combine's rotated row groups; round 6 changed nothing in it but the CSR tail's x (read from the permuted copy)")
    HOW = "scripts/experiments/round6/pmc_range.sh"
    EARLIER = ("## Round 4's figures on the same counters (profiles/r04_pmc_range.txt): 111.5 M requests, 478 clocks, 94 in flight, queue full 68 %, "
               "TA busy 89 %.\n")
rng = raw[raw.index('#### k_spmv_range'):raw.index('#### k_slab_combine')]
cmb = raw[raw.index('#### k_slab_combine'):]


def g(block, name):
    return float(re.search(name + r"\s+n=\s*\d+ avg=\s*([0-9.]+)", block).group(1))


n = int(re.search(r"n=\s*(\d+)", rng).group(1))
req, lat, gui = g(rng, 'TCP_TCC_READ_REQ_sum'), g(rng, 'TCP_TCC_READ_REQ_LATENCY_sum'), g(rng, 'GRBM_GUI_ACTIVE') / 8
pend, ta, addr = g(rng, 'TCP_PENDING_STALL_CYCLES_sum'), g(rng, 'TA_TA_BUSY_sum'), g(rng, 'TCP_TCP_TA_ADDR_STALL_CYCLES_sum')
hit, miss, ea = g(rng, 'TCC_HIT_sum'), g(rng, 'TCC_MISS_sum'), g(rng, 'TCC_EA0_RDREQ_sum')
tlbm, tlbr = g(rng, 'TCP_UTCL1_TRANSLATION_MISS_sum'), g(rng, 'TCP_UTCL1_REQUEST_sum')
wc, wi, wa, ai = g(rng, 'SQ_WAVE_CYCLES'), g(rng, 'SQ_WAIT_INST_ANY'), g(rng, 'SQ_WAIT_ANY'), g(rng, 'SQ_ACTIVE_INST_ANY')
creq, clat, cgui = g(cmb, 'TCP_TCC_READ_REQ_sum'), g(cmb, 'TCP_TCC_READ_REQ_LATENCY_sum'), g(cmb, 'GRBM_GUI_ACTIVE') / 8
cta, cwc, cwi, cwa = g(cmb, 'TA_TA_BUSY_sum'), g(cmb, 'SQ_WAVE_CYCLES'), g(cmb, 'SQ_WAIT_INST_ANY'), g(cmb, 'SQ_WAIT_ANY')
lds, valu, salu, spi = g(extra, 'SQ_LDS_IDX_ACTIVE'), g(extra, 'SQ_INSTS_VALU'), g(extra, 'SQ_INSTS_SALU'), g(extra, 'SPI_RA_WAVE_SIMD_FULL_CSN')
print(f"""## What holds k_spmv_range<double, 8, true, 2, double> (R-MAT 24, {SOURCES})?
## rocprofv3 --pmc passes (one counter group per pass, counters + kernel trace only), {HOW} on
## `bench.py --no-sub-configs --no-side-figures --steps 20 --warmup 5`; average per launch over {n} launches; SUMS over the chip: 256 CUs
## (TA / TCP / SQ), 128 L2 channels (TCC), 8 XCDs (GRBM).  Clock under the profiler 2.1 GHz.  (This file: scripts/experiments/pmc_range_report.py.)
{rng}
## Reading (per CU = sum / 256; the launch = GRBM_GUI_ACTIVE / 8 = {gui/1e6:.2f} M clocks = {gui/2.1e9*1e3:.2f} ms at 2.1 GHz under the profiler):
##   L1 read misses sent to L2        {req/1e6:.1f} M -> {req/256/1e3:.0f} k per CU and launch (~295 k of them cold gather lines = 28 % of the non-zeros, ~92 k stream lines, the rest table images / tile_ptr pairs that miss the scalar cache)
##   their summed latency             {lat/1e9:.2f} G clocks -> {lat/req:.0f} clocks per request on average (L2 hits ~300, HBM streams ~1000)
##   => requests in flight per CU     {lat/1e9:.2f} G / 256 / {gui/1e6:.2f} M = {lat/256/gui:.0f} on average, and the L1 reports its miss queue FULL
##      (TCP_PENDING_STALL) {pend/256/1e6:.2f} M of the {gui/1e6:.2f} M clocks = {pend/256/gui*100:.0f} % of the launch.  Throughput = slots / latency: {req/256/gui:.2f} lines per clock and CU.
##      (The shape probe, r04_probes.txt: streams alone 466 us, gathers + table reads alone 293 us, both together 810 us -- also with the two
##      classes of requests in different wavefronts.)
##   L2: {hit/1e6:.1f} M hits, {miss/1e6:.1f} M misses ({ea/1e6:.1f} M fabric reads x 128 B = {ea*128/1e9:.1f} GB: the streams + ~3 M cold lines; round 3: 45 M misses)
##   TLB: {tlbm/1e3:.1f} k misses per launch in {tlbr/1e6:.0f} M translations: the address translation is not it
##   address unit (TA) busy {ta/256/1e6:.2f} M of {gui/1e6:.2f} M clocks per CU = {ta/256/gui*100:.0f} %, {addr/256/1e6:.2f} M of them stalled by the L1 (ADDR_STALL {addr/256/gui*100:.0f} %)
##   wavefronts (8 per CU): {wi/wc*100:.0f} % waiting to issue (SQ_WAIT_INST_ANY), {wa/wc*100:.0f} % parked at s_waitcnt, {ai/wc*100:.0f} % issuing
## => the kernel is bound by the L1's outstanding-miss capacity x latency.  That is why prefetch depth 3, 16 vs 8 wavefronts, the
##    instruction count and the issue order all measure within 1-2 %, and why every percent of table coverage, every stream byte and every
##    vector-memory instruction shows (r04_probes.txt: flags in the codes -4 % stream bytes = -1..2 %; values in 16-byte loads -1.7 %;
##    CSR5HIP_OPT_NARROW_VALUES -36 % stream bytes = -12 %).
{EARLIER}
{cmb}
## more counter groups on the combine:
{extra}
## Reading: {creq/1e6:.2f} M L2 read requests of {clat/creq:.0f} clocks each (72 % miss L2: the partials were written 0.3 GB ago) in {cgui/1e3:.0f} k clocks per XCD =
## {clat/256/cgui:.0f} in flight per CU -- a quarter of what the L1 can hold -- with the address unit busy {cta/256/cgui*100:.0f} %, the LDS {lds/256/cgui*100:.0f} % (SQ_LDS_IDX_ACTIVE),
## {valu/65536:.0f} vector + {salu/65536:.0f} scalar instructions per wavefront, {cwc*4/256/cgui:.0f} wavefronts resident per CU of the 32 that fit (SQ_WAVE_CYCLES x 4 / 256 /
## launch clocks) although the dispatcher is almost never refused (SPI_RA_WAVE_SIMD_FULL {spi/1e6:.2f} M cycles).  {cwi/cwc*100:.0f} % of the wave time waits to issue,
## {cwa/cwc*100:.0f} % waits for data.  What was tried on it and did not help: r04_probes.txt, last sections; r05_probes.txt section 1 (the XCD skew, fixed).""")
