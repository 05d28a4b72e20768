#!/usr/bin/env python3
"""profiles/r05_pmc_k_spmv.txt from the raw counter averages of scripts/experiments/round5/pmc_plain.sh: the raw blocks plus a
per-CU / per-tile reading of every block.

    python scripts/experiments/round5/pmc_plain_report.py gpurun_out/r05_pmc_plain_raw.txt > profiles/r05_pmc_k_spmv.txt
"""
import re
import sys

raw = open(sys.argv[1]).read()
blocks = re.split(r"^#### ", raw, flags=re.M)[1:]
TILES = {"nd24k": 28054, "webbase": 8087, "scircuit": 2497}  # tiles 0 .. p-2 at sigma 16 (nd24k: both kernels at 16) / 6 / 6

print("""## What holds the plain-path tile kernels on the SuiteSparse-shaped stand-ins (round 5, final sources)?
## csr5::k_spmv = one tile per wavefront (rounds 1-4; + narrow column codes on the x-window variant since round 5);
## csr5::k_spmv_walk = the range-walking, software-pipelined kernel of round 5 (CSR5HIP_OPT_TILE_WALK = 2; DEPTH 3, 8 ranges per CU).
## rocprofv3 --pmc passes (one counter group per pass, counters + kernel trace only), scripts/experiments/round5/pmc_plain.sh on
## `bench.py --workload W --slabs 0 --tile-walk off|force --no-sub-configs --no-side-figures --steps 20 --warmup 5`; averages over the
## 40 COLD-protocol launches (rotating copies of matrix / x / y beyond the Infinity Cache); nd24k-like at sigma = 16 for both kernels
## (the walking kernel is compiled for sigma <= 16; 16 is also the fp32 auto rule's choice).  Sums over the chip: 256 CUs (TA / TCP / SQ),
## 128 L2 channels (TCC), 8 XCDs (GRBM).  SQ_* counters are in units of 4 clocks per wavefront.  Clock under the profiler ~2.1 GHz.
""")


def g(block, name):
    m = re.search(name + r"\s+n=\s*\d+ avg=\s*([0-9.]+)", block)
    return float(m.group(1)) if m else float("nan")


for b in blocks:
    head = b.splitlines()[0]
    w = head.split()[0]
    tiles = TILES.get(w, 1)
    print("#### " + b.rstrip())
    gui = g(b, "GRBM_GUI_ACTIVE") / 8
    req, lat = g(b, "TCP_TCC_READ_REQ_sum"), g(b, "TCP_TCC_READ_REQ_LATENCY_sum")
    pend, ta = g(b, "TCP_PENDING_STALL_CYCLES_sum"), g(b, "TA_TA_BUSY_sum")
    hit, miss, ea = g(b, "TCC_HIT_sum"), g(b, "TCC_MISS_sum"), g(b, "TCC_EA0_RDREQ_sum")
    wc, wi, wa, ai = g(b, "SQ_WAVE_CYCLES"), g(b, "SQ_WAIT_INST_ANY"), g(b, "SQ_WAIT_ANY"), g(b, "SQ_ACTIVE_INST_ANY")
    valu, salu, lds, vrd, waves = g(b, "SQ_INSTS_VALU"), g(b, "SQ_INSTS_SALU"), g(b, "SQ_INSTS_LDS"), g(b, "SQ_INSTS_VMEM_RD"), g(b, "SQ_WAVES")
    print(f"""## Reading ({w}: {tiles} tiles; the launch = GRBM_GUI_ACTIVE / 8 = {gui / 1e3:.1f} k clocks):
##   L1 read misses sent to L2      {req / 1e6:.2f} M = {req / 256 / 1e3:.1f} k per CU, {req / tiles:.0f} per tile; {ea / 1e6:.2f} M of them went on to the fabric (x 128 B = {ea * 128 / 1e6:.0f} MB)
##   their latency                  {lat / req:.0f} clocks on average -> {lat / 256 / gui:.0f} requests in flight per CU; the L1 reports its miss queue full
##                                  (TCP_PENDING_STALL) {100 * pend / 256 / gui:.0f} % of the launch
##   L2                             {hit / 1e6:.2f} M hits, {miss / 1e6:.2f} M misses
##   address unit (TA) busy         {100 * ta / 256 / gui:.0f} % of the launch
##   instructions per tile          {valu / tiles:.0f} vector ALU, {salu / tiles:.0f} scalar, {lds / tiles:.0f} LDS, {vrd / tiles:.0f} vector-memory reads  ({waves:.0f} wavefronts)
##   a wavefront's clocks per tile  {4 * wc / tiles:.0f} resident = {4 * ai / tiles:.0f} issuing + {4 * wi / tiles:.0f} waiting to issue + {4 * wa / tiles:.0f} parked at s_waitcnt (+ rest)
##   vector ALU time per SIMD       {valu * 4 / 1024 / 1e3:.1f} k clocks = {100 * valu * 4 / 1024 / gui:.0f} % of the launch; wavefronts resident per CU {4 * wc / 256 / gui:.1f}
""")
