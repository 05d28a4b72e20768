"""Row super-blocks of the hot path, EMULATED ("reading P while it is cached"): the W cost-balanced row blocks
of ONE R-MAT 24 run one after the other on one GPU, each through its own handle (16 slabs, hot table, x captured once), next to
the whole matrix in the same call.  A block's partial sums P (320 MB / W) are Infinity-Cache-resident when its combine reads
them -- exactly what a super-blocked persistent kernel would buy -- and a block pays what that kernel would pay per super-block:
the table refills, the slab-boundary waits, the range seams.  (It also pays three launch boundaries per block, ~3 us each: the
emulation is pessimistic by ~10 us per block, stated with the result.)  sum over the blocks < whole => worth building.
Usage: python superblock_emulate.py [--scale 24] [--worlds 2,4,8] [--steps 30]"""
import argparse
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
import shard_alone  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=24)
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()
    import torch
    dev = torch.device("cuda:0")
    whole = None
    for w in [int(v) for v in args.worlds.split(",")]:
        recs = [shard_alone.measure_block(args.scale, w, r, dev, steps=args.steps, x_snapshot=1)[0] for r in range(w)]
        total = sum(r["us"] for r in recs)
        if w == 1:
            whole = total
        print(json.dumps({"row_super_blocks": w, "sum_us": round(total, 1), "blocks_us": [r["us"] for r in recs],
                          "slabs": [r["slabs"] for r in recs], "hot_cover_pct": [r["hot_cover_pct"] for r in recs],
                          "P_bytes_per_block_MB": round(40e6 * 8 * (1 << (args.scale - 24)) / w, 1),
                          "vs_whole": None if whole is None else round(total / whole, 3),
                          "minus_launch_boundaries_us": round(total - 10.0 * w, 1)}), flush=True)


if __name__ == "__main__":
    main()
