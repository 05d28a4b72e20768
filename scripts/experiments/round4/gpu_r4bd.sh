#!/bin/bash
# round 4, call bd: R-MAT 25 / 26 again with the graph instantiated before the timed call
timeout 900 python scripts/experiments/scale_check.py --scale 25 2>&1 | tail -1
timeout 1500 python scripts/experiments/scale_check.py --scale 26 2>&1 | tail -1
