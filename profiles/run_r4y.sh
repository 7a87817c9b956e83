#!/bin/bash
# Round 4, GPU call Y: a SMALL set (configs[1] shape, 0.2-GB arrays) inside a 64-GB slab: packed, pitches 0.5 .. 8 GB, groups
set -x
O=gpurun_out
for i in 1 2 3 4; do
  timeout 300 python profiles/layout_probe.py --nc 10000 --ng 5000 --small 64 --rounds 2 >> $O/r4y_layout_small.jsonl 2>> $O/r4y_layout_small.err
done
cat $O/r4y_layout_small.jsonl
