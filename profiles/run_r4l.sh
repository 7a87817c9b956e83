#!/bin/bash
# Round 4, GPU call L: arrays spread over a 230-GB slab in groups -- how does the rate depend on the spread?
set -x
O=gpurun_out
for i in 1 2 3; do
  timeout 300 python profiles/layout_probe.py --spread 230 --rounds 2 >> $O/r4l_layout_spread.jsonl 2>> $O/r4l_layout_spread.err
done
cat $O/r4l_layout_spread.jsonl
tail -3 $O/r4l_layout_spread.err
