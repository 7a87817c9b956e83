#!/bin/bash
# Round 4, GPU call K: spacing of the arrays vs physical memory -- layouts inside ONE slab, several fresh processes
set -x
O=gpurun_out
for i in 1 2 3 4 5 6; do
  timeout 300 python profiles/layout_probe.py --rounds 2 >> $O/r4k_layout_probe.jsonl 2>> $O/r4k_layout_probe.err
done
cat $O/r4k_layout_probe.jsonl
tail -3 $O/r4k_layout_probe.err
