#!/bin/bash
# Round 4, GPU calls N / O: spacers of 0 vs 18 GB between the arrays of the candidate sets, alternating fresh processes
set -x
O=gpurun_out
TAG=${1:-r4n}
export BRIE_PLACEMENT_LOG=1
for i in 1 2 3 4 5; do
  for sp in 0 18; do
    BRIE_PLACEMENT_SPACER_GB=$sp timeout 200 python profiles/placement_ab.py --config c3 --handles 1 --tries 4 --out $O/${TAG}_spacer_${sp}_c3.jsonl > /dev/null 2>> $O/${TAG}_spacer_${sp}_c3.err
  done
done
for sp in 0 18; do
  grep -h "brie placement" $O/${TAG}_spacer_${sp}_c3.err | awk '{print $4, $5}' | tr '\n' ' '; echo " <- spacer $sp GB"
done
