#!/bin/bash
# Round 4, GPU call M: candidate sets allocated with spacers between the arrays (separate hipMallocs), fresh processes
set -x
O=gpurun_out
export BRIE_PLACEMENT_LOG=1
for sp in 0 8 18 28; do
  for i in 1 2 3; do
    BRIE_PLACEMENT_SPACER_GB=$sp timeout 200 python profiles/placement_ab.py --config c3 --handles 1 --tries 4 --out $O/r4m_spacer_${sp}_c3.jsonl > /dev/null 2>> $O/r4m_spacer_${sp}_c3.err
  done
  grep -h "brie placement" $O/r4m_spacer_${sp}_c3.err | awk '{print $4, $5}' | tr '\n' ' '; echo " <- spacer $sp GB"
done
