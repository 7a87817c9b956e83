#!/bin/bash
# Round 4, GPU call G: where do the sets of the placement search live (addresses) next to their rates?
set -x
O=gpurun_out
export BRIE_PLACEMENT_LOG=1
for i in 1 2 3 4 5 6; do
  timeout 200 python profiles/placement_ab.py --config c3 --handles 2 --tries 4 --out $O/r4g_placement_log_c3.jsonl > /dev/null 2>> $O/r4g_placement_log_c3.err
  timeout 100 python profiles/placement_ab.py --config c2 --handles 3 --tries 4 --out $O/r4g_placement_log_c2.jsonl > /dev/null 2>> $O/r4g_placement_log_c2.err
done
grep "brie placement" $O/r4g_placement_log_c3.err | head -60
