#!/bin/bash
# (the split-around-a-spacer candidate existed in the library for this call only: it did nothing for 0.2-GB arrays and was removed)
# Round 4, GPU call Z: the search with a split-around-a-spacer candidate first (then interleaved ones), product path, fresh processes
set -x
O=gpurun_out
TAG=${1:-r4z}
export BRIE_PLACEMENT_LOG=1
timeout 600 python -m pytest tests/test_gpu_placement.py -q -m gpu > $O/${TAG}_pytest_placement.log 2>&1
tail -3 $O/${TAG}_pytest_placement.log
for i in 1 2 3 4 5 6 7 8; do
  timeout 200 python profiles/placement_auto.py --config c3 >> $O/${TAG}_placement_auto_c3.jsonl 2>> $O/${TAG}_placement_auto_c3.err
done
for i in 1 2 3 4 5 6 7 8; do
  timeout 100 python profiles/placement_auto.py --config c2 >> $O/${TAG}_placement_auto_c2.jsonl 2>> $O/${TAG}_placement_auto_c2.err
done
cat $O/${TAG}_placement_auto_c3.jsonl $O/${TAG}_placement_auto_c2.jsonl
