#!/bin/bash
# Round 4, GPU calls W...: the product path in fresh processes -- what does the automatic search find, what does the step run at?
set -x
O=gpurun_out
TAG=${1:-r4w}
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 200 python profiles/placement_auto.py --config c3 >> $O/${TAG}_placement_auto_c3.jsonl 2>> $O/${TAG}_placement_auto.err
done
for i in 1 2 3 4 5 6; do
  timeout 100 python profiles/placement_auto.py --config c2 >> $O/${TAG}_placement_auto_c2.jsonl 2>> $O/${TAG}_placement_auto.err
done
cat $O/${TAG}_placement_auto_c3.jsonl $O/${TAG}_placement_auto_c2.jsonl
