#!/bin/bash
# Round 5, GPU call A: the placement search after its rewrite (rounds of three held candidates up to 8 sets, best effort,
# HBM / time bounds) -- its tests, then the product path in fresh processes: 12 x configs[2] at the defaults,
# 6 x with the round-4 cap of 4 sets (A/B on the same box), 4 x configs[1]
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_placement.py -x -q > $O/r5a_pytest_placement.log 2>&1
tail -3 $O/r5a_pytest_placement.log
for i in 1 2 3 4 5 6; do
  timeout 300 python profiles/placement_auto.py --config c3 >> $O/r5a_auto_c3_tries8.jsonl 2>> $O/r5a_auto.err
  BRIE_PLACEMENT_TRIES=4 timeout 300 python profiles/placement_auto.py --config c3 >> $O/r5a_auto_c3_tries4.jsonl 2>> $O/r5a_auto.err
  timeout 300 python profiles/placement_auto.py --config c3 >> $O/r5a_auto_c3_tries8.jsonl 2>> $O/r5a_auto.err
done
for i in 1 2 3 4; do
  timeout 300 python profiles/placement_auto.py --config c2 >> $O/r5a_auto_c2_tries8.jsonl 2>> $O/r5a_auto.err
done
cat $O/r5a_auto_c3_tries8.jsonl $O/r5a_auto_c3_tries4.jsonl $O/r5a_auto_c2_tries8.jsonl
tail -5 $O/r5a_auto.err
