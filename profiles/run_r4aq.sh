#!/bin/bash
# Round 4, GPU call AQ: the search's stopping rate for handles whose arrays are below 1 GiB (5850 instead of an unreachable
# 6050): placement tests, the product path at configs[1] in fresh processes, its bench line, an 8-way shard of configs[2]
set -x
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_placement.py -q -m gpu > $O/r4aq_pytest_placement.log 2>&1
tail -3 $O/r4aq_pytest_placement.log
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 100 python profiles/placement_auto.py --config c2 >> $O/r4aq_placement_auto_c2.jsonl 2>> $O/r4aq_placement_auto_c2.err
done
cut -c1-230 $O/r4aq_placement_auto_c2.jsonl
timeout 600 python bench.py --config c2 > $O/r4aq_bench_c2_n1.json 2> $O/r4aq_bench_c2_n1.err
tail -12 $O/r4aq_bench_c2_n1.err
timeout 600 python bench.py --config c3 --emulate-shard-of 8 --no-pmc --no-cpu-baseline --no-psi-check > $O/r4aq_bench_c3_shard_of8.json 2> $O/r4aq_bench_c3_shard_of8.err
tail -6 $O/r4aq_bench_c3_shard_of8.err
