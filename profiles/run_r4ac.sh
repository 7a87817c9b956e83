#!/bin/bash
# Round 4, GPU call AC: the state slab for small problems -- placement tests (incl. packing on a slab), product path in
# fresh processes at configs[1] and an 8-way shard of configs[2], then the whole suite
set -x
O=gpurun_out
export BRIE_PLACEMENT_LOG=1
timeout 600 python -m pytest tests/test_gpu_placement.py -q -m gpu -x > $O/r4ac_pytest_placement.log 2>&1
tail -5 $O/r4ac_pytest_placement.log
for i in 1 2 3 4 5 6 7 8; do
  timeout 100 python profiles/placement_auto.py --config c2 >> $O/r4ac_placement_auto_c2.jsonl 2>> $O/r4ac_placement_auto_c2.err
done
cat $O/r4ac_placement_auto_c2.jsonl
unset BRIE_PLACEMENT_LOG
timeout 600 python bench.py --config c2 > $O/r4ac_bench_c2_n1.json 2> $O/r4ac_bench_c2_n1.err
tail -12 $O/r4ac_bench_c2_n1.err
timeout 600 python bench.py --config c3 --emulate-shard-of 8 --no-pmc --no-cpu-baseline --no-psi-check > $O/r4ac_bench_c3_shard_of8.json 2> $O/r4ac_bench_c3_shard_of8.err
tail -6 $O/r4ac_bench_c3_shard_of8.err
timeout 2400 python -m pytest tests/ -q -m gpu > $O/r4ac_pytest_gpu.log 2>&1
tail -5 $O/r4ac_pytest_gpu.log
