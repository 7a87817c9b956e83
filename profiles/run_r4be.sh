#!/bin/bash
# Round 4, GPU call BE: per-rank projection of the strong split of configs[2]: rank 0's shard of 2 / 4 / 8 alone on one GPU,
# with the flags a driver passes (--steps 20 --warmup 3)
O=gpurun_out
for n in 2 4 8; do
  timeout 600 python bench.py --config c3 --emulate-shard-of $n --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-psi-check --no-e2e > $O/r4be_bench_c3_shard_of$n.json 2> $O/r4be_bench_c3_shard_of$n.err
  grep "placement\|timed" $O/r4be_bench_c3_shard_of$n.err
done
timeout 600 python bench.py --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-psi-check --no-e2e > $O/r4be_bench_c3_whole.json 2> $O/r4be_bench_c3_whole.err
grep "placement\|timed" $O/r4be_bench_c3_whole.err
