#!/bin/bash
# Round 4, GPU call BH: configs[4] (DMG, 100k x 30k, Kc = 5) whole on one GPU (a 150-GB set) and as the shard one of 8 ranks
# would own, with the final library
O=gpurun_out
timeout 1500 python bench.py --config c5 --no-pmc --no-cpu-baseline --no-psi-check > $O/r4bh_bench_c5_n1.json 2> $O/r4bh_bench_c5_n1.err
grep "placement\|timed\|whole" $O/r4bh_bench_c5_n1.err
timeout 900 python bench.py --config c5 --emulate-shard-of 8 --no-pmc --no-cpu-baseline --no-psi-check --no-e2e > $O/r4bh_bench_c5_shard_of8.json 2> $O/r4bh_bench_c5_shard_of8.err
grep "placement\|timed" $O/r4bh_bench_c5_shard_of8.err
