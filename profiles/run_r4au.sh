#!/bin/bash
# Round 4, GPU call AU: rows per chunk by the new rule (n_chunks = 128 / 256 for the streaming kernel) -- the whole suite,
# bench lines of configs[1] / [2] / an 8-way shard, MC_size 3 at configs[1]
set -x
O=gpurun_out
timeout 2400 python -m pytest tests/ -q -m gpu > $O/r4au_pytest_gpu.log 2>&1
tail -4 $O/r4au_pytest_gpu.log
timeout 900 python bench.py > $O/r4au_bench_c3_n1.json 2> $O/r4au_bench_c3_n1.err
grep "placement\|timed" $O/r4au_bench_c3_n1.err
timeout 600 python bench.py --config c2 > $O/r4au_bench_c2_n1.json 2> $O/r4au_bench_c2_n1.err
grep "placement\|timed" $O/r4au_bench_c2_n1.err
timeout 600 python bench.py --config c3 --emulate-shard-of 8 --no-pmc --no-cpu-baseline --no-psi-check > $O/r4au_bench_c3_shard_of8.json 2> $O/r4au_bench_c3_shard_of8.err
grep "placement\|timed" $O/r4au_bench_c3_shard_of8.err
timeout 300 python profiles/rpc_sweep.py --config c2 --mc 3 --rpc 64,79,200 --reps 2 --out $O/r4au_rpc_c2_mc3.json > $O/r4au_rpc_c2_mc3.log 2>&1
tail -4 $O/r4au_rpc_c2_mc3.log | cut -c1-200
timeout 300 python profiles/rpc_sweep.py --config c2 --rpc 64,79 --reps 3 --out $O/r4au_rpc_c2.json > $O/r4au_rpc_c2.log 2>&1
tail -3 $O/r4au_rpc_c2.log | cut -c1-200
