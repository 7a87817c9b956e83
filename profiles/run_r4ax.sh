#!/bin/bash
# Round 4, GPU call AX: clock preconditioning with the effect-free probe before the timed region -- the N > 1 bench tests, the
# default line, and short windows (--steps 20 --warmup 3, what a driver may pass) at configs[1] and an 8-way shard
set -x
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_bench.py -q -m gpu > $O/r4ax_pytest_bench.log 2>&1
tail -3 $O/r4ax_pytest_bench.log
timeout 900 python bench.py > $O/r4ax_bench_c3_n1.json 2> $O/r4ax_bench_c3_n1.err
grep "placement\|timed\|precond" $O/r4ax_bench_c3_n1.err
timeout 600 python bench.py --config c2 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-psi-check --no-e2e > $O/r4ax_bench_c2_short.json 2> $O/r4ax_bench_c2_short.err
grep "placement\|timed\|precond" $O/r4ax_bench_c2_short.err
timeout 600 python bench.py --config c3 --emulate-shard-of 8 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-psi-check --no-e2e > $O/r4ax_bench_c3_of8_short.json 2> $O/r4ax_bench_c3_of8_short.err
grep "placement\|timed\|precond" $O/r4ax_bench_c3_of8_short.err
