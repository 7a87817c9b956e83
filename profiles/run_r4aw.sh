#!/bin/bash
# Round 4, GPU call AW: bench defaults for the sub-millisecond configs (50 + 200 steps at configs[1], 100 + 500 at configs[0])
set -x
O=gpurun_out
timeout 600 python bench.py --config c2 > $O/r4aw_bench_c2_n1.json 2> $O/r4aw_bench_c2_n1.err
grep "placement\|timed" $O/r4aw_bench_c2_n1.err
timeout 600 python bench.py --config c1 > $O/r4aw_bench_c1_n1.json 2> $O/r4aw_bench_c1_n1.err
grep "placement\|timed" $O/r4aw_bench_c1_n1.err
timeout 600 python bench.py --config c2 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-psi-check --no-e2e > $O/r4aw_bench_c2_short.json 2> $O/r4aw_bench_c2_short.err
grep "placement\|timed" $O/r4aw_bench_c2_short.err
