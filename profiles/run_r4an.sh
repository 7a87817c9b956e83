#!/bin/bash
# Round 4, GPU call AN: the final library (panel products on the matrix cores) -- smoke, the default bench line + the other configs,
# rocprofv3 summary of the bench command, and what the panel fallbacks cost at the headline shape
set -x
O=gpurun_out
mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu --durations=8 > $O/r4an_pytest_gpu.log 2>&1
tail -14 $O/r4an_pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/r4an_smoke.log 2>&1
tail -2 $O/r4an_smoke.log
timeout 900 python bench.py > $O/r4an_bench_c3_n1.json 2> $O/r4an_bench_c3_n1.err
tail -14 $O/r4an_bench_c3_n1.err
timeout 600 python bench.py --config c2 > $O/r4an_bench_c2_n1.json 2> $O/r4an_bench_c2_n1.err
timeout 600 python bench.py --config c1 > $O/r4an_bench_c1_n1.json 2> $O/r4an_bench_c1_n1.err
timeout 600 bash profiles/run_profile.sh r4an > $O/r4an_run_profile.log 2>&1
head -8 $O/prof_r4an/summary.txt
