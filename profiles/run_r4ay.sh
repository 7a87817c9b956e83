#!/bin/bash
# Round 4, GPU call AY: the tree as it stands (new tiling rule, bench preconditioning) -- smoke, the default bench line and
# configs[1] / [0] with their defaults, rocprofv3 summary + PMC passes of the bench command
set -x
O=gpurun_out
timeout 300 python __graft_entry__.py smoke > $O/r4ay_smoke.log 2>&1
tail -1 $O/r4ay_smoke.log
timeout 900 python bench.py > $O/r4ay_bench_c3_n1.json 2> $O/r4ay_bench_c3_n1.err
tail -13 $O/r4ay_bench_c3_n1.err
timeout 600 python bench.py --config c2 > $O/r4ay_bench_c2_n1.json 2> $O/r4ay_bench_c2_n1.err
grep "placement\|timed" $O/r4ay_bench_c2_n1.err
timeout 600 python bench.py --config c1 > $O/r4ay_bench_c1_n1.json 2> $O/r4ay_bench_c1_n1.err
timeout 600 bash profiles/run_profile.sh r4ay > $O/r4ay_run_profile.log 2>&1
head -7 $O/prof_r4ay/summary.txt
tail -12 $O/prof_r4ay/summary.txt
