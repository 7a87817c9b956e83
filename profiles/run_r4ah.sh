#!/bin/bash
# Round 4, GPU call AH: the final library (gene-design panels in) -- smoke, the default bench line + the other configs,
# rocprofv3 summary of the bench command, and what the panel fallbacks cost at the headline shape
set -x
O=gpurun_out
mkdir -p $O
timeout 300 python __graft_entry__.py smoke > $O/r4ah_smoke.log 2>&1
tail -2 $O/r4ah_smoke.log
timeout 900 python bench.py > $O/r4ah_bench_c3_n1.json 2> $O/r4ah_bench_c3_n1.err
tail -14 $O/r4ah_bench_c3_n1.err
timeout 600 python bench.py --config c2 > $O/r4ah_bench_c2_n1.json 2> $O/r4ah_bench_c2_n1.err
timeout 600 python bench.py --config c1 > $O/r4ah_bench_c1_n1.json 2> $O/r4ah_bench_c1_n1.err
timeout 600 bash profiles/run_profile.sh r4ah > $O/r4ah_run_profile.log 2>&1
head -8 $O/prof_r4ah/summary.txt
timeout 600 python profiles/wide_ab.py --rounds 2 --steps 4 --cases 3:64,3:128,3:256,64:0,128:0 > $O/r4ah_panels_at_c3.log 2>&1
tail -3 $O/r4ah_panels_at_c3.log
