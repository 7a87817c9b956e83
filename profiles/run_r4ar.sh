#!/bin/bash
# Round 4, GPU call AR: candidates of small handles allocated one at a time (lazily) -- the product path at configs[1] in
# fresh processes; then the final library once more: whole suite, smoke, bench lines, rocprofv3 summary
set -x
O=gpurun_out
mkdir -p $O
for i in 1 2 3 4 5 6 7 8; do
  timeout 100 python profiles/placement_auto.py --config c2 >> $O/r4ar_placement_auto_c2.jsonl 2>> $O/r4ar_placement_auto_c2.err
done
cut -c1-230 $O/r4ar_placement_auto_c2.jsonl
timeout 2400 python -m pytest tests/ -q -m gpu > $O/r4ar_pytest_gpu.log 2>&1
tail -4 $O/r4ar_pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/r4ar_smoke.log 2>&1
tail -2 $O/r4ar_smoke.log
timeout 900 python bench.py > $O/r4ar_bench_c3_n1.json 2> $O/r4ar_bench_c3_n1.err
tail -14 $O/r4ar_bench_c3_n1.err
timeout 600 python bench.py --config c2 > $O/r4ar_bench_c2_n1.json 2> $O/r4ar_bench_c2_n1.err
grep placement $O/r4ar_bench_c2_n1.err
timeout 600 python bench.py --config c1 > $O/r4ar_bench_c1_n1.json 2> $O/r4ar_bench_c1_n1.err
timeout 600 bash profiles/run_profile.sh r4ar > $O/r4ar_run_profile.log 2>&1
head -6 $O/prof_r4ar/summary.txt
