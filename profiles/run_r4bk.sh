#!/bin/bash
# Round 4, GPU call BK: the final binary (wide_design_grad change in) -- the whole suite with the driver's flags, smoke, default bench
O=gpurun_out
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/r4bk_pytest_gpu.log 2>&1
tail -2 $O/r4bk_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r4bk_smoke.log 2>&1
tail -1 $O/r4bk_smoke.log
timeout 900 python bench.py > $O/r4bk_bench_c3_n1.json 2> $O/r4bk_bench_c3_n1.err
grep "placement\|timed" $O/r4bk_bench_c3_n1.err
timeout 600 python bench.py --config c2 > $O/r4bk_bench_c2_n1.json 2> $O/r4bk_bench_c2_n1.err
grep "placement\|timed" $O/r4bk_bench_c2_n1.err
