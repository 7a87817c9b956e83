#!/bin/bash
# Round 4, GPU call BC: the driver's own commands on the tree as it stands
O=gpurun_out
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/r4bc_pytest_gpu.log 2>&1
tail -3 $O/r4bc_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r4bc_smoke.log 2>&1
tail -1 $O/r4bc_smoke.log
timeout 900 python bench.py > $O/r4bc_bench_c3_n1.json 2> $O/r4bc_bench_c3_n1.err
grep "placement\|timed" $O/r4bc_bench_c3_n1.err
