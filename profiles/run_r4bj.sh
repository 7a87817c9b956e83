#!/bin/bash
# Round 4, GPU call BJ: wide_design_grad with unconditional, double-buffered loads -- wide-design tests, per-kernel times at Kc = 128
set -x
O=$(pwd)/gpurun_out
R=$(pwd)
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "wide or coupled or marginlik" > $O/r4bj_pytest_wide.log 2>&1
tail -3 $O/r4bj_pytest_wide.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r4bj -o t -- python3 $R/profiles/wide_ab.py --rounds 2 --steps 4 --cases 128:0,32:0 > $O/r4bj_run.log 2>&1
tail -2 $O/r4bj_run.log | cut -c1-400
f=$(find /tmp/r4bj -name "*kernel_stats.csv" | head -1)
cp $f $O/r4bj_kernel_stats.csv
grep "wide_design_grad\|panel_prior_mean" $O/r4bj_kernel_stats.csv | cut -c1-200
