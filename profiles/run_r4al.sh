#!/bin/bash
# Round 4, GPU call AL: as call AK after the loads of both MFMA panel kernels were made unconditional and double-buffered
# (call AK: one memory latency per MFMA) -- the whole suite, what the panels cost at the headline shape, per-kernel times at Kc = 3, Kg = 128
set -x
O=$(pwd)/gpurun_out
R=$(pwd)
timeout 2400 python -m pytest tests/ -q -m gpu > $O/r4al_pytest_gpu.log 2>&1
tail -5 $O/r4al_pytest_gpu.log
timeout 600 python profiles/wide_ab.py --rounds 2 --steps 4 --cases 3:64,3:128,3:256,0:100,128:0,256:0 > $O/r4al_panels_at_c3.log 2>&1
tail -2 $O/r4al_panels_at_c3.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r4al -o t -- python3 $R/profiles/wide_ab.py --rounds 1 --steps 4 --cases 3:128,128:0 > $O/r4al_run.log 2>&1
f=$(find /tmp/r4al -name "*kernel_stats.csv" | head -1)
cp $f $O/r4al_panels_kernel_stats.csv
head -12 $O/r4al_panels_kernel_stats.csv | cut -c1-200
