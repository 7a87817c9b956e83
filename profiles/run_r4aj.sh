#!/bin/bash
# Round 4, GPU call AJ: per-kernel times of a step with panels at the headline shape (Kc = 3, Kg = 128)
set -x
O=$(pwd)/gpurun_out
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r4aj -o t -- python3 $R/profiles/wide_ab.py --rounds 1 --steps 4 --cases 3:128 > $O/r4aj_run.log 2>&1
f=$(find /tmp/r4aj -name "*kernel_stats.csv" | head -1)
cp $f $O/r4aj_kg128_kernel_stats.csv
head -14 $O/r4aj_kg128_kernel_stats.csv | cut -c1-200
