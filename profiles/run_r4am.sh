#!/bin/bash
# Round 4, GPU call AM: the Wg_loc gradient with the residual tile staged through LDS -- parity / sharded tests of the
# gene designs, what the panels cost at the headline shape, per-kernel times
set -x
O=$(pwd)/gpurun_out
R=$(pwd)
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_comm.py tests/test_gpu_distributed.py -q -m gpu -k "very_wide or coupled or allreduce" > $O/r4am_pytest_kg.log 2>&1
tail -3 $O/r4am_pytest_kg.log
timeout 600 python profiles/wide_ab.py --rounds 2 --steps 4 --cases 3:128,3:256,0:100 > $O/r4am_panels_at_c3.log 2>&1
tail -2 $O/r4am_panels_at_c3.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r4am -o t -- python3 $R/profiles/wide_ab.py --rounds 1 --steps 4 --cases 3:128 > $O/r4am_run.log 2>&1
f=$(find /tmp/r4am -name "*kernel_stats.csv" | head -1)
cp $f $O/r4am_panels_kernel_stats.csv
grep "gene_design_grad\|panel_prior_mean" $O/r4am_panels_kernel_stats.csv | cut -c1-200
