#!/bin/bash
# Round 4, GPU call D: the default bench line (bounded CPU legs), the tests that failed or were missing in call C,
# the rocprofv3 summary of the bench command
set -x
O=gpurun_out
mkdir -p $O
timeout 900 python bench.py > $O/r4d_bench_c3_n1.json 2> $O/r4d_bench_c3_n1.err
tail -12 $O/r4d_bench_c3_n1.err
timeout 1200 python -m pytest tests/test_gpu_bench.py tests/test_gpu_placement.py tests/test_gpu_fullsize.py "tests/test_gpu_parity.py::test_very_wide_cell_design_runs_in_panels" "tests/test_gpu_parity.py::test_wide_cell_design_matches_oracle" -q -m gpu --durations=8 > $O/r4d_pytest_part.log 2>&1
tail -40 $O/r4d_pytest_part.log
timeout 600 bash profiles/run_profile.sh r4d > $O/r4d_run_profile.log 2>&1
tail -30 $O/r4d_run_profile.log
timeout 600 python bench.py --config c2 > $O/r4d_bench_c2_n1.json 2> $O/r4d_bench_c2_n1.err
tail -4 $O/r4d_bench_c2_n1.err
