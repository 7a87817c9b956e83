#!/bin/bash
# Round 4, GPU call B: the whole GPU suite on the pruned tree with the placement search, then the bench line
set -x
O=gpurun_out
mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu --deselect "tests/test_gpu_fullsize.py::test_psi_null_rule_on_gene_samples_of_the_full_size_configs_after_the_full_default_schedule[c3_api_512]" --deselect "tests/test_gpu_fullsize.py::test_psi_null_rule_on_gene_samples_of_the_full_size_configs_after_the_full_default_schedule[c3_api_512_s2]" --durations=15 > $O/r4b_pytest_gpu.log 2>&1
tail -40 $O/r4b_pytest_gpu.log
timeout 900 python bench.py > $O/r4b_bench_c3_n1.json 2> $O/r4b_bench_c3_n1.err
tail -c 6000 $O/r4b_bench_c3_n1.json
tail -5 $O/r4b_bench_c3_n1.err
