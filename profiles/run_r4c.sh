#!/bin/bash
# Round 4, GPU call C: why did the 2-rank bench fail and the 1-GPU bench not finish in call B?
set -x
O=gpurun_out
mkdir -p $O
BRIE_BENCH_SINGLE_DEVICE=1 BRIE_BENCH_STRICT=1 timeout 300 python bench.py --gpus 2 --config c1 --steps 5 --warmup 2 --no-pmc --cpu-seconds 1 > $O/r4c_bench_c1_n2.json 2> $O/r4c_bench_c1_n2.err
tail -c 3000 $O/r4c_bench_c1_n2.err
timeout 1200 python bench.py > $O/r4c_bench_c3_n1.json 2> $O/r4c_bench_c3_n1.err
tail -20 $O/r4c_bench_c3_n1.err
tail -c 3000 $O/r4c_bench_c3_n1.json
timeout 2400 python -m pytest tests/ -q -m gpu --deselect "tests/test_gpu_fullsize.py::test_psi_null_rule_on_gene_samples_of_the_full_size_configs_after_the_full_default_schedule[c3_api_512]" --deselect "tests/test_gpu_fullsize.py::test_psi_null_rule_on_gene_samples_of_the_full_size_configs_after_the_full_default_schedule[c3_api_512_s2]" --durations=15 > $O/r4c_pytest_gpu.log 2>&1
tail -60 $O/r4c_pytest_gpu.log
