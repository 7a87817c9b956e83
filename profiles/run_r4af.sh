#!/bin/bash
# Round 4, GPU call AF: gene designs beyond 64 features (Kg > 64) in panels -- the new parity tests, the coupled / wide /
# sharded tests around them, then the whole suite
set -x
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "very_wide or coupled or marginlik or errors_are_loud" > $O/r4af_pytest_kg.log 2>&1
tail -15 $O/r4af_pytest_kg.log
timeout 900 python -m pytest tests/test_gpu_comm.py tests/test_gpu_distributed.py -q -m gpu > $O/r4af_pytest_sharded.log 2>&1
tail -8 $O/r4af_pytest_sharded.log
timeout 2400 python -m pytest tests/ -q -m gpu > $O/r4af_pytest_gpu.log 2>&1
tail -5 $O/r4af_pytest_gpu.log
