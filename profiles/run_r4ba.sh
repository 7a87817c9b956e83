#!/bin/bash
# Round 4, GPU call BA: a gene design of 70 features through BRIE2.fit (the Python API end to end)
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "coupled_fit_through_python_api" > gpurun_out/r4ba_pytest_api_kg.log 2>&1
tail -15 gpurun_out/r4ba_pytest_api_kg.log
