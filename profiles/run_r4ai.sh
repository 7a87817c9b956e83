#!/bin/bash
# Round 4, GPU call AI: the Wg_loc gradient of the gene-design panels on the matrix cores -- parity and sharded tests, then
# what the panels cost at the headline shape
set -x
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_comm.py tests/test_gpu_distributed.py -q -m gpu -k "very_wide or coupled or allreduce" > $O/r4ai_pytest_kg.log 2>&1
tail -5 $O/r4ai_pytest_kg.log
timeout 600 python profiles/wide_ab.py --rounds 2 --steps 4 --cases 3:64,3:128,3:256,0:100,128:0 > $O/r4ai_panels_at_c3.log 2>&1
tail -3 $O/r4ai_panels_at_c3.log
