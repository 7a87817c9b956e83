#!/bin/bash
# Round 4, GPU call U: the layout-probe test, then one soak of the two random test families with fresh seeds on the final
# library (shapes now include very wide cell designs, Kc 65..160, in 64-feature panels)
set -x
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_placement.py -q -m gpu > $O/r4u_pytest_placement.log 2>&1
tail -3 $O/r4u_pytest_placement.log
timeout 2400 python tests/tools/soak_randomised.py 400 200 40417 150 450 > $O/r4u_soak_40417.log 2>&1
tail -6 $O/r4u_soak_40417.log
