#!/bin/bash
# Round 4, GPU call BF: last soak of the library as it stands (fresh seeds; both random families, arbitrary shapes incl. designs
# up to 160 features on either side)
timeout 2400 python tests/tools/soak_randomised.py 500 250 848484 150 900 > gpurun_out/r4bf_soak_848484.log 2>&1
tail -6 gpurun_out/r4bf_soak_848484.log
