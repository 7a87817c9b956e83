#!/bin/bash
# Round 4, GPU call AB: a second soak of the final library with other seeds (shapes incl. very wide designs, sequences)
set -x
timeout 2400 python tests/tools/soak_randomised.py 500 250 515151 200 600 > gpurun_out/r4ab_soak_515151.log 2>&1
tail -6 gpurun_out/r4ab_soak_515151.log
