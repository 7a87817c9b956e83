#!/bin/bash
# Round 4, GPU call AG: soak of the arbitrary-shape family after the gene-design panels (a quarter of the gene designs
# of the xg kinds now 65 .. 160 features), other seeds
timeout 1500 python tests/tools/soak_randomised.py 100 60 626262 60 900 > gpurun_out/r4ag_soak_626262.log 2>&1
tail -6 gpurun_out/r4ag_soak_626262.log
