#!/bin/bash
# Round 4, GPU call AO: soak of the final library (panel products on the matrix cores: panel_prior_mean now also serves
# loss_gene / the per-entry accessors of every design with 9+ features), fresh seeds
timeout 2000 python tests/tools/soak_randomised.py 300 150 737373 100 1000 > gpurun_out/r4ao_soak_737373.log 2>&1
tail -6 gpurun_out/r4ao_soak_737373.log
