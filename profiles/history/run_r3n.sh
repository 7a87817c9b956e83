#!/bin/bash
# round 3, GPU call n: a wider soak (arbitrary shapes: every residue of Ng mod 4 and mod 256) in assert mode
O=gpurun_out
mkdir -p $O
python tests/tools/soak_randomised.py 300 200 555001 900 > $O/r3n_soak_wide.log 2>&1; echo "soak rc=$?" >> $O/r3n_soak_wide.log
grep -E "FAILED|done|rc=" $O/r3n_soak_wide.log | cut -c1-700 | tail -20
