#!/bin/bash
# round 3, GPU call o: the shape / switch family over arbitrary shapes (coupled, wide, marginLik, fixed ...), assert mode
O=gpurun_out
mkdir -p $O
python tests/tools/soak_randomised.py 0 0 880011 600 900 > $O/r3o_soak_wide_shapes.log 2>&1; echo "soak rc=$?" >> $O/r3o_soak_wide_shapes.log
grep -E "FAILED|done|rc=" $O/r3o_soak_wide_shapes.log | cut -c1-500 | tail -30
