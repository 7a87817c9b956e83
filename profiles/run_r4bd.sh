#!/bin/bash
# Round 4, GPU call BD (two parts): HIP vs the fp32 oracle once more for the twenty cases of profiles/psi_null.py, with the
# library as it stands (the rows per chunk of calls r4at ff. change the order of the fp32 partial sums at Nc >= 6144)
set -x
timeout 3000 python profiles/psi_null.py --hip --cases $1 > gpurun_out/r4bd_psi_null_hip_$2.log 2>&1
tail -22 gpurun_out/r4bd_psi_null_hip_$2.log
