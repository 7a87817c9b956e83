#!/bin/bash
# Round 4, GPU call H: HIP vs the fp32 oracle for the cases of profiles/psi_null.py (the o32 caches of these cases travel)
set -x
timeout 3000 python profiles/psi_null.py --hip --cases $1 > gpurun_out/r4h_psi_null_hip_$2.log 2>&1
tail -30 gpurun_out/r4h_psi_null_hip_$2.log
