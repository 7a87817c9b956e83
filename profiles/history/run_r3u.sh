#!/bin/bash
# round 3, GPU call u: the HIP run of the one gene that made revision 1 of the parity rule fail (mid_cli_96_s3 gene 3:
# 98 cells beyond 1e-4 with no end-of-fit parameter shift) followed along the fit, next to the fp32 oracle's
mkdir -p gpurun_out
timeout 400 python profiles/cluster_trajectory.py mid_cli_96_s3 3 49 --hip > gpurun_out/r3u_cluster_trajectory_hip.log 2>&1
grep -v amdgpu.ids gpurun_out/r3u_cluster_trajectory_hip.log | tail -30
timeout 400 python profiles/cluster_trajectory.py mid_cli_96_s3 35 49 --hip >> gpurun_out/r3u_cluster_trajectory_hip.log 2>&1
grep -v amdgpu.ids gpurun_out/r3u_cluster_trajectory_hip.log | grep "gene 35"
