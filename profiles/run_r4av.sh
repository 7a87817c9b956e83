#!/bin/bash
# Round 4, GPU call AV: does MC_size 3 (two workgroups per CU) want other rows per chunk than MC_size 1 at Nc = 50k?
# (the bench line of call r4au: MC_size 3 at 0.820 with 196 rows where 256 rows gave 0.843 - 0.854)
set -x
O=gpurun_out
timeout 600 python profiles/rpc_sweep.py --config c3 --mc 3 --rpc 196,256,391,196,256 --reps 2 --steps 12 --out $O/r4av_rpc_c3_mc3.json > $O/r4av_rpc_c3_mc3.log 2>&1
tail -6 $O/r4av_rpc_c3_mc3.log | cut -c1-220
timeout 600 python profiles/rpc_sweep.py --config c3 --mc 3 --shard-of 8 --rpc 196,256,391 --reps 3 --steps 30 --out $O/r4av_rpc_c3_of8_mc3.json > $O/r4av_rpc_c3_of8_mc3.log 2>&1
tail -4 $O/r4av_rpc_c3_of8_mc3.log | cut -c1-220
