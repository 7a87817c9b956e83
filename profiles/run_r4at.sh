#!/bin/bash
# Round 4, GPU call AT: rows_per_chunk re-swept with the current kernel (the rule in configure_tiling dates from round 1)
# at configs[1], configs[2] and an 8-way shard of configs[2], MC_size 1 and 3
set -x
O=gpurun_out
timeout 300 python profiles/rpc_sweep.py --config c2 --rpc 32,50,64,79,100,128,157,200,256,313,400 --reps 3 --out $O/r4at_rpc_c2.json > $O/r4at_rpc_c2.log 2>&1
tail -14 $O/r4at_rpc_c2.log
timeout 300 python profiles/rpc_sweep.py --config c2 --mc 3 --rpc 64,128,200,256,400 --reps 2 --out $O/r4at_rpc_c2_mc3.json > $O/r4at_rpc_c2_mc3.log 2>&1
tail -7 $O/r4at_rpc_c2_mc3.log
timeout 600 python profiles/rpc_sweep.py --config c3 --rpc 128,196,256,391,512,782 --reps 2 --steps 12 --out $O/r4at_rpc_c3.json > $O/r4at_rpc_c3.log 2>&1
tail -8 $O/r4at_rpc_c3.log
timeout 600 python profiles/rpc_sweep.py --config c3 --shard-of 8 --rpc 128,196,256,391,512,782 --reps 3 --steps 30 --out $O/r4at_rpc_c3_of8.json > $O/r4at_rpc_c3_of8.log 2>&1
tail -8 $O/r4at_rpc_c3_of8.log
