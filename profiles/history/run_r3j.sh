#!/bin/bash
# round 3, GPU call j: workgroups per CU on configs[1] (effLen), on an 8-way shard of configs[2] and at MC_size 3, one handle each
O=gpurun_out
mkdir -p $O
for spec in "--config c2" "--config c2 --mc 3" "--config c3 --shard-of 8" "--config c3 --shard-of 8 --mc 3" "--config c3 --steps 12 --rounds 5"; do
  python profiles/occ_ab2.py $spec 2>/dev/null | tail -1 >> $O/r3j_occupancy_ab.log
done
cat $O/r3j_occupancy_ab.log
