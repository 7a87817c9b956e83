#!/bin/bash
# Round 4, GPU call AS: candidates of a configs[1] handle allocated one at a time (lazy = 1) against interleaved up front
# (lazy = 0), alternating fresh processes on one box (call r4ar: eight lazy processes in a row found no fast set).
# BRIE_PLACEMENT_LAZY existed for this call only (no difference: the switch is gone, small handles allocate lazily).
O=gpurun_out
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  for lazy in 0 1; do
    BRIE_PLACEMENT_LAZY=$lazy timeout 100 python profiles/placement_auto.py --config c2 >> $O/r4as_lazy_${lazy}_c2.jsonl 2>> $O/r4as_lazy_c2.err
  done
done
for lazy in 0 1; do echo "lazy $lazy"; cut -c30-200 $O/r4as_lazy_${lazy}_c2.jsonl; done
