#!/bin/bash
# (BRIE_CREATE_ORDER existed in the library for calls r4s / r4t only: the A/B showed no reliable gain and the switch was removed)
# Round 4, GPU call S: first placement of a fresh process with the state arrays created in two groups around the count
# layers ("split", the new order) against six in a row ("packed", rounds 1 - 3), alternating; no search (tries 1)
set -x
O=gpurun_out
TAG=${1:-r4s}
export BRIE_PLACEMENT_LOG=1
for i in 1 2 3 4 5 6 7 8; do
  for ord in split packed; do
    BRIE_CREATE_ORDER=$ord timeout 200 python profiles/placement_ab.py --config c3 --handles 1 --tries 1 --out $O/${TAG}_create_${ord}_c3.jsonl > /dev/null 2>> $O/${TAG}_create_${ord}_c3.err
  done
done
for ord in split packed; do
  grep -h "brie placement" $O/${TAG}_create_${ord}_c3.err | awk '{print $5}' | tr '\n' ' '; echo " <- first placement GB/s, create order $ord"
done
