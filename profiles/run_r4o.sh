#!/bin/bash
# Round 4, GPU call O: candidate sets allocated interleaved (array by array) vs one block after the other, alternating fresh processes
set -x
O=gpurun_out
TAG=${1:-r4o}
export BRIE_PLACEMENT_LOG=1
for i in 1 2 3 4 5 6; do
  for il in 1 0; do
    BRIE_PLACEMENT_INTERLEAVE=$il timeout 200 python profiles/placement_ab.py --config c3 --handles 1 --tries 4 --out $O/${TAG}_interleave_${il}_c3.jsonl > /dev/null 2>> $O/${TAG}_interleave_${il}_c3.err
  done
done
for il in 1 0; do
  grep -h "brie placement" $O/${TAG}_interleave_${il}_c3.err | awk '{print $4, $5}' | tr '\n' ' '; echo " <- interleave $il"
  python - <<PY
import json
for l in open("$O/${TAG}_interleave_${il}_c3.jsonl"):
    d = json.loads(l)
    if "summary" not in d: print(d["tune"]["seconds"], d["tune"]["GBs"], d["tune"]["kept"], d["step_ms_before"], d["step_ms_after"])
PY
done
for i in 1 2 3 4; do
  for il in 1 0; do
    BRIE_PLACEMENT_INTERLEAVE=$il timeout 100 python profiles/placement_ab.py --config c2 --handles 2 --tries 4 --out $O/${TAG}_interleave_${il}_c2.jsonl > /dev/null 2>> $O/${TAG}_interleave_${il}_c2.err
  done
done
for il in 1 0; do
  grep -h "brie placement" $O/${TAG}_interleave_${il}_c2.err | awk '{print $4, $5}' | tr '\n' ' '; echo " <- c2 interleave $il"
done
