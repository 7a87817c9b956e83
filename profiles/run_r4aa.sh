#!/bin/bash
# (BRIE_PLACEMENT_PADS existed for this call only: dummy sets widening the pitch of the interleaved candidates changed nothing)
# Round 4, GPU call AA: does a wider pitch of the interleaved candidates (dummy sets taking part in the interleaving) make them faster?
set -x
O=gpurun_out
export BRIE_PLACEMENT_LOG=1
for i in 1 2 3 4 5; do
  for pads in 0 2 4; do
    BRIE_PLACEMENT_PADS=$pads timeout 200 python profiles/placement_ab.py --config c3 --handles 1 --tries 4 --out $O/r4aa_pads_${pads}_c3.jsonl > /dev/null 2>> $O/r4aa_pads_${pads}_c3.err
  done
done
for pads in 0 2 4; do
  grep -h "brie placement" $O/r4aa_pads_${pads}_c3.err | awk '{print $4, $5}' | tr '\n' ' '; echo " <- pads $pads"
  python - <<PY
import json
print([json.loads(l)["tune"]["seconds"] for l in open("$O/r4aa_pads_${pads}_c3.jsonl") if "summary" not in l])
PY
done
