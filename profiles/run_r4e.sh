#!/bin/bash
# Round 4, GPU call E: the placement search over FRESH processes (>= 8 processes on >= 2 boxes):
# per process one configs[2] handle: first placement vs the kept one of three, step ms of both; then configs[1]
set -x
O=gpurun_out
TAG=${1:-r4e}
mkdir -p $O
for i in 1 2 3 4 5 6 7 8; do
  timeout 200 python profiles/placement_ab.py --config c3 --handles 1 --out $O/${TAG}_placement_procs_c3.jsonl > /dev/null 2>> $O/${TAG}_placement_procs.err
done
for i in 1 2 3 4 5 6 7 8; do
  timeout 100 python profiles/placement_ab.py --config c2 --handles 2 --out $O/${TAG}_placement_procs_c2.jsonl > /dev/null 2>> $O/${TAG}_placement_procs.err
done
grep -v summary $O/${TAG}_placement_procs_c3.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['pid'], d['probe_before_GBs'], d['step_ms_before'], d['tune']['GBs'], d['tune']['kept'], d['step_ms_after'], d['tune']['seconds'])"
