#!/bin/bash
# MC_size 3 at configs[1] under the occupancy caps of the step launch (BRIE_STEP_OCCUPANCY_CAP: 1 / 2 workgroups per CU, 0 = hardware)
for cap in 2 1 0 2 1 0; do
  BRIE_STEP_OCCUPANCY_CAP=$cap timeout 200 python bench.py --config c2 --mc 3 --no-pmc --no-e2e --no-cpu-baseline --no-psi-check 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; m=r.get('mc1',{})
print('cap $cap mc3', round(r['avg_kernel_ms'],4), round(r['frac'],4), 'mc1', m.get('avg_kernel_ms'), m.get('frac'), r['placement']['GBs'])"
done
