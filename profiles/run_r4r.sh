#!/bin/bash
# Round 4, GPU call R: one packed set moved through a 240-GB slab: where are the slow regions?  (headline shape, then the configs[1] shape)
set -x
O=gpurun_out
for i in 1 2 3; do
  timeout 300 python profiles/layout_probe.py --scan 240 --scan-step 8 --rounds 1 >> $O/r4r_region_scan_c3.jsonl 2>> $O/r4r_region_scan.err
  timeout 300 python profiles/layout_probe.py --nc 10000 --ng 5000 --scan 240 --scan-step 4 --rounds 1 >> $O/r4r_region_scan_c2.jsonl 2>> $O/r4r_region_scan.err
done
python - <<'PY'
import json
for f in ("gpurun_out/r4r_region_scan_c3.jsonl", "gpurun_out/r4r_region_scan_c2.jsonl"):
    print(f)
    for l in open(f):
        d = json.loads(l)
        print(d["pid"], " ".join("%d" % round(v / 100) for k, v in sorted(d["GBs"].items())))
PY
