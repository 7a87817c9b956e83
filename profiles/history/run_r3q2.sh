#!/bin/bash
# r3q2: the frozen PSI rule on the first two HELD-OUT cases (API schedule), and the tightened full-size file
O=gpurun_out
mkdir -p $O
python profiles/psi_delta.py --cases c2_api_512_s2,c3_api_512_s2 --out $O/r3q2_psi_delta_heldout_api.json > $O/r3q2_psi_delta_heldout_api.log 2>&1
tail -4 $O/r3q2_psi_delta_heldout_api.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3q2_psi_delta_heldout_api.json"))
for k, c in d["cases"].items():
    print(k, json.dumps(c["parity_rule"])[:1500])
PY
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -p no:cacheprovider > $O/r3q2_fullsize.log 2>&1; echo "rc $?" >> $O/r3q2_fullsize.log
tail -5 $O/r3q2_fullsize.log
