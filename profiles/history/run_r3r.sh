#!/bin/bash
# round 3, GPU call r: the frozen PSI parity rule on the remaining HELD-OUT cases (brie-quant schedule with other seeds;
# a shape no config has with a third set of seeds) and one more soak of the random test families with fresh seeds
O=gpurun_out
mkdir -p $O
python profiles/psi_delta.py --cases c2_cli_128_s2,c3_cli_128_s2,mid_api_256_s3,mid_cli_96_s3 --out $O/r3r_psi_delta_heldout.json > $O/r3r_psi_delta_heldout.log 2>&1
grep -v amdgpu.ids $O/r3r_psi_delta_heldout.log | tail -6
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3r_psi_delta_heldout.json"))
for k, c in d["cases"].items():
    print(k, json.dumps(c["parity_rule"])[:1200])
PY
timeout 1500 python tests/tools/soak_randomised.py 300 300 20261002 400 400 > $O/r3r_soak_fresh_seeds.log 2>&1; echo "soak rc=$?" >> $O/r3r_soak_fresh_seeds.log
grep -E "FAILED|done|rc=" $O/r3r_soak_fresh_seeds.log | cut -c1-500 | tail -20
