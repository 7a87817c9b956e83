#!/bin/bash
# round 3, GPU call g: staged result export (suite + A/B), the brie-quant schedule on the 128-gene configs[2] sample
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r3g_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r3g_pytest.log
grep -E "passed|failed|^FAILED|Error" $O/r3g_pytest.log | tail -8
python profiles/egress_ab.py --threads 2,4 --out $O/r3g_egress_ab.json > $O/r3g_egress_ab.log 2>&1
tail -9 $O/r3g_egress_ab.log | cut -c1-400
python profiles/psi_delta.py --cases c3_cli_128 --out $O/r3g_psi_delta_c3_cli.json > $O/r3g_psi_delta_c3_cli.log 2>&1
tail -3 $O/r3g_psi_delta_c3_cli.log
python bench.py > $O/r3g_bench_c3.json 2> $O/r3g_bench_c3.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3g_bench_c3.json").read().strip().splitlines()[-1])
print("c3: ms/step %.3f frac %.4f" % (d["ms_per_step"], d["roofline"]["frac"]), json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["pcie_inclusive"]["breakdown_s"].items() if k != "stage_s"}), round(d["pcie_inclusive"]["total_s"], 3))
PY
