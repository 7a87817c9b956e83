#!/bin/bash
# round 3, GPU call k: final tree -- suite, smoke, bench lines of configs[2] (headline), configs[4] whole on one GPU, configs[1]
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r3k_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r3k_pytest.log
grep -E "passed|failed|^FAILED|Error" $O/r3k_pytest.log | tail -5
python -c "import __graft_entry__ as g; g.smoke()" > $O/r3k_smoke.log 2>&1; tail -1 $O/r3k_smoke.log
python bench.py > $O/r3k_bench_c3.json 2> $O/r3k_bench_c3.err
python bench.py --config c5 > $O/r3k_bench_c5_whole.json 2> $O/r3k_bench_c5.err
python bench.py --config c2 > $O/r3k_bench_c2.json 2> $O/r3k_bench_c2.err
python - <<'PY'
import json
for f in ("r3k_bench_c3", "r3k_bench_c5_whole", "r3k_bench_c2"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    p = d.get("pcie_inclusive", {})
    print(f, "ms/step %.4f (unprofiled %.4f) frac %.4f traffic %s" % (d["ms_per_step"], d["ms_per_step_without_profiling_events"], d["roofline"]["frac"], d["roofline"].get("traffic")),
          "| e2e", round(p.get("total_s", 0), 3), json.dumps({k: round(v, 3) for k, v in p.get("breakdown_s", {}).items() if isinstance(v, float)}))
PY
