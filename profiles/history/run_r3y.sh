#!/bin/bash
# round 3, GPU call y: the final tree (model-variant PSI test, snapshot guard) -- the final tree -- suite (rule revision 2, full-size short-horizon rule, held-out case), smoke,
# default bench line, self-launched 2-rank line
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r3y_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r3y_pytest.log
grep -E "passed|failed|^FAILED|Error" $O/r3y_pytest.log | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/r3y_bench_c3.json 2> $O/r3y_bench_c3.err
BRIE_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --config c2 --no-pmc > $O/r3y_bench_c2_n2.json 2> $O/r3y_bench_c2_n2.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3y_bench_c3.json").read().strip().splitlines()[-1]); p = d["pcie_inclusive"]
print("c3 ms/step %.3f frac %.4f traffic %.4g | e2e %.3f" % (d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], p["total_s"]), {k: round(v, 3) for k, v in p["breakdown_s"].items() if isinstance(v, float)})
print("psi check:", {k: d["psi_delta_vs_cpu_ref"].get(k) for k in ("displaced_genes", "clustered_genes", "rule")})
d = json.loads(open("gpurun_out/r3y_bench_c2_n2.json").read().strip().splitlines()[-1])
print("n2:", d["n_gpus"], d["allgather"]["recomputed_on_rank0"], "cpu_baseline" in d)
PY
