cd $GRAFT_REPO_ROOT; O=gpurun_out/r02x; mkdir -p $O
( time python -m pytest tests -q -m gpu ) > $O/pytest_all.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
bash profiles/run_profile.sh r02x > $O/prof.log 2>&1
grep -E "passed|failed" $O/pytest_all.log; grep -E "^FAILED" $O/pytest_all.log | head; tail -n 1 $O/smoke.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02x/bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(round(d["ms_per_step"],3), round(r["avg_kernel_ms"],3), round(r["frac"],3), r["count_storage"], r.get("traffic"), r["measured_stream_ceiling_GBs"], d["pcie_inclusive"]["total_s"])
PY
head -n 4 gpurun_out/prof_r02x/summary.txt
