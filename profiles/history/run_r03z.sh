cd $GRAFT_REPO_ROOT; O=gpurun_out/r03z; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -n 2
timeout 900 python bench.py > $O/bench_c3.json 2> $O/bench_err.log; tail -n 1 $O/bench_c3.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['frac'], r['traffic'], r['avg_kernel_ms'], d['cpu_baseline']['value'], d['pcie_inclusive']['total_s'])"
bash profiles/run_profile.sh r03z > $O/profile.log 2>&1; head -n 6 gpurun_out/prof_r03z/summary.txt
timeout 600 python bench.py --mc 3 --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg --no-e2e > $O/bench_mc3.json 2>>$O/bench_err.log; tail -n 1 $O/bench_mc3.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mc3', d['ms_per_step'])"
timeout 600 python bench.py --config c2 --no-cpu-baseline --no-psi-check --no-pmc --no-e2e > $O/bench_c2.json 2>>$O/bench_err.log; tail -n 1 $O/bench_c2.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2', d['ms_per_step'], d['roofline']['frac'])"
timeout 900 python bench.py --config c5 --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg --no-e2e > $O/bench_c5.json 2>>$O/bench_err.log; tail -n 1 $O/bench_c5.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5', d['ms_per_step'], d['roofline']['frac'])"
