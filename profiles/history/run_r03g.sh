cd $GRAFT_REPO_ROOT; O=gpurun_out/r03g; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -n 2
timeout 600 python bench.py --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg > $O/bench_e2e.json 2>$O/err.log; tail -n 1 $O/bench_e2e.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['pcie_inclusive'])"
timeout 600 python bench.py --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg > $O/bench_e2e_b.json 2>$O/err.log; tail -n 1 $O/bench_e2e_b.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['pcie_inclusive'])"
