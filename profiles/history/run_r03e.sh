cd $GRAFT_REPO_ROOT; O=gpurun_out/r03e; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -Wno-unused-result profiles/micro/fast_log.hip -o /tmp/fast_log 2>/dev/null && /tmp/fast_log | tee $O/fast_log.log
bash profiles/ab.sh 3 "--no-pmc --no-f32-leg --no-e2e" base w3 2>&1 | tee $O/ab_w3.log
timeout 600 python bench.py --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg > $O/bench_e2e.json 2>$O/err.log; tail -n 1 $O/bench_e2e.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['pcie_inclusive'])"
timeout 600 python bench.py --mc 3 --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg --no-e2e > $O/bench_mc3.json 2>>$O/err.log; tail -n 1 $O/bench_mc3.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mc3', d['ms_per_step'])"
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -n 3 $O/pytest_gpu.log
