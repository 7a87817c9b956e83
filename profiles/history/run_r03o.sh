cd $GRAFT_REPO_ROOT; O=gpurun_out/r03o; mkdir -p $O
for i in 1 2; do python profiles/occ_ab.py 2>/dev/null | tr '\n' ' '; echo; python profiles/occ_ab.py uncapped 2>/dev/null | tr '\n' ' '; echo; done | tee $O/occ_ab.log
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -n 2
timeout 600 python bench.py --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg --no-e2e > $O/bench_c3.json 2>$O/err.log; tail -n 1 $O/bench_c3.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3', d['ms_per_step'], d['roofline']['frac'])"
timeout 600 python bench.py --config c2 --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg --no-e2e > $O/bench_c2.json 2>$O/err.log; tail -n 1 $O/bench_c2.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2', d['ms_per_step'], d['roofline']['frac'], d['roofline']['count_storage'])"
BRIE_STEP_OCCUPANCY_CAP=0 timeout 600 python bench.py --config c2 --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg --no-e2e > $O/bench_c2_uncapped.json 2>$O/err.log; tail -n 1 $O/bench_c2_uncapped.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 uncapped', d['ms_per_step'], d['roofline']['frac'])"
python profiles/wide_ab.py --rounds 3 --cases 16:0,32:0,64:0,3:8,3:16,3:32,3:64,32:32,64:32 2>/dev/null | tee $O/wide_ab.log | head -n 9
