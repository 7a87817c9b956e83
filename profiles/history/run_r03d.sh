cd $GRAFT_REPO_ROOT; O=gpurun_out/r03d; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -n 6 $O/pytest_gpu.log
for i in 1 2 3; do
timeout 300 python bench.py --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg --no-e2e > $O/bench_c3_q$i.json 2>>$O/bench_err.log; tail -n 1 $O/bench_c3_q$i.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('quad', d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['storage_bytes_per_launch'])"
BRIE_COUNT_TIERS=uniform timeout 300 python bench.py --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg --no-e2e > $O/bench_c3_u$i.json 2>>$O/bench_err.log; tail -n 1 $O/bench_c3_u$i.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('uniform', d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['storage_bytes_per_launch'])"
done
bash profiles/run_profile.sh r03d > $O/profile.log 2>&1; tail -n 25 $O/profile.log
