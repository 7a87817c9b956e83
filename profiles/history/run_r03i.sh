cd $GRAFT_REPO_ROOT; O=gpurun_out/r03i; mkdir -p $O
python profiles/e2e_fit_api.py > $O/e2e_prefetch.json 2>$O/err.log; tail -n 1 $O/e2e_prefetch.json
python profiles/e2e_fit_api.py --no-prefetch > $O/e2e_noprefetch.json 2>>$O/err.log; tail -n 1 $O/e2e_noprefetch.json
python bench.py --steps 996 --warmup 3 --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg --no-e2e > $O/bench_996.json 2>>$O/err.log; tail -n 1 $O/bench_996.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('996 steps', d['ms_per_step'], d['roofline']['avg_kernel_ms'])"
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg --no-e2e > $O/bench_20.json 2>>$O/err.log; tail -n 1 $O/bench_20.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('20 steps', d['ms_per_step'], d['roofline']['avg_kernel_ms'])"
