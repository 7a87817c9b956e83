cd $GRAFT_REPO_ROOT; O=gpurun_out/r03c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mixed or tier or pack or frozen or margin or wide or coupled" > $O/pytest_sel.log 2>&1; tail -n 4 $O/pytest_sel.log
timeout 600 python bench.py > $O/bench_c3.json 2> $O/bench_err.log; tail -n 1 $O/bench_c3.json | cut -c1-1500
BRIE_COUNT_TIERS=uniform timeout 300 python bench.py --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg --no-e2e > $O/bench_c3_uniform.json 2>>$O/bench_err.log; tail -n 1 $O/bench_c3_uniform.json | cut -c1-400
timeout 300 python bench.py --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg --no-e2e > $O/bench_c3_b.json 2>>$O/bench_err.log; tail -n 1 $O/bench_c3_b.json | cut -c1-400
