cd $GRAFT_REPO_ROOT; O=gpurun_out/r04l; mkdir -p $O
python profiles/occ_ab.py 2>/dev/null | tee $O/occ_ab.log
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -n 2
for cfg in c3 c2 c5; do for cap in 1 2; do
BRIE_STEP_OCCUPANCY_CAP=$cap timeout 900 python bench.py --config $cfg --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$cfg cap $cap', d['ms_per_step'], d['roofline']['frac'])"
done; done | tee $O/bench_caps.log
