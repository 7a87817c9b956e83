cd $GRAFT_REPO_ROOT; O=gpurun_out/r02r; mkdir -p $O
python -m pytest tests/test_c_abi.py -q -m gpu > $O/pytest.log 2>&1; tail -1 $O/pytest.log
for i in 1 2 3; do
  python bench.py --config c2 --steps 200 --warmup 20 --no-pmc --no-cpu-baseline --no-psi-check --no-f32-leg --no-e2e 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tiered', d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['count_storage'])"
  BRIE_COUNT_TIERS=uniform python bench.py --config c2 --steps 200 --warmup 20 --no-pmc --no-cpu-baseline --no-psi-check --no-f32-leg --no-e2e 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('uniform', d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['count_storage'])"
done
