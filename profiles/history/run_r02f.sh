cd $GRAFT_REPO_ROOT; O=gpurun_out/r02f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_distributed.py tests/test_gpu_comm.py -q -m gpu -k "wide or coupled or randomised or marginlik or in_library or distributed or sharded" -x > $O/pytest_tile.log 2>&1
tail -5 $O/pytest_tile.log
for kc in 3 16 32 64; do
  timeout 300 python bench.py --kc $kc --steps 10 --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg --no-e2e > $O/bench_kc${kc}_tile.json 2>> $O/err.log
  BRIE_WIDE_PATH=lds timeout 300 python bench.py --kc $kc --steps 10 --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg --no-e2e > $O/bench_kc${kc}_lds.json 2>> $O/err.log
done
timeout 900 python profiles/coupled_bench.py --steps 8 > $O/coupled_tile.log 2>> $O/err.log
BRIE_WIDE_PATH=lds timeout 900 python profiles/coupled_bench.py --steps 8 > $O/coupled_lds.log 2>> $O/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02f/bench_kc*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['ms_per_step'],3), d['roofline']['storage_bytes_per_launch'])
    except Exception as e: print(f, 'ERR', e)
PY
tail -1 $O/coupled_tile.log; tail -1 $O/coupled_lds.log; tail -5 $O/err.log
