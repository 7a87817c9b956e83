cd $GRAFT_REPO_ROOT; O=gpurun_out/r03a; mkdir -p $O
A="--no-pmc --no-cpu-baseline --no-psi-check --no-f32-leg --no-e2e"
for rep in 1 2; do for rpc in 32 48 64 96 128; do
  python bench.py --config c2 --steps 200 --warmup 20 --rows-per-chunk $rpc $A 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 rpc $rpc', round(d['ms_per_step'],4), round(d['roofline']['avg_kernel_ms'],4))" >> $O/rpc.log
done; done
for rep in 1 2; do for rpc in 128 192 256 384; do
  python bench.py --steps 12 --rows-per-chunk $rpc $A 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 rpc $rpc', round(d['ms_per_step'],4), round(d['roofline']['avg_kernel_ms'],4))" >> $O/rpc.log
done; done
cat $O/rpc.log
