cd $GRAFT_REPO_ROOT; O=gpurun_out/r04a; mkdir -p $O
X="--no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg --no-e2e --steps 20 --warmup 3"
for i in 1 2 3 4; do for rpc in 256 258 250 262 244; do
python bench.py $X --rows-per-chunk $rpc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('rpc $rpc round $i ms %.3f kernel %.3f' % (d['ms_per_step'], d['roofline']['avg_kernel_ms']))"
done; done | tee $O/rpc_tail.log
