cd $GRAFT_REPO_ROOT; O=gpurun_out/r03h; mkdir -p $O
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg > $O/bench_e2e_$i.json 2>$O/err.log; tail -n 1 $O/bench_e2e_$i.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['pcie_inclusive']['total_s'], d['pcie_inclusive']['breakdown_s'])"
done
nproc; free -g | head -n 2
