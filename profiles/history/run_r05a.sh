cd $GRAFT_REPO_ROOT; O=gpurun_out/r05a; mkdir -p $O
timeout 900 python bench.py > $O/bench_c3.json 2> $O/bench_err.log; tail -n 1 $O/bench_c3.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['frac'], r['traffic'], r['avg_kernel_ms'], d['cpu_baseline']['value'], d['pcie_inclusive']['total_s'], d['pcie_inclusive']['breakdown_s']['loss_gene_s'], d['pcie_inclusive']['breakdown_s']['read_wait_s'])"
bash profiles/run_profile.sh r05a > $O/profile.log 2>&1; head -n 4 gpurun_out/prof_r05a/summary.txt
