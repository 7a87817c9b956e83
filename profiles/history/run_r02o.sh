cd $GRAFT_REPO_ROOT; O=gpurun_out/r02p; mkdir -p $O
python bench.py --mc 3 --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg --no-e2e 2>/dev/null | tail -1 > $O/bench_mc3_before.json
python profiles/e2e_lrt_c3.py --min-iter 5000 --max-iter 20000 --mc 3 --common-noise --verbose > $O/lrt.log 2> $O/err.log
python bench.py --mc 3 --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg --no-e2e 2>/dev/null | tail -1 > $O/bench_mc3_after.json
grep "BRIE2" $O/lrt.log; tail -1 $O/lrt.log | cut -c1-330
python -c "
import json
for f in ('before','after'):
    d=json.load(open('gpurun_out/r02p/bench_mc3_%s.json'%f)); print(f, d['ms_per_step'])"
