cd $GRAFT_REPO_ROOT; O=gpurun_out/r02m; mkdir -p $O
( time python -m pytest tests -q -m gpu ) > $O/pytest_all.log 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
BRIE_COUNT_TIERS=uniform python bench.py --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg --no-e2e > $O/bench_uniform.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg --no-e2e > $O/bench2.json 2>> $O/bench.err
BRIE_COUNT_TIERS=uniform python bench.py --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg --no-e2e > $O/bench_uniform2.json 2>> $O/bench.err
BRIE_BENCH_SINGLE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --config c2 --steps 10 --warmup 2 > $O/bench_n2.json 2> $O/bench_n2.err
bash profiles/run_profile.sh r02m > $O/prof.log 2>&1
grep -E "passed|failed" $O/pytest_all.log; grep -E "^FAILED" $O/pytest_all.log | head
python - <<'PY'
import json
for f in ("bench","bench_uniform","bench2","bench_uniform2"):
    d=json.loads(open("gpurun_out/r02m/%s.json"%f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f, round(d["ms_per_step"],3), round(r["avg_kernel_ms"],3), r["count_storage"], r.get("traffic"))
d=json.loads(open("gpurun_out/r02m/bench_n2.json").read().strip().splitlines()[-1]); print(d["allgather"])
PY
head -6 gpurun_out/prof_r02m/summary.txt
