#!/bin/bash
# round 3, GPU call h: result export with the kernel of slab k + 1 ahead of the copies of slab k (two streams) against
# round 2's single-stream order; suite; bench
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r3h_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r3h_pytest.log
grep -E "passed|failed|^FAILED|Error" $O/r3h_pytest.log | tail -8
python profiles/egress_ab.py --modes one_stream,two_streams --reps 3 --out $O/r3h_egress_ab.json > $O/r3h_egress_ab.log 2>&1
tail -7 $O/r3h_egress_ab.log | cut -c1-400
python bench.py > $O/r3h_bench_c3.json 2> $O/r3h_bench_c3.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3h_bench_c3.json").read().strip().splitlines()[-1])
print("c3: ms/step %.3f (unprofiled %.3f) frac %.4f" % (d["ms_per_step"], d["ms_per_step_without_profiling_events"], d["roofline"]["frac"]), json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["pcie_inclusive"]["breakdown_s"].items() if k != "stage_s"}), round(d["pcie_inclusive"]["total_s"], 3))
PY
python bench.py --config c1 --steps 400 --warmup 20 --no-pmc > $O/r3h_bench_c1.json 2> $O/r3h_bench_c1.err
python -c "
import json
d=json.loads(open('gpurun_out/r3h_bench_c1.json').read().strip().splitlines()[-1]); print('c1 us/step', d['ms_per_step']*1e3, 'unprofiled', d['ms_per_step_without_profiling_events']*1e3)"
