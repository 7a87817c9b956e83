#!/bin/bash
# round 3, GPU call w: rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes of the headline bench workload with the
# round's final library, and the bench line of the same box next to it
O=gpurun_out
mkdir -p $O
bash profiles/run_profile.sh r3w > $O/r3w_profile.log 2>&1
tail -25 $O/r3w_profile.log
python bench.py --no-e2e --no-cpu-baseline --no-psi-check > $O/r3w_bench_c3.json 2> $O/r3w_bench_c3.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3w_bench_c3.json").read().strip().splitlines()[-1])
print("c3 ms/step %.3f avg_kernel_ms %.3f frac %.4f traffic %.4g" % (d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"], d["roofline"]["traffic"]))
PY
