#!/bin/bash
# round 3, GPU call i: the export order A/B inside whole fits (the A/B of the export alone showed no difference, the bench
# line did); the self-launched 2-rank dry run with the final bench.py
O=gpurun_out
mkdir -p $O
python profiles/e2e_fit_api.py --ab-export 8 > $O/r3i_e2e_export_ab.log 2>&1
grep export $O/r3i_e2e_export_ab.log
BRIE_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --config c2 > $O/r3i_bench_c2_n2_self_launched_gloo_one_gpu.json 2> $O/r3i_bench_c2_n2.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3i_bench_c2_n2_self_launched_gloo_one_gpu.json").read().strip().splitlines()[-1])
print("n_gpus", d["n_gpus"], "keys", sorted(d), "traffic", d["roofline"].get("traffic"), d["roofline"].get("traffic_source", "")[:60], "bit_identical", d["allgather"]["recomputed_on_rank0"])
PY
