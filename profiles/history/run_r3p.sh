#!/bin/bash
# round 3, GPU call p: the fitBRIE default path (per-batch convergence) at configs[2] with the round's library, per-round breakdown
O=gpurun_out
mkdir -p $O
python profiles/e2e_batchconv.py > $O/r3p_e2e_batchconv_c3.log 2>&1
tail -1 $O/r3p_e2e_batchconv_c3.log > $O/r3p_e2e_batchconv_c3.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3p_e2e_batchconv_c3.json").read())
print("fit_s", d["fit_s"], "mean n_iter", d["mean_n_iter"], "steps_run", d["steps_run"])
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in d["timing"].items() if k != "stage_s"})
for r in d["rounds"]:
    print(r)
PY
