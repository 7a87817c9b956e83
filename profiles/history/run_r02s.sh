cd $GRAFT_REPO_ROOT; O=gpurun_out/r02s; mkdir -p $O
python profiles/e2e_batchconv.py 2> $O/err.log | tail -1 > $O/batchconv.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02s/batchconv.json"))
print(d["fit_s"], d["steps_run"], d["mean_n_iter"], d["timing"])
for r in d["rounds"]: print({k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items()})
PY
tail -2 $O/err.log
