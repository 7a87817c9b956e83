#!/bin/bash
# round 3, GPU call s (three parts, the oracle caches of all cases do not fit one 512-MiB snapshot): revision 2 of the PSI
# parity rule evaluated on every case -- part 1: the seven cases revision 1 was frozen on; part 2: the first held-out set;
# part 3: the SECOND held-out set, generated after revision 2
O=gpurun_out
mkdir -p $O
PART=$1; CASES=$2
python profiles/psi_delta.py --cases "$CASES" --out $O/r3s_psi_delta_rev2_part$PART.json > $O/r3s_psi_delta_rev2_part$PART.log 2>&1
grep -v amdgpu.ids $O/r3s_psi_delta_rev2_part$PART.log | tail -8
python - $PART <<'PY'
import json, sys
d = json.load(open("gpurun_out/r3s_psi_delta_rev2_part%s.json" % sys.argv[1]))
for k, c in d["cases"].items():
    r = c["parity_rule"]
    print(k, "holds" if r.get("holds") else "VIOLATED " + r.get("violated", ""), {a: r[a] for a in ("displaced_genes", "clustered_genes") if a in r},
          {a: r["quiet_genes"][a] for a in ("genes", "gt_1e-4", "p99")} if "quiet_genes" in r else "")
PY
