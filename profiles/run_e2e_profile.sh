#!/bin/bash
# rocprofv3 --kernel-trace --stats of one (shortened) reference-default fit at C3: every kernel of the path with its
# average duration (profiles/e2e_fit.py: upload, 6 x 20 steps, 500-draw loss_gene, read-back of 4 matrices)
set -u
TAG=${1:-r01e_e2e}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
RAW=/tmp/brie_prof_$TAG
rm -rf $RAW; mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- python3 $R/profiles/e2e_fit.py --min-iter 120 > $OUT/e2e.log 2>&1
find $RAW -name "*kernel_stats.csv" -exec cp {} $OUT/ \;
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*kernel_stats.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "brie::" in r["Name"]]
print("%-90s %6s %12s" % ("kernel", "calls", "avg_ms"))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    print("%-90s %6s %12.3f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
tail -1 $OUT/e2e.log
