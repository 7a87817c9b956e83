#!/bin/bash
# One extra rocprofv3 --pmc pass with SQ counters for the dominant kernel (occupancy / issue mix).
#   bash profiles/run_sq_counters.sh <tag> [bench args]
set -u
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
RAW=/tmp/brie_sq_$TAG
rm -rf $RAW; mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-psi-check $*"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE \
  --output-format csv -d $RAW/sq -o sq -- python3 $R/bench.py $ARGS > $OUT/bench_sq.log 2>&1
python3 - <<PY > $OUT/sq_summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("$RAW/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "elbo_adam_step" in r["Kernel_Name"]:
            k = r["Counter_Name"]; agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
print("brie::elbo_adam_step -- per dispatch (rocprofv3 --pmc, one pass)")
for k, (n, v) in sorted(agg.items()):
    print("%-22s %14.6g   (dispatches %d)" % (k, v / n, n))
d = {k: v / n for k, (n, v) in agg.items()}
if "SQ_WAVE_CYCLES" in d and "SQ_ACTIVE_INST_VALU" in d:
    print("VALU active share of wave cycles : %.3f" % (d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"]))
if "SQ_WAVE_CYCLES" in d and "SQ_WAIT_ANY" in d:
    print("waves parked (SQ_WAIT_ANY)       : %.3f" % (d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]))
if "SQ_WAVES" in d and "SQ_INSTS_VALU" in d:
    print("VALU instructions per wave       : %.1f" % (d["SQ_INSTS_VALU"] / d["SQ_WAVES"]))
PY
cat $OUT/sq_summary.txt
