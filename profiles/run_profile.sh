#!/bin/bash
# Profiling recipe used for the committed summaries (run on the GPU box through gpurun):
#   bash profiles/run_profile.sh <tag> [bench args...]
# 1) rocprofv3 --kernel-trace --stats of the bench command
# 2) separate --pmc passes for FETCH_SIZE and WRITE_SIZE (HBM traffic), as
#    /opt/skills/guides/MI355X_MICROARCH.md prescribes (they do not fit one pass).
# Raw rocprofv3 output stays in /tmp; only the condensed summary + the stats CSVs are
# copied to gpurun_out/prof_<tag>/ (then committed under profiles/).
set -u
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
RAW=/tmp/brie_prof_$TAG
rm -rf $RAW; mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg --no-e2e $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- python3 $R/bench.py $ARGS > $OUT/bench_trace.log 2>&1
export BRIE_PLACEMENT_TRIES=1     # counter passes: traffic does not depend on the placement; no 26-GB copies under --pmc
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $RAW/pmc_fetch -o fetch -- python3 $R/bench.py $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $RAW/pmc_write -o write -- python3 $R/bench.py $ARGS > $OUT/bench_write.log 2>&1
python3 $R/profiles/summarize.py $RAW > $OUT/summary.txt 2>&1
find $RAW -name "*kernel_stats.csv" -exec cp {} $OUT/ \;
du -sh $RAW $OUT
cat $OUT/summary.txt
