cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r02y; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "MFMA" | head -30 > $O/mfma_counters.txt
ARGS="--kc 32 --steps 6 --warmup 2 --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg --no-e2e"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/mf1 -o m -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $O/run1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA --output-format csv -d /tmp/mf2 -o m -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $O/run2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mf3 -o m -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $O/run3.log 2>&1
python3 $GRAFT_REPO_ROOT/profiles/summarize.py /tmp/mf1 > $O/summary_busy.txt 2>&1
python3 $GRAFT_REPO_ROOT/profiles/summarize.py /tmp/mf2 > $O/summary_insts.txt 2>&1
python3 $GRAFT_REPO_ROOT/profiles/summarize.py /tmp/mf3 > $O/summary_trace.txt 2>&1
head -n 12 $O/mfma_counters.txt; grep -i tile $O/summary_busy.txt $O/summary_insts.txt; head -n 5 $O/summary_trace.txt; tail -n 3 $O/run1.log
