cd $GRAFT_REPO_ROOT; O=gpurun_out/r02h; mkdir -p $O
python profiles/alloc_cycles.py > $O/cycles.log 2> $O/err.log
python profiles/alloc_cycles.py >> $O/cycles.log 2>> $O/err.log
cat $O/cycles.log; tail -3 $O/err.log
