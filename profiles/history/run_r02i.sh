cd $GRAFT_REPO_ROOT; O=gpurun_out/r02k; mkdir -p $O
timeout 1200 python profiles/tile_cycles.py > $O/tile_cycles.log 2> $O/err.log
cat $O/tile_cycles.log; tail -2 $O/err.log
