cd $GRAFT_REPO_ROOT; O=gpurun_out/r04m; mkdir -p $O
X="--no-pmc --no-f32-leg --no-e2e --config c3"
bash profiles/ab.sh 4 "$X" base rw3 rw2 rw2pf2 2>&1 | tee $O/ab_row_waves.log
