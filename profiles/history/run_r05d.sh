cd $GRAFT_REPO_ROOT; O=gpurun_out/r05d; mkdir -p $O
X="--no-pmc --no-f32-leg --no-e2e --config c3"
bash profiles/ab.sh 6 "$X" base cp 2>&1 | tee $O/ab_chunk_perm.log
