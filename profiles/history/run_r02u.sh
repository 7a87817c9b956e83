cd $GRAFT_REPO_ROOT; O=gpurun_out/r02u; mkdir -p $O
for rep in 1 2 3; do for cfg in "3 32" "3 64" "3 16"; do
  python profiles/tile_c2_ab.py $cfg >> $O/ab.log 2>> $O/err.log
  BRIE_AMD_LIB=$GRAFT_REPO_ROOT/brie_amd/lib/libbrie_amd_c2fold.so python profiles/tile_c2_ab.py $cfg >> $O/ab.log 2>> $O/err.log
done; done
cat $O/ab.log; tail -n 2 $O/err.log
