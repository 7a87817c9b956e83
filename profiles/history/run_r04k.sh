cd $GRAFT_REPO_ROOT; O=gpurun_out/r04k; mkdir -p $O
for i in 1 2; do
echo "== base"; python profiles/occ_pad_sweep.py 2>/dev/null | grep quad
echo "== prefetch 2 rows"; BRIE_AMD_LIB=$GRAFT_REPO_ROOT/brie_amd/lib/variants/libbrie_amd_pf2.so python profiles/occ_pad_sweep.py 2>/dev/null | grep quad
done | tee $O/pf2_sweep.log
