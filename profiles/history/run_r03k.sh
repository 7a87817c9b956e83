cd $GRAFT_REPO_ROOT; O=gpurun_out/r03l; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_distributed.py tests/test_gpu_comm.py -x -q -m gpu -k "wide or coupled or randomised or margin or comm or shard" > $O/pytest_sel.log 2>&1; grep -E "passed|failed|Error" $O/pytest_sel.log | tail -n 5
BRIE_AMD_LIB=$GRAFT_REPO_ROOT/brie_amd/lib/variants/libbrie_amd_prof.so python profiles/tile_phases.py 2>/dev/null | tee $O/tile_phases.log
python profiles/wide_ab.py --rounds 3 2>/dev/null | tee $O/wide_ab.log | tail -n 12
