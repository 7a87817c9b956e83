cd $GRAFT_REPO_ROOT; O=gpurun_out/r04u; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_distributed.py tests/test_gpu_comm.py tests/test_gpu_api.py -x -q -m gpu -k "wide or coupled or randomised or margin or comm or shard or Kg or xg" > $O/pytest_sel.log 2>&1; grep -E "passed|failed|Error" $O/pytest_sel.log | tail -n 5
timeout 600 python profiles/soak_randomised.py 300 0 13579 2>&1 | grep -v amdgpu.ids | tail -n 4
python profiles/wide_ab.py --rounds 3 --cases 3:8,3:16,3:32,3:64,1:16,4:32 2>/dev/null | tee $O/wide_ab_regs.log | head -n 6
BRIE_TILE_KC_REGS=0 python profiles/wide_ab.py --rounds 3 --cases 3:8,3:16,3:32,3:64,1:16,4:32 2>/dev/null | tee $O/wide_ab_mfma.log | head -n 6
