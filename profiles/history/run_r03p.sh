cd $GRAFT_REPO_ROOT; O=gpurun_out/r03p; mkdir -p $O
BRIE_AMD_LIB=$GRAFT_REPO_ROOT/brie_amd/lib/variants/libbrie_amd_sync8.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/pytest_sync8.log 2>&1; grep -E "passed|failed" $O/pytest_sync8.log | tail -n 2
X="--no-pmc --no-f32-leg --no-e2e"
bash profiles/ab.sh 4 "$X --config c3" base sync4 sync8 sync16 2>&1 | tee $O/ab_sync.log
bash profiles/ab.sh 2 "$X --config c2" base sync4 sync8 sync16 2>&1 | tee -a $O/ab_sync.log
