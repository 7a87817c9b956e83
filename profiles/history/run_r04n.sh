cd $GRAFT_REPO_ROOT; O=gpurun_out/r04n; mkdir -p $O
BRIE_AMD_LIB=$GRAFT_REPO_ROOT/brie_amd/lib/variants/libbrie_amd_rb.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "single_steps or staged or ragged or randomised_shapes" > $O/pytest_rb.log 2>&1; grep -E "passed|failed" $O/pytest_rb.log | tail -n 1
X="--no-pmc --no-f32-leg --no-e2e --config c3"
bash profiles/ab.sh 5 "$X" base rb 2>&1 | tee $O/ab_row_blocks.log
bash profiles/ab.sh 3 "--no-pmc --no-f32-leg --no-e2e --config c2" base rb 2>&1 | tee -a $O/ab_row_blocks.log
