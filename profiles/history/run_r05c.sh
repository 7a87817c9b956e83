cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c; mkdir -p $O
BRIE_AMD_LIB=$GRAFT_REPO_ROOT/brie_amd/lib/variants/libbrie_amd_cf.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "single_steps or staged or ragged or randomised_shapes or shard" > $O/pytest_cf.log 2>&1; grep -E "passed|failed" $O/pytest_cf.log | tail -n 1
X="--no-pmc --no-f32-leg --no-e2e --config c3"
bash profiles/ab.sh 7 "$X" base cf 2>&1 | tee $O/ab_chunk_fast.log
