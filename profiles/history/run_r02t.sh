cd $GRAFT_REPO_ROOT; O=gpurun_out/r02t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_distributed.py -q -m gpu -k "wide or coupled or randomised or distributed or sharded" -x > $O/pytest_tile.log 2>&1
BRIE_TILE_HALVES=2 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "wide or coupled or randomised" -x > $O/pytest_tile_h2.log 2>&1
tail -n 2 $O/pytest_tile.log; tail -n 2 $O/pytest_tile_h2.log
timeout 1500 python profiles/wide_ab.py --cases 32:0,64:0,3:32,3:64,32:32 > $O/wide_ab.log 2> $O/err.log
tail -n 1 $O/wide_ab.log; tail -n 3 $O/err.log
