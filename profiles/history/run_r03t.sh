cd $GRAFT_REPO_ROOT; O=gpurun_out/r03t; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -n 2
python profiles/wide_ab.py --rounds 4 --cases 16:0,32:0,64:0,3:8,3:16,3:32,3:64,32:32,64:32 2>/dev/null | tee $O/wide_ab.log | head -n 9
