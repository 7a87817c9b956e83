cd $GRAFT_REPO_ROOT; O=gpurun_out/r02v; mkdir -p $O
timeout 1500 python profiles/wide_ab.py --cases 3:5,3:6,3:8,3:12 > $O/wide_ab_small_kg.log 2> $O/err.log
tail -n 1 $O/wide_ab_small_kg.log
( time python -m pytest tests -q -m gpu ) > $O/pytest_all.log 2>&1
grep -E "passed|failed" $O/pytest_all.log; grep -E "^FAILED" $O/pytest_all.log | head
