cd $GRAFT_REPO_ROOT; O=gpurun_out/r02z; mkdir -p $O
python profiles/graph_ab.py > $O/graph_ab.log 2> $O/err.log; cat $O/graph_ab.log; tail -n 3 $O/err.log
( time python -m pytest tests -q -m gpu -x ) > $O/pytest_all.log 2>&1
grep -E "passed|failed" $O/pytest_all.log; grep -E "^FAILED" $O/pytest_all.log | head
