cd $GRAFT_REPO_ROOT; O=gpurun_out/r02l; mkdir -p $O
( time python -m pytest tests -q -m gpu ) > $O/pytest_all.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
bash profiles/run_profile.sh r02l > $O/prof.log 2>&1
grep -E "passed|failed" $O/pytest_all.log; grep -E "^FAILED" $O/pytest_all.log | head; tail -1 $O/smoke.log; head -c 400 $O/bench.json; echo; head -5 gpurun_out/prof_r02l/summary.txt
