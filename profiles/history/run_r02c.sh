#!/bin/bash
# round 2, third GPU call: whole GPU suite (no -x), end-to-end fit with / without the streamed results, LRT e2e
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02c
mkdir -p $O
cd $R
( time python -m pytest tests -q -m gpu ) > $O/pytest_all.log 2>&1
python profiles/e2e_fit_api.py > $O/e2e_fit_prefetch.json 2> $O/e2e.err
python profiles/e2e_fit_api.py --no-prefetch > $O/e2e_fit_noprefetch.json 2>> $O/e2e.err
python profiles/e2e_fit_api.py > $O/e2e_fit_prefetch2.json 2>> $O/e2e.err
python profiles/e2e_lrt_c3.py > $O/e2e_lrt_c3_api_defaults.json 2>> $O/e2e.err
python bench.py --no-cpu-baseline --no-psi-check --no-pmc > $O/bench_quick.json 2>> $O/e2e.err
grep -E "passed|failed" $O/pytest_all.log; grep -E "^FAILED|^ERROR" $O/pytest_all.log | head -20; cat $O/e2e_fit_*.json; tail -5 $O/e2e.err
