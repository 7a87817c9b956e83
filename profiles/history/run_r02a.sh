#!/bin/bash
# round 2, first GPU call: new tests, PSI-delta evidence (cached-oracle cases), bench, rocprofv3 profile
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r02a
cd $R
( time python -m pytest tests/test_gpu_comm.py tests/test_gpu_fullsize.py -x -q -m gpu -s ) > gpurun_out/r02a/pytest_new.log 2>&1
( time python profiles/psi_delta.py --cases c1_api,c1_kc0_api,c1_cli,c2_api,c2_cli,c3_api --out gpurun_out/r02a/psi_delta_partial.json ) > gpurun_out/r02a/psi.log 2>&1
python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
bash profiles/run_profile.sh r02a > gpurun_out/r02a/prof.log 2>&1
( time python -m pytest tests -x -q -m gpu ) > gpurun_out/r02a/pytest_all.log 2>&1
tail -5 gpurun_out/r02a/pytest_new.log gpurun_out/r02a/pytest_all.log; tail -12 gpurun_out/r02a/psi.log
