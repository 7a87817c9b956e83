#!/bin/bash
# round 2, second GPU call: mixed count tiers, comm, full-schedule parity tests; bench with f32 leg + live PMC;
# 2-rank bench dry run (gloo, both ranks on GPU 0); strict-math A/B; c3_cli PSI evidence; whole GPU suite
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
( time python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mixed or full_default or tiers or staged" -s ) > $O/pytest_new.log 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
BRIE_BENCH_SINGLE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --config c2 --steps 10 --warmup 2 > $O/bench_n2_gloo_c2.json 2> $O/bench_n2.err
BRIE_AMD_LIB=$R/brie_amd/lib/libbrie_amd_hip_strict.so python bench.py --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg > $O/bench_strict.json 2> $O/bench_strict.err
BRIE_AMD_LIB=$R/brie_amd/lib/libbrie_amd_hip_strict.so python bench.py --mc 3 --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg > $O/bench_strict_mc3.json 2>> $O/bench_strict.err
python bench.py --mc 3 --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg > $O/bench_mc3.json 2>> $O/bench.err
BRIE_COUNT_TIERS=uniform python bench.py --no-cpu-baseline --no-psi-check --no-pmc --no-f32-leg > $O/bench_uniform_u16.json 2>> $O/bench.err
( time python profiles/psi_delta.py --cases c3_cli --out $O/psi_c3cli.json ) > $O/psi.log 2>&1
( time python -m pytest tests -x -q -m gpu ) > $O/pytest_all.log 2>&1
tail -4 $O/pytest_new.log $O/pytest_all.log; grep "^c3" $O/psi.log; head -c 1500 $O/bench.json; echo; tail -3 $O/bench_n2.err
