#!/bin/bash
# round 3, GPU call a: GPU suite with the round's new paths + first measurements
# (fused finalize on configs[0], rows_per_chunk sweep on configs[1], staged ingest A/B, bench lines incl. the
#  self-launched 2-rank dry run on one GPU)
mkdir -p gpurun_out
O=gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r3a_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r3a_pytest.log
tail -5 $O/r3a_pytest.log
python profiles/c1_fused_ab.py --out $O/r3a_c1_fused_ab.json > $O/r3a_c1_fused_ab.log 2>&1
python profiles/rpc_sweep.py --config c2 --out $O/r3a_rpc_c2.json > $O/r3a_rpc_c2.log 2>&1
python profiles/rpc_sweep.py --config c3 --rpc 128,192,196,200,256,384,512 --steps 12 --reps 2 --out $O/r3a_rpc_c3.json > $O/r3a_rpc_c3.log 2>&1
python profiles/ingest_ab.py --threads 2,4,6,8 --out $O/r3a_ingest_ab.json > $O/r3a_ingest_ab.log 2>&1
python bench.py --config c1 --no-pmc --steps 200 --warmup 20 > $O/r3a_bench_c1.json 2> $O/r3a_bench_c1.err
python bench.py --config c2 --no-pmc > $O/r3a_bench_c2.json 2> $O/r3a_bench_c2.err
BRIE_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --config c2 --no-pmc > $O/r3a_bench_c2_n2_gloo_one_gpu.json 2> $O/r3a_bench_c2_n2.err
python bench.py > $O/r3a_bench_c3.json 2> $O/r3a_bench_c3.err
tail -3 $O/r3a_c1_fused_ab.log $O/r3a_ingest_ab.log
