#!/bin/bash
# round 3, GPU call c: loss_gene_eval A/B (round-2 build vs branch-free + packed forward pass), PSI-delta on the
# 512-gene samples, soak of the random test families recording what the state bounds need, nccl branch of bench.py
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r3c_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r3c_pytest.log
grep -E "passed|failed" $O/r3c_pytest.log | tail -2
python profiles/loss_gene_ab.py --libs default,brie_amd/lib/variants/libbrie_amd_lg_round2.so --out $O/r3c_loss_gene_ab.json > $O/r3c_loss_gene_ab.log 2>&1
python profiles/psi_delta.py --cases c1_api,c1_kc0_api,c1_cli,c2_api_512,c3_api_512 --out $O/r3c_psi_delta_api.json > $O/r3c_psi_delta_api.log 2>&1
BRIE_SOAK_RECORD=$O/r3c_soak_record.json python tests/tools/soak_randomised.py 650 350 31337 > $O/r3c_soak.log 2>&1
BRIE_BENCH_FORCE_GATHER=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --config c2 --no-pmc --no-e2e \
  > $O/r3c_bench_c2_n1_nccl_forced_gather.json 2> $O/r3c_bench_c2_nccl.err
python bench.py --config c2 > $O/r3c_bench_c2.json 2> $O/r3c_bench_c2.err
tail -4 $O/r3c_loss_gene_ab.log $O/r3c_psi_delta_api.log $O/r3c_soak.log
