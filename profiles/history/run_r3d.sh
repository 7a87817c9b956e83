#!/bin/bash
# round 3, GPU call d: suite with the frozen parity rule + tightened state bounds, the rule evaluated on the 512-gene
# samples (and the brie-quant schedule on configs[1]), the one soak sequence that failed in r3c with its message,
# the bench line of the headline config with its rocprofv3 summary
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r3d_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r3d_pytest.log
grep -E "passed|failed|^FAILED|Error" $O/r3d_pytest.log | tail -15
python - > $O/r3d_soak_seq61.log 2>&1 <<'PY'
import sys, traceback
sys.path.insert(0, ".")
from tests import test_gpu_parity as T
for case in T._op_sequences(350, seed=31338):
    if case[0] == 61:
        print(case)
        try:
            T.test_randomised_operation_sequences(None, *case)
            print("PASSED")
        except Exception:
            traceback.print_exc()
PY
tail -12 $O/r3d_soak_seq61.log
python profiles/psi_delta.py --cases c1_api,c1_kc0_api,c1_cli,c2_api_512,c3_api_512,c2_cli_128 --out $O/r3d_psi_delta.json > $O/r3d_psi_delta.log 2>&1
tail -7 $O/r3d_psi_delta.log
bash profiles/run_profile.sh r3d > $O/r3d_profile.log 2>&1
python bench.py > $O/r3d_bench_c3.json 2> $O/r3d_bench_c3.err
python bench.py --config c2 > $O/r3d_bench_c2.json 2> $O/r3d_bench_c2.err
python bench.py --config c1 --steps 400 --warmup 20 > $O/r3d_bench_c1.json 2> $O/r3d_bench_c1.err
python bench.py --mc 3 --no-pmc --no-e2e --no-cpu-baseline --no-psi-check > $O/r3d_bench_c3_mc3.json 2> $O/r3d_bench_c3_mc3.err
