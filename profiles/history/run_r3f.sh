#!/bin/bash
# round 3, GPU call f: suite after the ragged-quad packing fix, a fresh-seed soak in ASSERT mode (the frozen state bounds
# on 1000 new cases), the LRT end to end at the API defaults with the round's library
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r3f_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r3f_pytest.log
grep -E "passed|failed|^FAILED|Error" $O/r3f_pytest.log | tail -8
python tests/tools/soak_randomised.py 650 350 777777 > $O/r3f_soak_assert.log 2>&1; echo "soak rc=$?" >> $O/r3f_soak_assert.log
tail -6 $O/r3f_soak_assert.log | cut -c1-600
python profiles/e2e_lrt_c3.py > $O/r3f_e2e_lrt_c3_api_defaults.log 2>&1
tail -1 $O/r3f_e2e_lrt_c3_api_defaults.log > $O/r3f_e2e_lrt_c3_api_defaults.json
tail -2 $O/r3f_e2e_lrt_c3_api_defaults.log | cut -c1-800
