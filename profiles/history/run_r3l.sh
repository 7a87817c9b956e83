#!/bin/bash
# round 3, GPU call l: the one-generation block cache of brie_destroy (suite, A/B at configs[2] and configs[4] sizes),
# the new 2-rank coupled test
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r3l_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r3l_pytest.log
grep -E "passed|failed|^FAILED|Error" $O/r3l_pytest.log | tail -8
python profiles/alloc_cache_ab.py --config c3 --out $O/r3l_alloc_cache_ab_c3.json > $O/r3l_alloc_cache_ab_c3.log 2>&1
python profiles/alloc_cache_ab.py --config c5 --reps 3 --out $O/r3l_alloc_cache_ab_c5.json > $O/r3l_alloc_cache_ab_c5.log 2>&1
grep -h what $O/r3l_alloc_cache_ab_c3.log $O/r3l_alloc_cache_ab_c5.log | cut -c1-200
python bench.py --config c5 --no-pmc > $O/r3l_bench_c5_whole.json 2> $O/r3l_bench_c5.err
python -c "
import json
d=json.loads(open('gpurun_out/r3l_bench_c5_whole.json').read().strip().splitlines()[-1]); p=d['pcie_inclusive']
print('c5 e2e', round(p['total_s'],3), {k: round(v,3) for k,v in p['breakdown_s'].items() if isinstance(v,float)})"
