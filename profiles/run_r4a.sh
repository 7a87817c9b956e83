#!/bin/bash
# Round 4, GPU call A: (1) quick parity after the prune, (2) placement probe vs step time, (3) Infinity-Cache go/no-go probe
set -x
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "single_steps or noise_stream or psi_after_staged or count_storage_tiers" > $O/r4a_pytest_quick.log 2>&1
tail -3 $O/r4a_pytest_quick.log
for i in 1 2; do
  timeout 600 python profiles/placement_ab.py --config c3 --handles 5 --out $O/r4a_placement_c3.jsonl > $O/r4a_placement_c3_p$i.log 2>&1
  timeout 300 python profiles/placement_ab.py --config c2 --handles 8 --out $O/r4a_placement_c2.jsonl > $O/r4a_placement_c2_p$i.log 2>&1
done
cat $O/r4a_placement_c3.jsonl $O/r4a_placement_c2.jsonl
# Infinity-Cache residency probe: small shards of configs[1], default (non-temporal) build vs BRIE_NT=0
COMMON="--config c2 --steps 300 --warmup 20 --no-f32-leg --no-e2e --no-cpu-baseline --no-psi-check"
for shard in 0 7 20 40; do
  S=""; [ $shard != 0 ] && S="--emulate-shard-of $shard"
  timeout 300 python bench.py $COMMON $S > $O/r4a_ic_c2_of${shard}_nt1.json 2> $O/r4a_ic_c2_of${shard}_nt1.err
  BRIE_AMD_LIB=$PWD/brie_amd/lib/variants/libbrie_amd_nt0_fast1.so timeout 300 python bench.py $COMMON $S > $O/r4a_ic_c2_of${shard}_nt0.json 2> $O/r4a_ic_c2_of${shard}_nt0.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4a_ic_c2_of*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, "ms/step", d["ms_per_step"], "kernel_ms", r.get("avg_kernel_ms"), "elems", d["config"].get("elements_per_step"), "frac", r["frac"], "traffic", r.get("traffic"), "alg", r.get("algorithmic_bytes_per_launch"))
    except Exception as e:
        print(f, "ERR", e)
PY
