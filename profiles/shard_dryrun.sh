#!/bin/bash
# per-rank step time of the strong-scaling run (C3 split over N GPUs), measured on ONE GPU: shard 0 of N
mkdir -p gpurun_out
for n in 1 2 4 8 1 2 4 8; do
  if [ $n = 1 ]; then extra=""; else extra="--emulate-shard-of $n"; fi
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-psi-check $extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('shard_of=$n genes=%d ms_per_step=%.3f kernel_ms=%.3f frac=%.3f' % (d['config']['genes_per_rank'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['frac']))"
done | tee gpurun_out/shard_dryrun.log
