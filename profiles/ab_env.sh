#!/bin/bash
# bash profiles/ab_env.sh <rounds> "<bench args A>" "<bench args B>" ...  (interleaved; default library)
ROUNDS=$1; shift
for i in $(seq 1 $ROUNDS); do
  for ARGS in "$@"; do
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-psi-check $ARGS 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('[$ARGS]', 'round', $i, 'ms/step %.3f kernel_ms %.3f frac %.4f storage %s' % (d['ms_per_step'], r['avg_kernel_ms'], r['frac'], r.get('count_storage')))"
  done
done
