#!/bin/bash
# bash profiles/sweep.sh "<bench args>" v1 v2 ...   -> one line per --rows-per-chunk value
ARGS=$1; shift
for r in "$@"; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-psi-check $ARGS --rows-per-chunk $r 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('rpc', $r, 'ms/step %.3f kernel_ms %.3f frac %.4f' % (d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['frac']))"
done
