#!/bin/bash
# bash profiles/ab.sh <rounds> "<bench args>" variant1 variant2 ...   (interleaved rounds, one line per run)
# a variant may carry an environment prefix:  "BRIE_LAYOUT=rowmajor:base"
ROUNDS=$1; ARGS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq 1 $ROUNDS); do
  for spec in "$@"; do
    v=${spec##*:}; envs=""
    if [[ "$spec" == *:* ]]; then envs=${spec%:*}; fi
    env $envs BRIE_AMD_LIB=$R/brie_amd/lib/variants/libbrie_amd_$v.so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-psi-check $ARGS 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$spec', 'round', $i, 'ms/step %.3f kernel_ms %.3f frac %.4f' % (d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['frac']))"
  done
done
