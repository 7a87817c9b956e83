#!/bin/bash
# One parametrised GPU call (replaces round 4's sixty run_r4*.sh):  bash profiles/gpu_call.sh <tag> <leg> [<leg> ...]
# Everything is written under gpurun_out/<tag>_*; copy what is cited into profiles/r5/.  Legs:
#   smoke        __graft_entry__.smoke()
#   pytest       python -m pytest tests -m gpu -q           (PYTEST_ARGS narrows it)
#   bench        python bench.py                (configs[2], the driver's command)      bench_c2 / bench_c1 / bench_mc3 likewise
#   auto_c3:N    N fresh processes of profiles/placement_auto.py --config c3 (the product path's placement search)
#   auto_c2:N    the same at configs[1]
#   forced_c3:N  N fresh processes with an unreachable stop rate (all 8 sets or the time limit)
#   panels       profiles/wide_ab.py at Kc = 128, 256, Kg = 128, Kc = 70 (designs beyond 64 features)
#   profile      rocprofv3 --kernel-trace --stats of the bench command + summary (profiles/run_profile.sh)
set -u
TAG=$1; shift
O=gpurun_out
mkdir -p $O
for LEG in "$@"; do
  case $LEG in
    smoke)     timeout 300 python __graft_entry__.py smoke > $O/${TAG}_smoke.log 2>&1; tail -1 $O/${TAG}_smoke.log ;;
    pytest)    timeout 2400 python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} > $O/${TAG}_pytest_gpu.log 2>&1; tail -4 $O/${TAG}_pytest_gpu.log ;;
    bench)     timeout 900 python bench.py > $O/${TAG}_bench_c3_n1.json 2> $O/${TAG}_bench_c3_n1.err; tail -4 $O/${TAG}_bench_c3_n1.err; cat $O/${TAG}_bench_c3_n1.json ;;
    bench_c2)  timeout 600 python bench.py --config c2 > $O/${TAG}_bench_c2_n1.json 2> $O/${TAG}_bench_c2_n1.err; cat $O/${TAG}_bench_c2_n1.json ;;
    bench_c1)  timeout 600 python bench.py --config c1 > $O/${TAG}_bench_c1_n1.json 2> $O/${TAG}_bench_c1_n1.err; cat $O/${TAG}_bench_c1_n1.json ;;
    bench_mc3) timeout 900 python bench.py --mc 3 --no-pmc --no-e2e > $O/${TAG}_bench_c3_mc3.json 2> $O/${TAG}_bench_c3_mc3.err; cat $O/${TAG}_bench_c3_mc3.json ;;
    auto_c3:*|auto_c2:*)
      CFG=${LEG%%:*}; CFG=${CFG#auto_}; N=${LEG##*:}
      for i in $(seq 1 $N); do timeout 300 python profiles/placement_auto.py --config $CFG >> $O/${TAG}_auto_${CFG}.jsonl 2>> $O/${TAG}_auto.err; done
      cat $O/${TAG}_auto_${CFG}.jsonl ;;
    forced_c3:*)                     # the placement search with an unreachable stop rate: what 8 sets / the time limit cost
      N=${LEG##*:}
      for i in $(seq 1 $N); do BRIE_PLACEMENT_GOOD_GBS=99999 timeout 300 python profiles/placement_auto.py --config c3 >> $O/${TAG}_forced_c3.jsonl 2>> $O/${TAG}_auto.err; done
      cat $O/${TAG}_forced_c3.jsonl ;;
    panels)    timeout 900 python profiles/wide_ab.py --rounds 2 --steps 4 --cases 128:0,256:0,3:128,70:0 > $O/${TAG}_panels_at_c3.log 2>&1; tail -1 $O/${TAG}_panels_at_c3.log ;;
    profile)   timeout 900 bash profiles/run_profile.sh $TAG > $O/${TAG}_run_profile.log 2>&1; head -12 $O/prof_${TAG}/summary.txt ;;
    *) echo "unknown leg $LEG" ;;
  esac
done
