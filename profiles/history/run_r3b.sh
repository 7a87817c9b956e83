#!/bin/bash
# round 3, GPU call b: counter evidence for the non-headline kernels, per-rank dry runs of the
# strong split with the final library (item 8), occupancy A/B on configs[1]
O=gpurun_out
mkdir -p $O
for spec in "loss_gene:loss_gene_eval:--what loss_gene --launches 2" "c2_step:elbo_adam_step:--what c2_step --launches 8" \
            "tile_kc48:elbo_adam_step_tile:--what tile_kc48 --launches 3" "tile_kg32:elbo_adam_step_tile:--what tile_kg32 --launches 3" \
            "c3_step:elbo_adam_step:--what c3_step --launches 4"; do
  tag=${spec%%:*}; rest=${spec#*:}; kern=${rest%%:*}; wargs=${rest#*:}
  bash profiles/kernel_counters.sh r3b_$tag $kern $wargs > $O/r3b_counters_$tag.txt 2>&1
done
for cap in 1 2 0 1 2 0; do
  BRIE_STEP_OCCUPANCY_CAP=$cap python bench.py --config c2 --no-pmc --no-e2e --no-cpu-baseline --no-psi-check --no-f32-leg --steps 40 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('c2 occupancy cap $cap: ms/step %.4f kernel_ms %.4f frac %.4f' % (d['ms_per_step'], r['avg_kernel_ms'], r['frac']))" >> $O/r3b_c2_occupancy.log
done
for n in 2 4 8; do
  python bench.py --config c3 --emulate-shard-of $n --no-pmc --no-cpu-baseline --no-psi-check > $O/r3b_shard_dryrun_c3_of$n.json 2> $O/r3b_shard_dryrun_c3_of$n.err
  python bench.py --config c5 --emulate-shard-of $n --no-pmc --no-cpu-baseline --no-psi-check > $O/r3b_shard_dryrun_c5_of$n.json 2> $O/r3b_shard_dryrun_c5_of$n.err
done
cat $O/r3b_c2_occupancy.log
for t in loss_gene c2_step tile_kc48 tile_kg32 c3_step; do tail -25 $O/counters_r3b_$t/summary.txt; done
