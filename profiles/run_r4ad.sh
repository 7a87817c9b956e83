#!/bin/bash
# Round 4, GPU call AD: why did the product's slab candidate read 5.9 TB/s (call r4ac) where the slab experiment read
# 6.05 - 6.17 (call r4y)?  (1) the experiment tool with product-like layouts; (2) the product with aligned pieces at
# exactly GAP GiB (mode 1) and with the count layers in the slab as well (mode 3), gaps 32 / 40 / 48
set -x
O=gpurun_out
for i in 1 2 3; do
  timeout 200 python profiles/layout_probe.py --nc 10000 --ng 5000 --small 64 --rounds 2 >> $O/r4ad_layout_small.jsonl 2>> $O/r4ad_layout_small.err
done
cat $O/r4ad_layout_small.jsonl
export BRIE_PLACEMENT_LOG=1
for mode in 0 1 3; do for gap in 32 40 48; do
  for i in 1 2; do
    BRIE_PLACEMENT_SLAB_MODE=$mode BRIE_PLACEMENT_SLAB_GB=$gap timeout 100 python profiles/placement_auto.py --config c2 \
      >> $O/r4ad_auto_c2_mode${mode}_gap${gap}.jsonl 2>> $O/r4ad_auto_c2_mode${mode}_gap${gap}.err
  done
  echo "mode $mode gap $gap"; cat $O/r4ad_auto_c2_mode${mode}_gap${gap}.jsonl | cut -c1-200
done; done
