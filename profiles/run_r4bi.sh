#!/bin/bash
# Round 4, GPU call BI: the product path of the final library in fresh processes (defaults), configs[2] x 12 and configs[1] x 12, another box
O=gpurun_out
for i in $(seq 1 12); do timeout 200 python profiles/placement_auto.py --config c3 >> $O/r4bi_placement_auto_c3.jsonl 2>> $O/r4bi.err; done
for i in $(seq 1 12); do timeout 100 python profiles/placement_auto.py --config c2 >> $O/r4bi_placement_auto_c2.jsonl 2>> $O/r4bi.err; done
cut -c1-220 $O/r4bi_placement_auto_c3.jsonl; cut -c1-220 $O/r4bi_placement_auto_c2.jsonl
