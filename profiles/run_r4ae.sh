#!/bin/bash
# Round 4, GPU call AE: call r4ad saw ONE layout in ONE slab read 5.1 TB/s in round 0 and 6.1 in round 1 of the same
# process, and later every set of every process 4.9 - 5.0: a state of the whole GPU that changes with time, next to the
# layout law?  Sample what the GPU reports about itself (clock levels, power, temperatures; read-only) while a few
# layouts are probed round after round, then while the product path runs in fresh processes.
set -x
O=gpurun_out
ls /sys/class/drm/ > $O/r4ae_sysfs.txt 2>&1
ls /sys/class/drm/card*/device/ >> $O/r4ae_sysfs.txt 2>&1
python profiles/gpu_state_sampler.py --seconds 170 --period 0.1 > $O/r4ae_state.jsonl 2> $O/r4ae_state.err &
SAMPLER=$!
( for i in $(seq 1 80); do echo "{\"t\": $(date +%s.%N)}"; timeout 20 rocm-smi --showclocks --showpower --showtemp --showperflevel --json 2>&1; echo; sleep 1.5; done ) > $O/r4ae_rocm_smi.jsonl 2>&1 &
SMI=$!
timeout 30 amd-smi metric -g 0 --json > $O/r4ae_amd_smi_idle.json 2>&1
for i in 1 2 3; do
  timeout 200 python profiles/layout_probe.py --nc 10000 --ng 5000 --small 64 --rounds 25 \
     --only packed_at_0GB,pitch_8GB,two_groups_32GB,two_groups_48GB >> $O/r4ae_series_small.jsonl 2>> $O/r4ae_series_small.err
done
timeout 30 amd-smi metric -g 0 --json > $O/r4ae_amd_smi_after_series.json 2>&1
for i in 1 2 3 4 5 6 7 8; do
  BRIE_PLACEMENT_SLAB_GB=0 timeout 100 python profiles/placement_auto.py --config c2 >> $O/r4ae_auto_c2.jsonl 2>> $O/r4ae_auto_c2.err
done
for i in 1 2 3; do
  BRIE_PLACEMENT_SLAB_GB=0 timeout 100 python profiles/placement_auto.py --config c3 >> $O/r4ae_auto_c3.jsonl 2>> $O/r4ae_auto_c3.err
done
kill $SAMPLER $SMI
wait
cut -c1-260 $O/r4ae_series_small.jsonl | head -80
cat $O/r4ae_auto_c2.jsonl $O/r4ae_auto_c3.jsonl | cut -c1-220
head -c 1500 $O/r4ae_state.jsonl
