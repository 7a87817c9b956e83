#!/bin/bash
# Round 4, GPU call AZ: the probe on the SAME arrays from the first moment of an idle device on (no search): does the
# reading climb?  configs[1] and configs[2], back to back and with host pauses between the probes
O=gpurun_out
for i in 1 2 3 4; do timeout 100 python profiles/probe_series.py --config c2 --n 40 >> $O/r4az_probe_series_c2.jsonl 2>> $O/r4az.err; done
for i in 1 2; do timeout 100 python profiles/probe_series.py --config c2 --n 30 --sleep-ms 20 >> $O/r4az_probe_series_c2.jsonl 2>> $O/r4az.err; done
for i in 1 2 3; do timeout 200 python profiles/probe_series.py --config c3 --n 12 >> $O/r4az_probe_series_c3.jsonl 2>> $O/r4az.err; done
cut -c1-700 $O/r4az_probe_series_c2.jsonl; cut -c1-500 $O/r4az_probe_series_c3.jsonl
