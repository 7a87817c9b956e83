#!/bin/bash
# Round 4, GPU call AP: the one failing sequence of soak r4ao (index 46 of _op_sequences(150, 737374)) step by step
timeout 600 python tests/tools/repro_sequence.py 150 737374 46 > gpurun_out/r4ap_repro_seq46.log 2>&1
tail -30 gpurun_out/r4ap_repro_seq46.log
