#!/bin/bash
set -x
timeout 300 python tests/tools/repro_wide_sequence.py 150 40419 1117 > gpurun_out/r4v_repro_seq1117.log 2>&1
cat gpurun_out/r4v_repro_seq1117.log
