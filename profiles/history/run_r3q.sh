#!/bin/bash
# r3q: what the full-size short-horizon parity cases NEED (printed by tests/test_gpu_fullsize.py::_quad_states_close),
# before their bounds are tightened to the rule of assert_states_close
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "config2 or config3 or config5" > gpurun_out/r3q_fullsize_needs.log 2>&1
echo "rc $?" >> gpurun_out/r3q_fullsize_needs.log
grep -E "^C[235] |passed|failed|rc " gpurun_out/r3q_fullsize_needs.log | tail -80
