#!/usr/bin/env python
"""Where does a step of a MID-SIZE input go?  300 two-launch steps of one shape under rocprofv3 --kernel-trace --stats:
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mid -o t -- python3 profiles/midsize_trace.py 2000 1000 3 2
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util          # noqa: E402

Nc, Ng, Kc, L = (int(x) for x in sys.argv[1:5])
P = util.problem(Nc, Ng, Kc, L, seed=77)
if L == 2:
    P["effLen"] = None
sh = util.device_shard(P, Nc, Ng, Kc, 5)
sh.set_step_fusion(0)
sh.step(20, 0.005, 1, trace=False)
sh.synchronize()
t0 = time.perf_counter()
sh.step(300, 0.005, 1, trace=False)
sh.synchronize()
print("wall us per step", (time.perf_counter() - t0) / 300 * 1e6)
