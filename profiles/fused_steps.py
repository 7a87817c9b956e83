#!/usr/bin/env python
"""Wall time per Adam step of small inputs: two launches per step against ONE launch per brie_step (the PERSIST variant of the
step kernel, brie_set_step_fusion), same handle, alternating -- and the state digests of both paths.

    python profiles/fused_steps.py > gpurun_out/r6_fused_steps.json
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [("configs[0] 200 x 500", 200, 500, 0, 2, False), ("200 x 500, Kc 1", 200, 500, 1, 2, False),
          ("300 x 2000, effLen, Kc 2", 300, 2000, 2, 3, True), ("500 x 5000, Kc 1", 500, 5000, 1, 2, False),
          ("1000 x 3000, effLen, Kc 1", 1000, 3000, 1, 3, True), ("2000 x 1000, Kc 3", 2000, 1000, 3, 2, False),
          ("100 x 20000, Kc 0", 100, 20000, 0, 2, False)]


def main():
    from brie_amd import _capi
    from tests import util
    steps = int(os.environ.get("STEPS", "500"))
    global SHAPES
    if os.environ.get("SHAPES"):            # "NcxNgxKcxL[e],..." (e = with effective lengths)
        SHAPES = []
        for t in os.environ["SHAPES"].split(","):
            eff = t.endswith("e")
            Nc, Ng, Kc, L = (int(x) for x in t.rstrip("e").split("x"))
            SHAPES.append(("%d x %d, Kc %d, L %d%s" % (Nc, Ng, Kc, L, ", effLen" if eff else ""), Nc, Ng, Kc, L, eff))
    out = {"steps_per_call": steps, "cases": []}
    for name, Nc, Ng, Kc, L, eff in SHAPES:
        P = util.problem(Nc, Ng, Kc, L, seed=77)
        if not eff:
            P["effLen"] = None
        rec = {"shape": name}
        for mc in (1, 3):
            t, dig = {}, {}
            for mode in (0, 1):
                sh = util.device_shard(P, Nc, Ng, Kc, 5)
                sh.set_step_fusion(mode)
                sh.step(20, 0.005, mc, trace=False)
                sh.synchronize()
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    sh.step(steps, 0.005, mc, trace=False)
                    sh.synchronize()
                    best = min(best, (time.perf_counter() - t0) / steps)
                h = hashlib.sha256()
                for w in (_capi.Z_LOC, _capi.Z_STD_LOG, _capi.INTERCEPT, _capi.SIGMA_LOG):
                    h.update(np.ascontiguousarray(sh.read(w)).tobytes())
                t[mode], dig[mode] = best * 1e6, h.hexdigest()[:16]
                info = sh.step_fusion_info()
                sh.close()
            rec["mc%d" % mc] = {"two_launches_us": round(t[0], 2), "one_launch_us": round(t[1], 2), "same_bits": dig[0] == dig[1],
                                "fused_launches": info["launches"]}
        out["cases"].append(rec)
        print(json.dumps(rec), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
