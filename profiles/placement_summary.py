#!/usr/bin/env python
"""profiles/r4_placement_ab.json from the per-handle records of placement_ab.py (calls r4a, r4e, r4f): step time on the
first placement of a handle (what BRIE_PLACEMENT_TRIES=1 runs on) against the kept one of three, per config and call."""
import glob
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ALG = {"c3": 56.0e9, "c2": 3.0e9}            # algorithmic bytes per launch (48 + 4 L per element)
GOOD_MS = {"c3": 8.3, "c2": 3.0e9 / 0.80 / 8e12 * 1e3}     # configs[2]: <= 8.3 ms (frac >= 0.843); configs[1]: frac >= 0.80
out = {"what": __doc__, "good_ms": GOOD_MS, "calls": {}}
for f in sorted(glob.glob(os.path.join(HERE, "r4?_placement*_c?.jsonl"))):
    rows = [json.loads(l) for l in open(f) if l.strip() and "summary" not in l]
    if not rows or "step_ms_before" not in rows[0]:
        continue
    cfg = rows[0]["config"]
    fresh = [r for r in rows if r["it"] == 0]          # first handle of a process: nothing of an earlier handle is reused
    def stats(rs):
        b, a = [r["step_ms_before"] for r in rs], [r["step_ms_after"] for r in rs]
        return {"handles": len(rs), "processes": len({r["pid"] for r in rs}),
                "first_placement_ms": {"min": min(b), "max": max(b), "mean": sum(b) / len(b),
                                       "hits": sum(x <= GOOD_MS[cfg] for x in b)},
                "kept_of_three_ms": {"min": min(a), "max": max(a), "mean": sum(a) / len(a),
                                     "hits": sum(x <= GOOD_MS[cfg] for x in a)},
                "frac_first": [round(ALG[cfg] / (x * 1e-3) / 8e12, 3) for x in b],
                "frac_kept": [round(ALG[cfg] / (x * 1e-3) / 8e12, 3) for x in a],
                "search_seconds": {"min": min(r["tune"]["seconds"] for r in rs), "max": max(r["tune"]["seconds"] for r in rs)},
                "probe_GBs_of_the_sets": [r["tune"]["GBs"] for r in rs]}
    out["calls"][os.path.basename(f)] = {"config": cfg, "all_handles": stats(rows), "first_handle_of_each_process": stats(fresh)}
tot = {}
for name, c in out["calls"].items():
    t = tot.setdefault(c["config"], {"handles": 0, "first_hits": 0, "kept_hits": 0})
    s = c["all_handles"]
    t["handles"] += s["handles"]; t["first_hits"] += s["first_placement_ms"]["hits"]; t["kept_hits"] += s["kept_of_three_ms"]["hits"]
out["totals"] = tot
out["bit_identical"] = "tests/test_gpu_placement.py (forced search over three sets, 4 storage variants; automatic search at 312 MB per step)"
json.dump(out, open(os.path.join(HERE, "r4_placement_ab.json"), "w"), indent=1)
print(json.dumps(out["totals"]))
for name, c in out["calls"].items():
    s = c["all_handles"]
    print(name, c["config"], s["handles"], "first", s["first_placement_ms"], "kept", s["kept_of_three_ms"])
