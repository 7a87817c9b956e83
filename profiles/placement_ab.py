"""Does the effect-free placement probe predict the step time of a handle, and what does keeping the best of N
placements buy?  

    python profiles/placement_ab.py --config c3 --handles 6 --out gpurun_out/placement_c3.jsonl

Per handle (fresh allocations: block cache off, automatic tuning off): probe rate, step-kernel ms (HIP events, 12
steps), then brie_placement_tune over `--tries` sets with an unreachable target (every set is tried, the fastest
kept), probe rate and step ms again.  One JSON line per handle; a summary line at the end."""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("BRIE_DEVICE_CACHE", "0")
os.environ.setdefault("BRIE_PLACEMENT_TRIES", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def step_ms(sh, mc, n=12):
    sh.step(2, 0.005, mc, trace=False)
    sh.synchronize()
    sh.profile_enable(True)
    sh.step(n, 0.005, mc, trace=False)
    ms, k = sh.profile_read()
    sh.profile_enable(False)
    return ms / k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--handles", type=int, default=6)
    ap.add_argument("--tries", type=int, default=3)
    ap.add_argument("--mc", type=int, default=1)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    import bench
    from brie_amd import _capi
    cfg = bench.CONFIGS[args.config]
    dev = torch.device("cuda", 0)
    seed = bench.config_seed(args.config)
    Xc, size, layers, eff = bench.make_inputs(torch, dev, cfg, 0, cfg["Ng"], seed)
    rows = []
    out = open(args.out, "a") if args.out else None
    for it in range(args.handles):
        t0 = time.time()
        sh = _capi.Shard(cfg["Nc"], cfg["Ng"], cfg["Kc"], n_layers=cfg["L"], has_efflen=eff is not None, seed=seed)
        for l in range(cfg["L"]):
            sh.upload(_capi.COUNT1 + l, layers[l])
        if eff is not None:
            sh.upload(_capi.EFFLEN, eff)
        sh.add_pseudo_count(0.01)
        if cfg["Kc"]:
            sh.upload(_capi.XC, Xc)
        sh.init_state()
        sh.synchronize()
        create_s = time.time() - t0
        rec = {"config": args.config, "pid": os.getpid(), "it": it, "create_s": round(create_s, 3),
               "storage_GB": round(sh.step_storage_bytes() / 1e9, 3)}
        rec["probe_before_GBs"] = round(sh.placement_probe(3), 1)
        rec["step_ms_before"] = round(step_ms(sh, args.mc), 4)
        rec["probe_before2_GBs"] = round(sh.placement_probe(3), 1)
        info = sh.placement_tune(args.tries, 1e30)
        rec["tune"] = info
        rec["probe_after_GBs"] = round(sh.placement_probe(3), 1)
        rec["step_ms_after"] = round(step_ms(sh, args.mc), 4)
        rec["step_GBs_before"] = round(sh.step_storage_bytes() / rec["step_ms_before"] / 1e6, 1)
        rec["step_GBs_after"] = round(sh.step_storage_bytes() / rec["step_ms_after"] / 1e6, 1)
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + "\n")
            out.flush()
        rows.append(rec)
        sh.close()
    import numpy as np
    pb = np.array([r["probe_before_GBs"] for r in rows] + [r["probe_after_GBs"] for r in rows])
    sb = np.array([r["step_GBs_before"] for r in rows] + [r["step_GBs_after"] for r in rows])
    summ = {"summary": args.config, "n": len(rows),
            "corr_probe_vs_step_rate": round(float(np.corrcoef(pb, sb)[0, 1]), 3) if len(rows) > 1 else None,
            "step_ms_before": [r["step_ms_before"] for r in rows], "step_ms_after": [r["step_ms_after"] for r in rows],
            "mean_before": round(float(np.mean([r["step_ms_before"] for r in rows])), 4),
            "mean_after": round(float(np.mean([r["step_ms_after"] for r in rows])), 4),
            "tune_seconds": [r["tune"]["seconds"] for r in rows]}
    print(json.dumps(summ), flush=True)
    if out:
        out.write(json.dumps(summ) + "\n")


if __name__ == "__main__":
    main()
