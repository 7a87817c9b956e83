"""The product path as it is: a fresh process, one handle of a BASELINE shape, the first brie_step searches by itself
(defaults: four sets in all, interleaved candidates, stop at 6 050 GB/s); report what it probed, what it kept and the
step time on the kept set.   python profiles/placement_auto.py --config c3 >> gpurun_out/r4w_placement_auto_c3.jsonl"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("BRIE_DEVICE_CACHE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--mc", type=int, default=1)
    args = ap.parse_args()
    import torch
    import bench
    from brie_amd import _capi
    cfg = bench.CONFIGS[args.config]
    dev = torch.device("cuda", 0)
    seed = bench.config_seed(args.config)
    Xc, size, layers, eff = bench.make_inputs(torch, dev, cfg, 0, cfg["Ng"], seed)
    sh = _capi.Shard(cfg["Nc"], cfg["Ng"], cfg["Kc"], n_layers=cfg["L"], has_efflen=eff is not None, seed=seed)
    for l in range(cfg["L"]):
        sh.upload(_capi.COUNT1 + l, layers[l])
    if eff is not None:
        sh.upload(_capi.EFFLEN, eff)
    sh.add_pseudo_count(0.01)
    if cfg["Kc"]:
        sh.upload(_capi.XC, Xc)
    del layers
    torch.cuda.empty_cache()
    sh.init_state()
    sh.synchronize()
    t0 = time.time()
    sh.step(3, 0.005, args.mc, trace=False)          # the first step searches
    sh.synchronize()
    first = time.time() - t0
    sh.profile_enable(True)
    sh.step(12, 0.005, args.mc, trace=False)
    ms, n = sh.profile_read()
    info = sh.placement_info()
    alg = sh.step_algorithmic_bytes()
    print(json.dumps({"config": args.config, "pid": os.getpid(), "t": round(time.time(), 3), "placement": info, "kernel_ms": round(ms / n, 4),
                      "frac": round(alg / (ms / n * 1e-3) / 8e12, 4), "first_three_steps_s": round(first, 3)}), flush=True)
    sh.close()


if __name__ == "__main__":
    main()
