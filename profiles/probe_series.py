"""Is the first probe of a fresh process taken on a GPU that has not left its idle clocks?  One handle of a BASELINE shape, no
search (BRIE_PLACEMENT_TRIES=1), the effect-free probe on the SAME arrays again and again from the first moment on.
    python profiles/probe_series.py --config c2 >> gpurun_out/r4az_probe_series_c2.jsonl"""
import argparse
import json
import os
import sys
import time

os.environ["BRIE_PLACEMENT_TRIES"] = "1"
os.environ.setdefault("BRIE_DEVICE_CACHE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--n", type=int, default=40)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--sleep-ms", type=float, default=0.0, help="host pause between probes (does the device fall back?)")
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
    time.sleep(0.5)                                   # an idle device, as after host-side work
    t0 = time.perf_counter()
    series = []
    for k in range(args.n):
        g = sh.placement_probe(args.iters)
        series.append((round((time.perf_counter() - t0) * 1e3, 2), round(g, 1)))
        if args.sleep_ms:
            time.sleep(args.sleep_ms * 1e-3)
    print(json.dumps({"config": args.config, "pid": os.getpid(), "iters": args.iters, "sleep_ms": args.sleep_ms,
                      "t_ms__GBs": series}), flush=True)
    sh.close()


if __name__ == "__main__":
    main()
