"""Workgroups per CU of the narrow step kernel on ONE handle of a BASELINE config, the cap toggled every `--steps` steps
(BRIE_STEP_OCCUPANCY_CAP_DYNAMIC): one / two per CU against the hardware's occupancy.

    python profiles/occ_ab2.py --config c2 [--mc 1] [--shard-of 8] [--rounds 8]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BRIE_STEP_OCCUPANCY_CAP_DYNAMIC"] = "1"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--mc", type=int, default=1)
    ap.add_argument("--shard-of", type=int, default=0)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--rounds", type=int, default=8)
    args = ap.parse_args()
    import torch
    import bench
    from brie_amd import _capi
    from brie_amd.sharding import gene_shard
    cfg = dict(bench.CONFIGS[args.config])
    Nc, Ng, Kc, L = cfg["Nc"], cfg["Ng"], cfg["Kc"], cfg["L"]
    g0, g1 = (0, Ng) if not args.shard_of else gene_shard(Ng, 0, args.shard_of)
    dev = torch.device("cuda", 0)
    seed = bench.config_seed(args.config)
    Xc, size, layers, eff = bench.make_inputs(torch, dev, cfg, g0, g1, seed)
    sh = _capi.Shard(Nc, g1 - g0, Kc, n_layers=L, has_efflen=L == 3, seed=seed, gene_offset=g0)
    for l in range(L):
        sh.upload(_capi.COUNT1 + l, layers[l])
    sh.add_pseudo_count(0.01)
    if L == 3:
        sh.upload(_capi.EFFLEN, eff.cpu().numpy())
    if Kc:
        sh.upload(_capi.XC, Xc)
    del layers
    sh.init_state(); sh.step(5, 0.005, args.mc, trace=False); sh.synchronize()
    tot = {"1": 0.0, "2": 0.0, "0": 0.0}
    for rnd in range(args.rounds):
        row = {"config": args.config, "genes": g1 - g0, "mc": args.mc, "storage": sh.count_storage}
        for cap, name in (("1", "one_per_CU"), ("2", "two_per_CU"), ("0", "hardware_occupancy")):
            os.environ["BRIE_STEP_OCCUPANCY_CAP"] = cap
            sh.step(3, 0.005, args.mc, trace=False); sh.synchronize()
            t0 = time.perf_counter(); sh.step(args.steps, 0.005, args.mc, trace=False); sh.synchronize()
            ms = (time.perf_counter() - t0) / args.steps * 1e3
            row[name] = round(ms, 4); tot[cap] += ms
        print(json.dumps(row), flush=True)
    print(json.dumps({"config": args.config, "genes": g1 - g0, "mc": args.mc,
                      "mean_ms": {"one_per_CU": tot["1"] / args.rounds, "two_per_CU": tot["2"] / args.rounds, "hardware": tot["0"] / args.rounds}}))
    sh.close()


if __name__ == "__main__":
    main()
