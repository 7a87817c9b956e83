#!/usr/bin/env python
"""Two builds of libbrie_amd.so on the same problems, alternating processes: wall time per Adam step and the sha256 of the
state after the steps (a change that claims to leave every bit alone has to show the same digests).

    python profiles/lib_ab.py --a brie_amd/lib/libbrie_amd_prev.so --b brie_amd/lib/libbrie_amd.so --out gpurun_out/lib_ab.json
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = {
    # name: Nc, Ng, Kc, L, effLen
    "c2_10kx5k_eff3": (10000, 5000, 1, 3, True),
    "3000x5000_eff3_kc2": (3000, 5000, 2, 3, True),
    "20kx4k_eff3_kc0": (20000, 4000, 0, 3, True),
    "10kx5k_eff2_kc3": (10000, 5000, 3, 2, True),
    "10kx5k_lik2_kc3": (10000, 5000, 3, 2, False),
    "50kx2560_lik2_kc3": (50000, 2560, 3, 2, False),          # a 1/8 gene shard of configs[2]
    "10kx5k_eff3_kc8": (10000, 5000, 8, 3, True),
    "10kx5k_lik2_kc0": (10000, 5000, 0, 2, False),
    "10kx5k_eff3_kc4": (10000, 5000, 4, 3, True),
    "10kx5k_eff3_kc6": (10000, 5000, 6, 3, True),
    "10kx5k_eff3_kc7": (10000, 5000, 7, 3, True),
    "10kx5k_eff2_kc5": (10000, 5000, 5, 2, True),
    "10kx5k_lik2_kc8": (10000, 5000, 8, 2, False),
    "10kx5k_lik2_kc5": (10000, 5000, 5, 2, False),
}


def child(args):
    import numpy as np
    import torch
    import bench
    from brie_amd import _capi
    dev = torch.device("cuda", 0)
    out = {}
    for name in args.shapes.split(","):
        Nc, Ng, Kc, L, eff_on = SHAPES[name]
        cfg = dict(Nc=Nc, Ng=Ng, Kc=Kc, L=3 if eff_on else L, theta=1.5, depth=2.0)
        Xc, size, layers, eff = bench.make_inputs(torch, dev, cfg, 0, Ng, 4242)
        layers = layers[:L]
        sh = _capi.Shard(Nc, Ng, Kc, n_layers=L, has_efflen=eff_on, seed=9)
        for l in range(L):
            sh.upload(_capi.COUNT1 + l, layers[l])
        sh.add_pseudo_count(0.01)
        if eff_on:
            sh.upload(_capi.EFFLEN, eff.cpu().numpy())
        if Kc:
            sh.upload(_capi.XC, Xc)
        sh.init_state()
        for mc in (3, 1):
            sh.step(20, 0.005, mc, trace=False)
            sh.synchronize()
            t0 = time.perf_counter()
            losses = sh.step(args.steps, 0.005, mc, trace=True)
            sh.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            h = hashlib.sha256()
            for which in (_capi.Z_LOC, _capi.Z_STD, _capi.WC_LOC, _capi.INTERCEPT, _capi.SIGMA):
                if which == _capi.WC_LOC and not Kc:
                    continue
                h.update(np.ascontiguousarray(sh.read(which)).tobytes())
            h.update(np.asarray(losses, np.float32).tobytes())
            out["%s_mc%d" % (name, mc)] = {"ms_per_step": dt * 1e3, "sha256": h.hexdigest()[:16]}
            if args.keep:       # a slice of the state, for the size of a difference where the digests differ
                np.savez(os.path.join(args.keep, "%s_mc%d_%s.npz" % (name, mc, args.tag)), Z_loc=sh.read(_capi.Z_LOC)[:, :256],
                         sigma=sh.read(_capi.SIGMA), intercept=sh.read(_capi.INTERCEPT), losses=np.asarray(losses))
        sh.close()
        del layers
        torch.cuda.empty_cache()
    print("RESULT " + json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--a", default=os.path.join(ROOT, "brie_amd", "lib", "libbrie_amd_prev.so"))
    ap.add_argument("--b", default=os.path.join(ROOT, "brie_amd", "lib", "libbrie_amd.so"))
    ap.add_argument("--c", default=None, help="a third build (optional)")
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "lib_ab.json"))
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--keep", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--tag", default="", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.child:
        return child(args)
    libs = [("a", args.a), ("b", args.b)] + ([("c", args.c)] if args.c else [])
    runs = {t: [] for t, _ in libs}
    for rnd in range(args.rounds):
        for tag, lib in libs:
            env = dict(os.environ, BRIE_AMD_LIB=os.path.abspath(lib))
            keep = "/tmp/lib_ab_keep"
            os.makedirs(keep, exist_ok=True)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--steps", str(args.steps), "--shapes", args.shapes]
                               + (["--keep", keep, "--tag", tag] if rnd == 0 else []), env=env, capture_output=True, text=True, timeout=1500)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            if p.returncode != 0 or not line:
                raise SystemExit("child failed (%s): %s" % (lib, p.stderr[-3000:]))
            runs[tag].append(json.loads(line[0][7:]))
    rep = {"a": args.a, "b": args.b, "c": args.c, "steps": args.steps, "cases": {}}
    for case in runs["a"][0]:
        ta = min(r[case]["ms_per_step"] for r in runs["a"])
        tb = min(r[case]["ms_per_step"] for r in runs["b"])
        da = {r[case]["sha256"] for r in runs["a"]}
        db = {r[case]["sha256"] for r in runs["b"]}
        rep["cases"][case] = {"a_ms": round(ta, 4), "b_ms": round(tb, 4), "b_over_a": round(tb / ta, 4),
                              "same_bits": da == db and len(da) == 1, "sha256_a": sorted(da), "sha256_b": sorted(db)}
        if da != db:
            import numpy as np
            za, zb = (np.load(os.path.join("/tmp/lib_ab_keep", "%s_%s.npz" % (case, t))) for t in ("a", "b"))
            rep["cases"][case]["difference"] = {k: float(np.abs(za[k].astype(np.float64) - zb[k]).max()) for k in za.files}
        if args.c:
            tc = min(r[case]["ms_per_step"] for r in runs["c"])
            dc = {r[case]["sha256"] for r in runs["c"]}
            rep["cases"][case].update(c_ms=round(tc, 4), c_over_a=round(tc / ta, 4), c_same_bits=da == dc and len(da) == 1)
        print(case, rep["cases"][case], flush=True)
    with open(args.out, "w") as f:
        json.dump(rep, f, indent=1)


if __name__ == "__main__":
    main()
