"""Is it the SPACING of the streamed arrays or the physical memory they sit in that decides the streaming rate?
One slab per process; the eight arrays of the headline shape (six state arrays of 4.04 GB, two u8 count layers of 1.01 GB)
are carved out of it with different spacings and orders, the placement probe is timed for each layout on that same memory.

    python profiles/layout_probe.py [--rounds 2] >> gpurun_out/r4k_layout_probe.jsonl"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--nc", type=int, default=50000)
    ap.add_argument("--ng", type=int, default=20000)
    ap.add_argument("--small", type=int, default=0, help="GB: the fourth experiment, a small set inside a slab of this size")
    ap.add_argument("--scan", type=int, default=0, help="GB: the third experiment, one packed set moved through a slab of this size")
    ap.add_argument("--scan-step", type=int, default=8)
    ap.add_argument("--spread", type=int, default=0, help="GB: the second experiment, arrays spread over a slab of this size")
    ap.add_argument("--only", default="", help="comma-separated layout names: time series of a few layouts over many rounds")
    args = ap.parse_args()
    from brie_amd import _capi
    Nc, Ng = args.nc, args.ng
    ld = -(-Ng // 256) * 256
    mat, cnt = Nc * ld * 4, Nc * ld
    MB = 1 << 20
    up = lambda x, a: -(-x // a) * a
    S = up(mat, 2 * MB)                       # what hipMalloc's 2-MiB granularity gives back-to-back allocations
    layouts = {}

    def pack(spacing_state, spacing_cnt=None, start=0, order=None):
        o, pos = [], start
        for i in range(6):
            o.append(pos)
            pos += spacing_state
        sc = spacing_cnt or up(cnt, 2 * MB)
        for i in range(2):
            o.append(pos)
            pos += sc
        return o

    layouts["back_to_back_2MiB"] = pack(S)
    for name, d in (("plus_4KiB", 4096), ("plus_68KiB", 69632), ("plus_1MiB_4KiB", MB + 4096), ("plus_2MiB", 2 * MB),
                    ("plus_6MiB", 6 * MB), ("plus_34MiB", 34 * MB), ("plus_257MiB", 257 * MB), ("plus_1GiB_2MiB", (1 << 30) + 2 * MB)):
        layouts["spacing_" + name] = pack(S + d)
    # uneven: every array its own extra (multiples of 2 MiB times small primes)
    o, pos = [], 0
    for i, p in enumerate((0, 3, 7, 13, 29, 53, 101, 211)):
        o.append(pos)
        pos += (S if i < 6 else up(cnt, 2 * MB)) + p * 2 * MB
    layouts["uneven_primes_x_2MiB"] = o
    o, pos = [], 0
    for i, p in enumerate((0, 3, 7, 13, 29, 53, 101, 211)):
        o.append(pos)
        pos += (S if i < 6 else up(cnt, 2 * MB)) + p * 4352
    layouts["uneven_primes_x_4352B"] = o
    # two groups far apart (what the fast sets of call r4g looked like: three arrays elsewhere)
    far = 8 * S + (1 << 30)
    o = pack(S)
    o[0] += far; o[6] += far; o[7] += far
    layouts["mu_and_counts_far_away"] = o
    if args.spread:
        # second experiment: a slab of most of the HBM, the arrays spread over it in groups
        GB = 1 << 30
        c2 = up(cnt, 2 * MB)
        layouts = {}
        layouts["contiguous_at_0"] = pack(S)
        layouts["contiguous_at_100GB"] = pack(S, start=100 * GB)
        layouts["contiguous_at_190GB"] = pack(S, start=190 * GB)
        for name, step in (("spread_18GB", 18), ("spread_27GB", 27), ("spread_36GB", 36)):
            o = [i * step * GB for i in range(6)]
            o += [6 * step * GB if 6 * step * GB + 2 * c2 + mat < args.spread * GB else 5 * step * GB + S,
                  (6 * step * GB if 6 * step * GB + 2 * c2 + mat < args.spread * GB else 5 * step * GB + S) + c2]
            layouts[name] = o
        layouts["two_groups_0_110GB"] = [0, S, 2 * S, 110 * GB, 110 * GB + S, 110 * GB + 2 * S, 3 * S, 110 * GB + 3 * S]
        layouts["three_groups_0_70_140GB"] = [0, S, 70 * GB, 70 * GB + S, 140 * GB, 140 * GB + S, 2 * S, 70 * GB + 2 * S]
        layouts["pairs_mu_m_v_together"] = [0, 60 * GB, S, 2 * S, 60 * GB + S, 60 * GB + 2 * S, 3 * S, 60 * GB + 3 * S]
        layouts["mu_and_counts_33GB_away"] = pack(S)
        layouts["mu_and_counts_33GB_away"][0] += 8 * S + GB
        layouts["mu_and_counts_33GB_away"][6] += 8 * S + GB
        layouts["mu_and_counts_33GB_away"][7] += 8 * S + GB
        layouts["only_counts_far"] = pack(S)
        layouts["only_counts_far"][6] += 100 * GB
        layouts["only_counts_far"][7] += 100 * GB
        layouts["only_mu_far"] = pack(S)
        layouts["only_mu_far"][0] += 100 * GB
    if args.scan:
        # third experiment: ONE packed set moved through the slab -- which regions are slow?
        GB = 1 << 30
        layouts = {}
        span = 6 * S + 2 * up(cnt, 2 * MB)
        x = 0
        while x * GB + span + 64 * MB < args.scan * GB:
            layouts["packed_at_%03dGB" % x] = pack(S, start=x * GB)
            x += args.scan_step
    if args.small:
        # fourth experiment: a SMALL set (configs[1]: 0.2-GB arrays) inside a slab of args.small GB: how far apart must its arrays lie?
        GB = 1 << 30
        c2 = up(cnt, 2 * MB)
        layouts = {}
        for start in (0, 20, 40):
            layouts["packed_at_%dGB" % start] = pack(S, start=start * GB)
        for pitch_gb in (0.5, 1, 2, 3, 4, 5, 6, 7, 8):
            pitch = int(pitch_gb * GB)
            layouts["pitch_%sGB" % pitch_gb] = [i * pitch for i in range(8)]
        layouts["two_groups_16GB"] = [0, S, 2 * S, 16 * GB, 16 * GB + S, 16 * GB + 2 * S, 3 * S, 16 * GB + 3 * S]
        layouts["two_groups_32GB"] = [0, S, 2 * S, 32 * GB, 32 * GB + S, 32 * GB + 2 * S, 3 * S, 32 * GB + 3 * S]
        layouts["two_groups_48GB"] = [0, S, 2 * S, 48 * GB, 48 * GB + S, 48 * GB + 2 * S, 3 * S, 48 * GB + 3 * S]
        layouts["three_groups_20GB"] = [0, S, 20 * GB, 20 * GB + S, 40 * GB, 40 * GB + S, 2 * S, 20 * GB + 2 * S]
        # what a slab candidate of the product looked like in call r4ac (pieces not 2-MiB aligned, the second group 40 GiB
        # after the end of the first, both count layers next to the second group), and its variants
        B = 3 * mat + 40 * GB
        layouts["product_like_unaligned_40GB"] = [0, mat, 2 * mat, B, B + mat, B + 2 * mat, B + 3 * mat + 2 * MB, B + 3 * mat + 2 * MB + c2]
        layouts["aligned_40GB_counts_at_B"] = [0, S, 2 * S, 40 * GB, 40 * GB + S, 40 * GB + 2 * S, 40 * GB + 3 * S, 40 * GB + 3 * S + c2]
        layouts["aligned_32GB_counts_at_B"] = [0, S, 2 * S, 32 * GB, 32 * GB + S, 32 * GB + 2 * S, 32 * GB + 3 * S, 32 * GB + 3 * S + c2]
        layouts["unaligned_32GB_counts_split"] = [0, mat, 2 * mat, 32 * GB, 32 * GB + mat, 32 * GB + 2 * mat, 3 * mat, 32 * GB + 3 * mat]
    if args.only:
        layouts = {n: layouts[n] for n in args.only.split(",")}
    names = list(layouts)
    off = np.array([layouts[n] for n in names], np.int64)
    slab = int(off.max() + mat + 64 * MB)
    if args.spread:
        slab = max(slab, args.spread << 30)
    if args.scan:
        slab = max(slab, args.scan << 30)
    if args.small:
        slab = max(slab, args.small << 30)
    for r in range(args.rounds):
        g = _capi.probe_layouts(Nc, Ng, slab, off, iters=3)
        print(json.dumps({"pid": os.getpid(), "t": round(time.time(), 3), "round": r, "slab_GB": round(slab / 1e9, 1),
                          "GBs": {n: round(float(x), 1) for n, x in zip(names, g)}}), flush=True)


if __name__ == "__main__":
    main()
