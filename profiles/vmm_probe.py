#!/usr/bin/env python
"""Placement without the lottery? (round 6)  The streamed arrays of a configs[2] / configs[1]-sized problem
built from the HIP virtual-memory API with physical chunks created array after array (seq:<MiB>) or ROUND ROBIN over the arrays
(rr:<MiB>), every array one allocation with unmapped spacers of <MiB> between them (spread:<MiB>) or behind one leading spacer
(skip:<MiB>), next to plain hipMalloc, the placement probe timed on each -- one JSON line per fresh process:

    for i in $(seq 1 10); do python profiles/vmm_probe.py --config c3 >> gpurun_out/r6_vmm_c3.jsonl; done
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = {"c3": (50000, 20000, 2), "c2": (10000, 5000, 3), "c5": (100000, 3750, 2)}
MB = 1 << 20


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3", choices=sorted(SHAPES))
    ap.add_argument("--layouts", default="malloc,seq:2,rr:2,rr:32,rr:256,rr:1024,seq:256,malloc")
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    from brie_amd import _capi
    Nc, Ng, L = SHAPES[args.config]
    names = [x for x in args.layouts.split(",") if x]
    chunk, order = [], []
    for n in names:
        kind, _, mb = n.partition(":")
        chunk.append(int(mb or 2) * MB)
        order.append({"seq": 0, "rr": 1, "malloc": 2, "spread": 3, "skip": 4}[kind])
    gbs, secs = _capi.probe_vmm(Nc, Ng, L, chunk, order, iters=args.iters)
    print(json.dumps({"config": args.config, "layouts": names, "GBs": [round(float(g), 1) for g in gbs],
                      "build_seconds": [round(float(s), 3) for s in secs]}), flush=True)


if __name__ == "__main__":
    main()
