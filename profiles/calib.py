"""Measured HBM ceilings of this box for the stream mixes that matter (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brie_amd import _capi
for nr, nw in ((1, 1), (8, 6)):
    for lds, label in ((0, "32 waves/CU"), (40 << 10, "16 waves/CU"), (80 << 10, " 8 waves/CU"), (160 << 10, " 4 waves/CU")):
        g = _capi.calibrate_stream(nr, nw, 2048 << 20, iters=5, lds_bytes=lds)
        print("stream_mix %dR/%dW 2 GiB/stream, %s: %8.1f GB/s" % (nr, nw, label, g))
