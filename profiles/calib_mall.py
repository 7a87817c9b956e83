"""Does a working set that fits the 256 MiB Infinity Cache stream faster than HBM? (run on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brie_amd import _capi
for nr, nw in ((8, 6), (1, 1)):
    for mb in (1, 2, 4, 8, 12, 16, 24, 32, 64, 256):
        tot = (nr + nw) * mb
        g = max(_capi.calibrate_stream(nr, nw, mb << 20, iters=40, nt=nt) for nt in (False, True))
        print("stream_mix %dR/%dW %4d MiB/stream (working set %5d MiB): %8.1f GB/s" % (nr, nw, mb, tot, g))
