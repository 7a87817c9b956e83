"""Streaming ceilings per calibration variant (run on the GPU box): 1 / 2 vectors in flight per stream, grid-strided
slabs; variant 4 = a workgroup owns contiguous 256-KiB pieces of every stream (the fused kernel's access order)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brie_amd import _capi
for rep in range(2):
    for var in ("1", "2", "4"):
        os.environ["BRIE_CALIB_VARIANT"] = var
        for nr, nw in ((1, 1), (8, 6)):
            for nt in (False, True):
                g = _capi.calibrate_stream(nr, nw, 1 << 30, iters=5, nt=nt)
                print("variant %s  stream_mix %dR/%dW 1 GiB/stream nt=%d: %8.1f GB/s" % (var, nr, nw, nt, g), flush=True)
