"""Is the two-speed behaviour of fresh allocations (5.25 / 6.15 TB/s for the same kernel) a matter of physical contiguity?
stream_mix 8R/6W non-temporal, U = 4 (the step kernel's walk), fresh buffers per call, alternating hipMalloc and
hipExtMallocWithFlags(hipDeviceMallocContiguous)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brie_amd import _capi

os.environ["BRIE_CALIB_VARIANT"] = "4"
for rnd in range(12):
    row = {}
    for ct in ("0", "1"):
        os.environ["BRIE_CALIB_CONTIGUOUS"] = ct
        try:
            row["contiguous" if ct == "1" else "default"] = round(_capi.calibrate_stream(8, 6, 1 << 30, 5, nt=True))
        except Exception as e:      # noqa
            row["contiguous_error"] = str(e)[:80]
    print(json.dumps(row), flush=True)
