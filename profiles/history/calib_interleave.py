"""14 separate arrays against one array of interleaved 1-KiB pieces: the no-arithmetic stream_mix kernel (8 reads, 6 writes,
non-temporal, the step kernel's mix), alternating, several rounds.  BRIE_CALIB_INTERLEAVE=1 switches the layout."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brie_amd import _capi

for rnd in range(5):
    row = {}
    for il in ("0", "1"):
        os.environ["BRIE_CALIB_INTERLEAVE"] = il
        for var in ("1", "2", "4"):
            os.environ["BRIE_CALIB_VARIANT"] = var
            row["%s_U%s" % ("interleaved" if il == "1" else "separate", var)] = round(_capi.calibrate_stream(8, 6, 1 << 30, 5, nt=True))
    print(json.dumps(row), flush=True)
