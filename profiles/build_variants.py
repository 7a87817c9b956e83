"""Build tuning variants of libbrie_amd.so into brie_amd/lib/variants/ (they travel to the GPU box)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brie_amd.build import compile_library, LIB_DIR

VARIANTS = {
    "base": [],
    "fast": ["BRIE_FAST_MATH=1"],
}
names = sys.argv[1:] or sorted(VARIANTS)
for n in names:
    out = os.path.join(LIB_DIR, "variants", "libbrie_amd_%s.so" % n)
    compile_library(out=out, defines=VARIANTS[n], verbose=True)
