"""Build tuning variants of libbrie_amd.so into brie_amd/lib/variants/ (they travel to the GPU box)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brie_amd.build import compile_library, LIB_DIR

VARIANTS = {
    "nt1_fast1": ["BRIE_NT=1", "BRIE_FAST_MATH=1"],
    "nt0_fast1": ["BRIE_NT=0", "BRIE_FAST_MATH=1"],
    "nt1_fast0": ["BRIE_NT=1", "BRIE_FAST_MATH=0"],
    "nt0_fast0": ["BRIE_NT=0", "BRIE_FAST_MATH=0"],
}
names = sys.argv[1:] or sorted(VARIANTS)
for n in names:
    out = os.path.join(LIB_DIR, "variants", "libbrie_amd_%s.so" % n)
    compile_library(out=out, defines=VARIANTS[n], verbose=True)
