#!/usr/bin/env python
"""Does a VALU instruction overwrite a data register of an INLINE-ASSEMBLY `global_store_dwordx4 ... sc1` within the two
instructions behind it?  gfx950 wants two wait states there (stores of more than 8 bytes); the compiler inserts them for
stores it knows, not behind inline assembly.  Call r8a lost its published rows exactly so (docs/evidence_r6.md section 5).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ibrie_amd/csrc -DBRIE_KC=0 --cuda-device-only -S \
        brie_amd/csrc/brie_inst.hip -o /tmp/inst_kc0.s
    python profiles/asm_store_hazard_scan.py "/tmp/inst_kc*.s"

(An `s_nop` inside the assembly text follows the store on the same logical statement; it counts as wait states.)
"""
import re, sys, glob
def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()
tot = bad = 0
for f in sorted(glob.glob(sys.argv[1])):
    lines = [l.strip() for l in open(f).read().split("\n")]
    kern = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\S+):", l)
        if m: kern = m.group(1)
        if l.startswith("global_store_dwordx4") and "sc1" in l:
            data = regs(l.split(",")[1].strip())
            tot += 1
            k, seen = i + 1, 0
            while seen < 2 and k < len(lines):
                x = lines[k]; k += 1
                if not x or x.startswith(";") or x.startswith(".") or x.endswith(":"): continue
                if x.startswith("s_nop"):
                    seen += 1 + int(x.split()[1])
                    continue
                seen += 1
                if x.startswith("v_") and not x.startswith("v_cmp") and not x.startswith("v_readlane") and not x.startswith("v_readfirstlane"):
                    dst = regs(x.split()[1].rstrip(","))
                    if dst & data:
                        bad += 1
                        print(f.split("/")[-1], kern[:60], "store", l, "| then (%d)" % seen, x)
print("asm stores", tot, "hazards", bad)
