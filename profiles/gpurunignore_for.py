#!/usr/bin/env python
"""Write .gpurunignore so that ONLY the fp32 oracle caches of the given cases travel to the GPU box (an evidence call
of profiles/psi_null.py --hip), or none at all (no argument: the default the tests rely on).

    python profiles/gpurunignore_for.py c3_api_512,c2_api_512      # before the call
    python profiles/gpurunignore_for.py                            # afterwards
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEAD = """brie_amd/build/
brie_amd/build/*
brie_amd/lib/variants/
brie_amd/lib/variants/*
# Oracle caches of profiles/psi_delta.py / psi_null.py (git-ignored, 2 GB, reproducible: tests/golden/psi_null_caches.json
# has the sha256 of the ones the tests can use and the commands).  The GPU TESTS read the committed fixtures under
# tests/golden/ instead, so nothing of the cache travels by default; an evidence call of `psi_null.py --hip` un-ignores
# the *_float32.npz files of the cases it evaluates for that call (profiles/gpurunignore_for.py).
"""
cases = [c for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else []) if c]
lines = [HEAD]
if not cases:
    lines.append("profiles/_psi_cache/\nprofiles/_psi_cache/*\n")
else:
    keep = {"%s_float32.npz" % c for c in cases}
    total = 0
    for f in sorted(os.listdir(os.path.join(ROOT, "profiles", "_psi_cache"))):
        if f in keep:
            total += os.path.getsize(os.path.join(ROOT, "profiles", "_psi_cache", f))
        else:
            lines.append("profiles/_psi_cache/%s\n" % f)
    sys.stderr.write("caches that travel: %.0f MiB\n" % (total / 2.0 ** 20))
with open(os.path.join(ROOT, ".gpurunignore"), "w") as fh:
    fh.write("".join(lines))
