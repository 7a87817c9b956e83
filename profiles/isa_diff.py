#!/usr/bin/env python
"""Which kernels of a translation unit changed between two source trees?  Both trees' brie_inst.hip are compiled to gfx950
assembly for one KC (hipcc --cuda-device-only -S) and every kernel's instruction stream is compared, labels normalised,
comments dropped.  Round 6 used it to show that ONLY the MC_size 3 instantiations of elbo_adam_step differ from round 5's.

    git archive c926a58 brie_amd/csrc include | tar -x -C /tmp/r5src
    python profiles/isa_diff.py /tmp/r5src . --kc 0,1,2,3,4,5,6,7,8
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile


def kernels(path):
    out, cur = {}, None
    for ln in open(path).read().split("\n"):
        m = re.match(r"^(_Z\S+):\s", ln)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is not None:
            if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
                cur = None
                continue
            out[cur].append(ln)
    return out


def norm(lines, name):
    res = []
    for l in lines:
        l = re.sub(r"\.LBB\d+_", ".LBB_", l)
        l = re.sub(r";.*$", "", l).rstrip().replace(name, "SELF")
        if l.strip():
            res.append(l)
    return res


def compile_s(root, kc, out):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(root, "include"),
                    "-I" + os.path.join(root, "brie_amd", "csrc"), "-DBRIE_KC=%d" % kc, "--cuda-device-only", "-S",
                    os.path.join(root, "brie_amd", "csrc", "brie_inst.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("old")
    ap.add_argument("new")
    ap.add_argument("--kc", default="1")
    args = ap.parse_args()
    tmp = tempfile.mkdtemp()
    for kc in [int(k) for k in args.kc.split(",")]:
        fa, fb = os.path.join(tmp, "a%d.s" % kc), os.path.join(tmp, "b%d.s" % kc)
        compile_s(os.path.abspath(args.old), kc, fa)
        compile_s(os.path.abspath(args.new), kc, fb)
        a, b = kernels(fa), kernels(fb)
        same, only_new, by = 0, sorted(set(b) - set(a)), collections.Counter()
        for k in a:
            if k not in b:
                by["gone"] += 1
            elif norm(a[k], k) == norm(b[k], k):
                same += 1
            else:
                m = re.search(r"elbo_adam_stepILi\dELi\dELi(\d)", k)
                by["elbo_adam_step MC_size %s" % m.group(1) if m else k[:40]] += 1
        print("KC %d: %d kernels identical, differing: %s, only in the new tree: %d (%s)" % (
            kc, same, dict(by), len(only_new), ", ".join(sorted(set(re.sub(r"I.*", "", re.sub(r"^_ZN4brie\d+", "", n)) for n in only_new)))),
            flush=True)


if __name__ == "__main__":
    main()
