"""Build the HIP shared library in-tree (gfx950 only).

    python -m brie_amd.build

Translation units: brie_capi.hip (C ABI + small kernels), brie_comm.hip (RCCL entry points) and brie_inst.hip once per cell-feature
count KC = 0..8 (the template instantiations of the two streaming kernels), compiled in parallel
and linked into brie_amd/lib/libbrie_amd.so.
"""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libbrie_amd.so")
SOURCES = [os.path.join(CSRC, "brie_capi.hip"), os.path.join(CSRC, "brie_inst.hip"), os.path.join(CSRC, "brie_comm.hip"),
           os.path.join(CSRC, "brie_tile_inst.hip")]
HEADERS = [os.path.join(CSRC, "brie_kernels.hip.h"), os.path.join(CSRC, "brie_step_body.inc"), os.path.join(CSRC, "brie_launch.h"),
           os.path.join(CSRC, "brie_comm_internal.h"), os.path.join(CSRC, "brie_tile.hip.h"),
           os.path.join(ROOT, "include", "brie_amd.h")]
MAX_KC = 8


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(s) > t for s in SOURCES + HEADERS)


def compile_library(force=False, verbose=False, out=None, jobs=None, defines=None):
    """hipcc --offload-arch=gfx950 -> brie_amd/lib/libbrie_amd.so (cross-compiles without a GPU).

    `out`: write the library somewhere else (a clean rebuild to compare with the shipped one).
    `defines`: extra -D macros for every unit of a VARIANT build (needs `out`; e.g. ['BRIE_TILE_PROF=1'] for
    profiles/tile_phases.py, ['BRIE_LEANLOG_COND=false'] for an A/B of the MC_size 3 instantiations)."""
    if defines and out is None:
        raise ValueError("a variant build (defines=...) must not overwrite the shipped library: pass out=")
    if out is None and not force and not needs_build():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libbrie_amd.so")
    out = out or LIB_PATH
    os.makedirs(os.path.dirname(out), exist_ok=True)
    obj_dir = os.path.join(HERE, "build", os.path.splitext(os.path.basename(out))[0])
    os.makedirs(obj_dir, exist_ok=True)
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
            "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + ["-D" + d for d in (defines or [])]
    units = [(os.path.join(CSRC, "brie_capi.hip"), os.path.join(obj_dir, "brie_capi.o"), []),
             (os.path.join(CSRC, "brie_comm.hip"), os.path.join(obj_dir, "brie_comm.o"), [])]   # RCCL, bound by dlopen
    for mode in range(3):             # MFMA tile kernel of the wide designs: one unit per likelihood mode
        units.append((os.path.join(CSRC, "brie_tile_inst.hip"), os.path.join(obj_dir, "brie_tile_mode%d.o" % mode),
                      ["-DBRIE_TILE_MODE=%d" % mode]))
    for kc in range(MAX_KC + 1):
        units.append((os.path.join(CSRC, "brie_inst.hip"), os.path.join(obj_dir, "brie_inst_kc%d.o" % kc),
                      ["-DBRIE_KC=%d" % kc]))

    def cc(unit):
        src, obj, extra = unit
        cmd = base + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(cc, units))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", out]
    if verbose:
        print(" ".join(link))
    subprocess.run(link, check=True)
    return out


if __name__ == "__main__":
    print(compile_library(force=True, verbose="-v" in os.sys.argv))
