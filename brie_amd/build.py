"""Build the HIP shared library in-tree (gfx950 only)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libbrie_amd.so")
SOURCES = [os.path.join(HERE, "csrc", "brie_capi.hip")]
HEADERS = [os.path.join(HERE, "csrc", "brie_kernels.hip.h"), os.path.join(ROOT, "include", "brie_amd.h")]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(s) > t for s in SOURCES + HEADERS)


def compile_library(force=False, fast_math=None, verbose=False, out=None, defines=()):
    """hipcc --offload-arch=gfx950 -> brie_amd/lib/libbrie_amd.so (cross-compiles without a GPU).

    `out` / `defines` build tuning variants (e.g. -DBRIE_MIN_WAVES=4) next to the default library;
    select one at run time with the BRIE_AMD_LIB environment variable."""
    if out is None and not force and not needs_build():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libbrie_amd.so")
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-value", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc")]
    if fast_math is None:
        fast_math = os.environ.get("BRIE_FAST_MATH")
    if fast_math is not None:
        cmd.append("-DBRIE_FAST_MATH=%d" % int(fast_math))
    cmd += ["-D" + d for d in defines]
    out = out or LIB_PATH
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd += SOURCES + ["-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(compile_library(force=True, verbose=True))
