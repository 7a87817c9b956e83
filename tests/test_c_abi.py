"""The boundary from plain C: tests/c_abi/smoke.c is compiled with gcc against include/brie_amd.h and linked to
libbrie_amd.so (CPU: build + link only; `-m gpu`: run it).  No Python, no torch in that process."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "c_abi", "_build", "smoke")


def _build():
    from brie_amd.build import compile_library
    compile_library()
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    lib_dir = os.path.join(ROOT, "brie_amd", "lib")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", os.path.join(ROOT, "tests", "c_abi", "smoke.c"),
                    "-I" + os.path.join(ROOT, "include"), "-L" + lib_dir, "-lbrie_amd", "-lm",
                    "-Wl,-rpath," + lib_dir, "-Wl,-rpath-link,/opt/rocm/lib", "-o", EXE], check=True)
    return EXE


def test_c_client_compiles_and_links_against_the_header():
    exe = _build()
    out = subprocess.run(["nm", "-u", exe], check=True, capture_output=True, text=True).stdout
    for sym in ("brie_create", "brie_upload", "brie_step", "brie_read", "brie_destroy", "brie_last_error"):
        assert sym in out                                  # resolved from libbrie_amd.so at run time


@pytest.mark.gpu
def test_c_client_runs_on_the_gpu():
    exe = _build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "loss" in r.stdout and "mc_size" in r.stdout    # the refused call left its message in brie_last_error()
    print(r.stdout.strip())
