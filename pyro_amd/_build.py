"""Builds pyro_amd/libpyrovi.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
SRC = [os.path.join(PKG, "csrc", "pyrovi.hip")]
HDR = [os.path.join(ROOT, "include", "pyrovi.h")]


def sources():
    """Every file the library is compiled from: the translation unit, the kernel files it includes, the C ABI header."""
    csrc = os.path.join(PKG, "csrc")
    return sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".inc", ".h"))) + HDR
OUT = os.path.join(PKG, "libpyrovi.so")
# -ffp-contract=off: the f64 kernels mirror the reference's NumPy arithmetic (no implicit FMA)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (needed to build libpyrovi.so for gfx950)")
    return exe


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(p) <= t for p in sources())


def build(force=False, verbose=True):
    if not force and up_to_date():
        return OUT
    cmd = [hipcc()] + FLAGS + ["-o", OUT] + SRC
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=ROOT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", OUT)
