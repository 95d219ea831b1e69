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
# -fno-slp-vectorize: the pairing pass packs independent scalar float32 steps of the action loops into v_pk_* pairs that
#   need their halves transposed first (15 register moves per four cells of the 4-D loop); where packed math pays, the
#   kernels spell it out on 2-vectors themselves
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared"]


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


OUT_SAN = os.path.join(PKG, "libpyrovi_ubsan.so")


def sanitizer_runtime():
    """Path of clang's shared UndefinedBehaviorSanitizer runtime (to LD_PRELOAD under python)."""
    clang = "/opt/rocm/lib/llvm/bin/clang"
    return subprocess.run([clang, "-print-file-name=libclang_rt.ubsan_standalone-x86_64.so"], capture_output=True,
                          text=True, check=True).stdout.strip()


def build_sanitized(force=False, verbose=True):
    """The same translation unit with the HOST side of the C ABI instrumented (device code untouched):
    pyro_amd/libpyrovi_ubsan.so, loaded instead of the product library when PYROVI_LIB points to it.
    UndefinedBehaviorSanitizer with bounds checks, every report fatal.  (AddressSanitizer was the first choice, and the
    library builds with it -- but ROCm's ASan runtime intercepts hsa_amd_memory_pool_allocate to put DEVICE memory under
    its allocator and aborts with "out of memory" on the GPU boxes, with or without HSA_XNACK=1: it cannot be preloaded
    under python there.)"""
    if not force and os.path.exists(OUT_SAN) and all(os.path.getmtime(p) <= os.path.getmtime(OUT_SAN) for p in sources()):
        return OUT_SAN
    cmd = [hipcc(), "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
           "-fsanitize=undefined,bounds", "-fno-sanitize=vptr,function", "-fno-sanitize-recover=undefined",
           "-fno-gpu-sanitize", "-shared-libsan", "-o", OUT_SAN] + SRC
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=ROOT)
    return OUT_SAN


def build(force=False, verbose=True):
    if not force and up_to_date():
        return OUT
    cmd = [hipcc()] + FLAGS + ["-o", OUT] + SRC
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=ROOT)
    return OUT


def build_all(force=False, verbose=True):
    """Product and sanitized library side by side (two hipcc processes: the translation unit compiles in ~75 s / ~95 s, one
    after the other they were the 3 minutes of every build)."""
    need = force or not up_to_date()
    need_san = force or not (os.path.exists(OUT_SAN) and all(os.path.getmtime(p) <= os.path.getmtime(OUT_SAN) for p in sources()))
    cmds = []
    if need:
        cmds.append([hipcc()] + FLAGS + ["-o", OUT] + SRC)
    if need_san:
        cmds.append([hipcc(), "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
                     "-shared", "-fsanitize=undefined,bounds", "-fno-sanitize=vptr,function", "-fno-sanitize-recover=undefined",
                     "-fno-gpu-sanitize", "-shared-libsan", "-o", OUT_SAN] + SRC)
    procs = []
    for cmd in cmds:
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd, cwd=ROOT)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    return OUT, OUT_SAN


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", OUT)
    if "--no-sanitized" not in sys.argv:
        print("built", build_sanitized(force="--force" in sys.argv))
