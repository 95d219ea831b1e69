"""Builds pyro_amd/libpyrovi.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
# The library is three translation units over two shared headers (core.h: device-side common code; host.h: the handle, helpers,
# cross-unit entry points).  They compile side by side (~30 s instead of the 75 s of the one-file build of rounds 1-4) and an
# edit recompiles only the units that include the file:
UNITS = {
    "pyrovi": ["pyrovi.hip", "sweep_spline.inc", "shard.inc"],   # C ABI, exact / table / spline / n = 3 kernels, rollouts, sharding
    "f64": ["f64.hip"],                                          # k_sweep64 family, k_sweep64m
    "lean": ["lean.hip", "sweep_lean.inc", "sweep_lean4.inc"],   # float32 LDS-window families + k_sweep_fast, their set-up
}
COMMON = [os.path.join(CSRC, "core.h"), os.path.join(CSRC, "host.h")]
SRC = [os.path.join(CSRC, u[0]) for u in UNITS.values()]
HDR = [os.path.join(ROOT, "include", "pyrovi.h")]
OBJ = os.path.join(PKG, "_obj")


def sources():
    """Every file the library is compiled from: the translation units, the kernel files they include, the headers."""
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".inc", ".h"))) + HDR
OUT = os.path.join(PKG, "libpyrovi.so")
# -ffp-contract=off: the f64 kernels mirror the reference's NumPy arithmetic (no implicit FMA)
# -fno-slp-vectorize: the pairing pass packs independent scalar float32 steps of the action loops into v_pk_* pairs that
#   need their halves transposed first (15 register moves per four cells of the 4-D loop); where packed math pays, the
#   kernels spell it out on 2-vectors themselves
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC"]
FLAGS_SAN = ["--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
             "-fsanitize=undefined,bounds", "-fno-sanitize=vptr,function", "-fno-sanitize-recover=undefined", "-fno-gpu-sanitize"]


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
    return _make(OUT_SAN, "san", FLAGS_SAN, ["-shared", "-fPIC", "-fsanitize=undefined,bounds", "-shared-libsan"], force, verbose)


def _stale(obj, unit):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [os.path.join(CSRC, f) for f in UNITS[unit]] + COMMON + HDR
    return any(os.path.getmtime(p) > t for p in deps)


def _compile_jobs(tag, flags, force, asm=False):
    """The compilations that are due.  asm: next to each object the unit's device assembly (same flags, `-S
    --offload-device-only`) -- what pyro_amd/kernel_manifest.py hashes per kernel."""
    os.makedirs(OBJ, exist_ok=True)
    jobs = []
    for unit in UNITS:
        obj = os.path.join(OBJ, "%s.%s.o" % (unit, tag))
        src = os.path.join(CSRC, UNITS[unit][0])
        if force or _stale(obj, unit):
            jobs.append([hipcc()] + flags + ["-c", "-o", obj, src])
        if asm and (force or _stale(os.path.join(OBJ, "%s.s" % unit), unit)):
            jobs.append([hipcc()] + flags + ["-S", "--offload-device-only", "-o", os.path.join(OBJ, "%s.s" % unit), src])
    return jobs


MANIFEST = os.path.join(PKG, "kernel_manifest.json")


def write_manifest(verbose=True):
    """pyro_amd/kernel_manifest.json: per device kernel of THIS build the hash of its ISA (pyro_amd/kernel_manifest.py).  It
    travels with libpyrovi.so (git-ignored like it); bench.py and the tests read it to say which kernels are verified code
    objects (profiles/verified_kernels.json) and which kernel a set of PMC counters was taken with."""
    import json
    from pyro_amd import kernel_manifest
    sfiles = [os.path.join(OBJ, "%s.s" % u) for u in UNITS]
    if os.path.exists(MANIFEST) and all(os.path.getmtime(f) <= os.path.getmtime(MANIFEST) for f in sfiles + [OUT]):
        return MANIFEST
    man = kernel_manifest.manifest_of(sfiles)
    man["_library"] = kernel_manifest.file_id(OUT)        # (the library this manifest describes: bench.py checks it)
    with open(MANIFEST, "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)
    if verbose:
        print("wrote %s (%d kernels)" % (MANIFEST, len(man) - 1), flush=True)
    return MANIFEST


def _run_all(cmds, verbose):
    procs = []
    for cmd in cmds:
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd, cwd=ROOT)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)


def _link(out, tag, link_flags, verbose):
    cmd = [hipcc(), "--offload-arch=gfx950"] + link_flags + ["-o", out] + [os.path.join(OBJ, "%s.%s.o" % (u, tag)) for u in UNITS]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=ROOT)


def _make(out, tag, flags, link_flags, force, verbose):
    jobs = _compile_jobs(tag, flags, force)
    if not jobs and os.path.exists(out) and all(os.path.getmtime(os.path.join(OBJ, "%s.%s.o" % (u, tag))) <= os.path.getmtime(out) for u in UNITS):
        return out
    _run_all(jobs, verbose)
    _link(out, tag, link_flags, verbose)
    return out


def build(force=False, verbose=True):
    _run_all(_compile_jobs("o3", FLAGS, force, asm=True), verbose)
    out = _make(OUT, "o3", FLAGS, ["-shared", "-fPIC"], False, verbose)
    write_manifest(verbose)
    return out


def build_variant(name, defines=(), extra_flags=(), force=False, verbose=True):
    """An experiment build next to the product library: pyro_amd/libpyrovi_<name>.so compiled with -D<define> ... (A/B runs on
    one box select it with PYROVI_LIB).  Not part of build(): tools/ scripts call it before a gpurun."""
    out = os.path.join(PKG, "libpyrovi_%s.so" % name)
    flags = FLAGS + ["-D" + d for d in defines] + list(extra_flags)
    return _make(out, "x_" + name, flags, ["-shared", "-fPIC"], force, verbose)


def build_all(force=False, verbose=True):
    """Product and sanitized library: all six compilations side by side, then the two links."""
    jobs = _compile_jobs("o3", FLAGS, force, asm=True) + _compile_jobs("san", FLAGS_SAN, force)
    _run_all(jobs, verbose)
    build(False, verbose)
    build_sanitized(False, verbose)
    return OUT, OUT_SAN


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", OUT)
    if "--no-sanitized" not in sys.argv:
        print("built", build_sanitized(force="--force" in sys.argv))
