#!/usr/bin/env python3
"""
TEST INFRASTRUCTURE.  Builds tests/emu/_build/libpyrovi_emu.so: the SOURCES of libpyrovi (pyro_amd/csrc/*, include/pyrovi.h)
compiled as ordinary C++ for the host against the HIP emulation of tests/emu/include/hip/hip_runtime.h + tests/emu/emu_runtime.cpp.
The product sources are not touched and carry no emulation switch: a copy is transformed textually --

  * `asm volatile(...)` statements (GCN wait counts, register-class constraints that keep a value live) are dropped;
  * `extern __shared__ ... T name[];` becomes a pointer to the workgroup's emulated dynamic LDS;
  * `__attribute__((address_space(N)))` is dropped (LDS addresses are host pointers below 16 MiB in the emulation);
  * `(const void*)kernel<...>` (a kernel taken as an untyped pointer for a cooperative launch) becomes emu::coop_thunk(&kernel<...>);

-- and everything else (builtins, vector types, the runtime API, launches) is supplied by the header.  Nothing in pyro_amd/ loads
this library; tests/test_emu_cpu.py points PYROVI_LIB at it in subprocesses.
"""
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "pyro_amd", "csrc")
BUILD = os.path.join(HERE, "_build")
OUT = os.path.join(BUILD, "libpyrovi_emu.so")
RCCL = os.path.join(BUILD, "librccl_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
UNITS = ["pyrovi.hip", "f64.hip", "lean.hip"]


LDSTRACE = False     # (build(ldstrace=True): the gathers of the 4-D window sweep report their LDS addresses)


def transform(text):
    if LDSTRACE:     # sweep_lean4.inc lean4_gather: g.r[k] = *(lds_vpair*)(size_t)(ADDR);
        text = re.sub(r'\*\(lds_vpair\*\)\(size_t\)\(([^;]+)\);', r'emu::lds_rd<v2f>(\1);', text)
    # asm statements: whole statement up to the terminating ");"
    text = re.sub(r'asm\s+volatile\s*\((?:[^;"]|"[^"]*")*\)\s*;', "/* asm */ ;", text)
    # dynamic LDS declarations
    text = re.sub(r'extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?([A-Za-z_][\w ]*?)\s+(\w+)\[\];',
                  r'\1* \2 = (\1*)emu::g_blk->dyn_lds;', text)
    text = re.sub(r'__attribute__\(\(address_space\(\d+\)\)\)', "", text)
    # kernels taken as untyped function pointers (cooperative launches): record how to call them
    text = re.sub(r'\(const void\*\)(k_\w+<[^<>;]*>)', r'emu::coop_thunk(&\1)', text)
    return text


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [os.path.join(ROOT, "include", "pyrovi.h"),
                                                                      os.path.join(HERE, "emu_runtime.cpp"), os.path.join(HERE, "rccl_emu.cpp"),
                                                                      os.path.join(HERE, "include", "hip", "hip_runtime.h"), __file__]


def up_to_date():
    return os.path.exists(OUT) and os.path.exists(RCCL) and all(os.path.getmtime(p) <= os.path.getmtime(OUT) for p in sources())


def asan_runtime():
    return subprocess.run([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True, check=True).stdout.strip()


def build(force=False, verbose=False, opt="-O1", asan=False, ldstrace=False):
    """asan=True: the same library under AddressSanitizer (_build/libpyrovi_emu_asan.so; LD_PRELOAD asan_runtime() under python):
    "device" memory is the sanitizer's heap, so a kernel that reads or writes one element beyond an allocation is reported with
    the kernel's source line.  (ROCm's own ASan runtime cannot be preloaded on the GPU boxes: DESIGN.md 4.7.)"""
    global LDSTRACE
    LDSTRACE = bool(ldstrace)
    if ldstrace:
        return _build(os.path.join(BUILD, "libpyrovi_emu_ldstrace.so"), "ldstrace", [], force, verbose, opt)
    if asan:
        return _build(os.path.join(BUILD, "libpyrovi_emu_asan.so"), "asan", ["-fsanitize=address", "-shared-libasan"], force, verbose, opt)
    return _build(OUT, "", [], force, verbose, opt)


def _build(OUT, tag, extra, force, verbose, opt):
    if os.path.exists(OUT) and os.path.exists(RCCL) and all(os.path.getmtime(p) <= os.path.getmtime(OUT) for p in sources()) and not force:
        return OUT
    src = os.path.join(BUILD, "pyro_amd" + ("_" + tag if tag else ""), "csrc")
    os.makedirs(src, exist_ok=True)
    os.makedirs(os.path.join(BUILD, "include"), exist_ok=True)
    shutil.copy(os.path.join(ROOT, "include", "pyrovi.h"), os.path.join(BUILD, "include", "pyrovi.h"))
    for f in os.listdir(CSRC):
        with open(os.path.join(CSRC, f)) as fi, open(os.path.join(src, f.replace(".hip", ".cpp")), "w") as fo:
            fo.write(transform(fi.read()))
    flags = [opt, "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-fno-strict-aliasing", "-fno-omit-frame-pointer", "-mno-omit-leaf-frame-pointer", "-pthread",
             "-I" + os.path.join(HERE, "include"), "-Wno-unknown-attributes", "-Wno-ignored-attributes", "-Wno-unused-value",
             "-Wno-deprecated-declarations", "-DPVI_EMU_BUILD=1"] + extra
    objs, procs = [], []
    for u in UNITS + ["emu_runtime.cpp"]:
        s = os.path.join(src, u.replace(".hip", ".cpp")) if u != "emu_runtime.cpp" else os.path.join(HERE, u)
        o = os.path.join(BUILD, os.path.basename(s) + tag + ".o")
        objs.append(o)
        cmd = [CLANG] + flags + ["-c", "-o", o, s]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd, cwd=src, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for cmd, p in procs:
        out = p.communicate()[0]
        if p.returncode:
            failed = True
            sys.stderr.write(out[-6000:] if not verbose else out)
    if failed:
        raise RuntimeError("emulated build failed")
    cmd = [CLANG, "-shared", "-fPIC", "-pthread", "-o", OUT] + extra + objs
    subprocess.run(cmd, check=True)
    # the stand-in for librccl.so (PVI_RCCL_LIB): rank processes of one machine over a shared-memory segment
    subprocess.run([CLANG, "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I" + os.path.join(HERE, "include"), "-o", RCCL,
                    os.path.join(HERE, "rccl_emu.cpp"), "-lrt"], check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, asan="--asan" in sys.argv, ldstrace="--ldstrace" in sys.argv))
