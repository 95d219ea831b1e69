#!/usr/bin/env python3
"""
Code-object provenance for libpyrovi.so: which device kernels of a build are the SAME code objects as ones that have run on
an MI355X with the GPU suite green, and which are not.

Round 5 showed that a different register assignment of an unchanged instruction sequence turned one kernel of this library from
bit-deterministic into random corruption, and then re-rolled 61 verified kernels without noticing.  A kernel's
identity is therefore its device ISA, not its source and not its name:

(`python tools/kernel_manifest.py ...` and `python -m pyro_amd.kernel_manifest ...` are the same program.)

    python tools/kernel_manifest.py build [--tree DIR] [-o manifest.json]   compile every unit of DIR/pyro_amd/csrc with
                      `hipcc -S --offload-device-only` (same flags as pyro_amd/_build.py), split per kernel, hash
    python tools/kernel_manifest.py diff A.json B.json                       kernels whose code differs / new / gone
    python tools/kernel_manifest.py check [manifest.json]                    this build against profiles/verified_kernels.json:
                      prints the kernels that are NOT verified code objects; exit code 1 if a production kernel is among them
    python tools/kernel_manifest.py bless manifest.json --commit C --evidence "..."   (on/after a green GPU run) add the
                      manifest's hashes to profiles/verified_kernels.json

Two hashes per kernel:
  exact  the instruction stream and the kernel descriptor (.amdhsa_* block) with comments stripped and basic-block label numbers
         renumbered in order of appearance -- everything that reaches the GPU except the position of the function in its unit;
  loose  the same with the byte offsets of scalar kernel-argument loads (`s_load_* sN, s[0:1], 0x..`) and the kernarg size
         blanked: a parameter block that gained a field moves those offsets and nothing else.
A kernel is VERIFIED when its exact hash is listed; `loose`-only matches are reported separately ("layout-only").
"""
import argparse
import concurrent.futures as cf
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILT = os.path.join(ROOT, "pyro_amd", "kernel_manifest.json")     # written by pyro_amd/_build.py next to libpyrovi.so
VERIFIED = os.path.join(ROOT, "profiles", "verified_kernels.json")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC"]
CXXFILT = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"


def compile_units(tree, outdir, extra=()):
    tree = os.path.abspath(tree)
    csrc = os.path.join(tree, "pyro_amd", "csrc")
    units = sorted(f for f in os.listdir(csrc) if f.endswith(".hip"))
    os.makedirs(outdir, exist_ok=True)

    def one(u):
        out = os.path.join(outdir, u[:-4] + ".s")
        cmd = ["hipcc"] + FLAGS + list(extra) + ["-S", "--offload-device-only", "-o", out, os.path.join(csrc, u)]
        r = subprocess.run(cmd, cwd=tree, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError("%s\n%s" % (" ".join(cmd), r.stderr[-2000:]))
        return out
    with cf.ThreadPoolExecutor(len(units)) as ex:
        return list(ex.map(one, units))


_LABEL = re.compile(r"\.L(BB|JTI|tmp|func_begin|func_end)(\d+)(_(\d+))?")
_KARG = re.compile(r"^(s_load_dword(?:x\d+)?\s+s(?:\[\d+:\d+\]|\d+),\s*s\[0:1\],\s*)(0x[0-9a-f]+|\d+)(.*)$")


def split_kernels(path):
    """{mangled name: [normalised lines]} for every function of one device assembly file: the body from its entry label to
    .Lfunc_end, the .amdhsa_kernel block included."""
    out, name, buf = {}, None, []
    for raw in open(path):
        line = raw.split(";", 1)[0].rstrip()
        s = line.strip()
        if name is None:
            m = re.match(r"^([A-Za-z_$][\w$.]*):\s*;\s*@", raw)
            if m:
                name, buf = m.group(1), []
            continue
        if re.match(r"^\.Lfunc_end\d+:", s):
            out[name] = buf
            name = None
            continue
        if not s or s.startswith((".p2align", ".section", ".text", ".size", ".type")):
            continue
        buf.append(s)
    return out


def normalise(lines, mangled):
    """Label numbers renumbered by first appearance; the function's own name replaced (it appears in the descriptor)."""
    ids = {}

    def rep(m):
        key = m.group(0)
        if key not in ids:
            ids[key] = "%s%d" % (m.group(1), len(ids))
        return ".L" + ids[key]
    exact = [_LABEL.sub(rep, ln.replace(mangled, "@K")) for ln in lines]
    loose = []
    for ln in exact:
        m = _KARG.match(ln)
        if m:
            ln = m.group(1) + "KARG" + m.group(3)
        elif ln.startswith(".amdhsa_kernarg_size"):
            ln = ".amdhsa_kernarg_size KARG"
        loose.append(ln)
    return exact, loose


def demangle(names):
    r = subprocess.run([CXXFILT], input="\n".join(names), capture_output=True, text=True, check=True)
    return dict(zip(names, r.stdout.split("\n")))


def short(dem):
    """`void k<...>(args)` -> `k<...>` (the form bench.norm_kernel and pvi_describe's kernel= use, spaces removed)."""
    n = dem[5:] if dem.startswith("void ") else dem
    depth = 0
    for i, ch in enumerate(n):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            n = n[:i]
            break
    return n.replace(" ", "")


def manifest_of(sfiles):
    man = {}
    for path in sfiles:
        unit = os.path.basename(path)[:-2]
        ks = split_kernels(path)
        dem = demangle(list(ks)) if ks else {}
        for mangled, lines in ks.items():
            if not any(ln.startswith(".amdhsa_kernel") for ln in lines):
                continue                                          # a device function that was not inlined, not a kernel
            exact, loose = normalise(lines, mangled)
            desc = {ln.split()[0][8:]: ln.split()[1] for ln in exact if ln.startswith(".amdhsa_") and len(ln.split()) == 2}
            ninstr = sum(1 for ln in exact if not ln.startswith(".") and not ln.endswith(":"))
            key = short(dem[mangled])
            if key in man:                                        # (overloads that differ only in arguments)
                key = key + "@" + mangled
            man[key] = {"unit": unit, "exact": hashlib.sha256("\n".join(exact).encode()).hexdigest()[:16],
                        "loose": hashlib.sha256("\n".join(loose).encode()).hexdigest()[:16], "instructions": ninstr,
                        "vgpr": int(desc.get("next_free_vgpr", 0)), "sgpr": int(desc.get("next_free_sgpr", 0)),
                        "lds": int(desc.get("group_segment_fixed_size", 0)), "kernarg": int(desc.get("kernarg_size", 0))}
    return man


def file_id(path):
    """{"sha256", "bytes"} of a file: the manifest written next to libpyrovi.so names the library it describes."""
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return {"sha256": h.hexdigest(), "bytes": os.path.getsize(path)}


def load_manifest(path, library=None):
    """The kernels of a manifest file (keys starting with "_" are metadata).  With `library`: {} unless the manifest was written
    for exactly that file ("_library": its sha256) -- file times do not survive every way a tree is copied to a GPU box."""
    man = json.load(open(path))
    meta = man.pop("_library", None)
    man = {k: v for k, v in man.items() if not k.startswith("_")}
    if library is not None and (meta is None or meta.get("sha256") != file_id(library)["sha256"]):
        return {}
    return man


# ---- the shipped binary against the assembly the manifest was hashed from ------------------------------------------------------------
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
OBJCOPY = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
_BRANCH = re.compile(r"^(s_branch|s_cbranch_\w+|s_call_b64|s_setpc_b64)\b")


def _instr(line):
    """One instruction in a form that the assembly printer and the disassembler share: no comments, single spaces, branch targets
    dropped (a label there, an address here); None for labels and directives."""
    line = line.split("//")[0].split(";")[0].strip()
    if not line or line.endswith(":") or line.startswith("."):
        return None
    line = re.sub(r"\s+", " ", line)
    m = _BRANCH.match(line)
    return m.group(1) if m else line


def code_objects(so_path, outdir):
    """The gfx950 ELF code objects inside a host library's .hip_fatbin section (one clang offload bundle per translation unit)."""
    import struct
    os.makedirs(outdir, exist_ok=True)
    fat = os.path.join(outdir, "fatbin.bin")
    # (objcopy with ONE file name rewrites that file in place: always name an output, and throw it away -- round 6 found the
    #  product library re-laid-out by its own provenance test, its sha256 no longer the manifest's)
    discard = os.path.join(outdir, "discard.so")
    subprocess.run([OBJCOPY, "--dump-section", ".hip_fatbin=" + fat, so_path, discard], check=True)
    os.remove(discard)
    d = open(fat, "rb").read()
    out = []
    for i, m in enumerate(re.finditer(b"\x7fELF", d)):
        o = m.start()
        e_shoff = struct.unpack_from("<Q", d, o + 0x28)[0]
        e_shentsize, e_shnum = struct.unpack_from("<HH", d, o + 0x3A)
        path = os.path.join(outdir, "co%d.elf" % i)
        open(path, "wb").write(d[o:o + e_shoff + e_shentsize * e_shnum])
        out.append(path)
    return out


def verify_binary(so_path, sfiles, outdir=None):
    """Is the device code inside `so_path` the code the manifest describes?  Disassembles every kernel of the library's code
    objects and compares the instruction stream with the assembly files (`-S --offload-device-only` of the same compilation) the
    manifest was hashed from.  Returns (kernels compared, [kernels that differ])."""
    outdir = outdir or os.path.join("/tmp", "pvi_verify_binary_%d" % os.getpid())
    dis = {}
    for elf in code_objects(so_path, outdir):
        cur = None
        for ln in subprocess.run([OBJDUMP, "-d", elf], capture_output=True, text=True, check=True).stdout.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
            if m:
                cur = m.group(1)
                dis[cur] = []
            elif cur is not None:
                x = _instr(ln)
                if x:
                    dis[cur].append(x)
    bad, n = [], 0
    for path in sfiles:
        for name, lines in split_kernels(path).items():
            if not any(ln.startswith(".amdhsa_kernel") for ln in lines):
                continue
            a = [x for x in (_instr(ln) for ln in lines if not ln.startswith((".amdhsa", ".end_amdhsa"))) if x]
            n += 1
            if name not in dis or dis[name][:len(a)] != a:      # (the disassembly pads a function to its alignment: cut there)
                bad.append(name)
    return n, bad


def build(tree=ROOT, outdir=None, extra=()):
    outdir = outdir or os.path.join("/tmp", "pvi_kernel_manifest_%d" % os.getpid())
    return manifest_of(compile_units(tree, outdir, extra))


def diff(a, b):
    """(changed, layout_only, new, gone): kernels of b against a."""
    changed = sorted(k for k in b if k in a and b[k]["exact"] != a[k]["exact"] and b[k]["loose"] != a[k]["loose"])
    layout = sorted(k for k in b if k in a and b[k]["exact"] != a[k]["exact"] and b[k]["loose"] == a[k]["loose"])
    return changed, layout, sorted(k for k in b if k not in a), sorted(k for k in a if k not in b)


def load_verified(path=VERIFIED):
    if not os.path.exists(path):
        return {"runs": [], "kernels": {}}
    return json.load(open(path))


def classify(man, ver=None):
    """{'verified': [...], 'layout_only': [...], 'unverified': [...]} of a manifest against the verified list."""
    ver = ver if ver is not None else load_verified()
    exact = {h for k in ver["kernels"].values() for h in k["exact"]}
    loose = {h for k in ver["kernels"].values() for h in k.get("loose", [])}
    out = {"verified": [], "layout_only": [], "unverified": []}
    for k, v in sorted(man.items()):
        out["verified" if v["exact"] in exact else "layout_only" if v["loose"] in loose else "unverified"].append(k)
    return out


def bless(man, commit, evidence, path=VERIFIED):
    ver = load_verified(path)
    run = len(ver["runs"])
    ver["runs"].append({"commit": commit, "evidence": evidence, "kernels": len(man)})
    for k, v in man.items():
        e = ver["kernels"].setdefault(k, {"exact": [], "loose": [], "runs": []})     # runs[i]: the runs hash i was part of
        if v["exact"] not in e["exact"]:
            e["exact"].append(v["exact"])
            e["loose"].append(v["loose"])
            e["runs"].append([run])
        else:
            e["runs"][e["exact"].index(v["exact"])].append(run)
    json.dump(ver, open(path, "w"), indent=0, sort_keys=True)
    return ver


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    b = sub.add_parser("build")
    b.add_argument("--tree", default=ROOT)
    b.add_argument("-o", "--out", default=None)
    b.add_argument("-D", action="append", default=[])
    b.add_argument("--asm", default=None, help="directory of *.s files already compiled (skips hipcc)")
    d = sub.add_parser("diff")
    d.add_argument("a")
    d.add_argument("b")
    c = sub.add_parser("check")
    c.add_argument("manifest", nargs="?")
    c.add_argument("--opt-in", default=os.path.join(ROOT, "profiles", "optin_kernels.txt"))
    bl = sub.add_parser("bless")
    bl.add_argument("manifest")
    bl.add_argument("--commit", required=True)
    bl.add_argument("--evidence", required=True)
    bl.add_argument("--only", default=None, help="file with one kernel name per line (e.g. the kernels a trace saw): bless these only")
    bl.add_argument("--file", default=VERIFIED, help="the verified list to extend (default profiles/verified_kernels.json)")
    a = ap.parse_args()
    if a.cmd == "build":
        if a.asm:
            man = manifest_of(sorted(os.path.join(a.asm, f) for f in os.listdir(a.asm) if f.endswith(".s")))
        else:
            man = build(a.tree, extra=["-D" + x for x in a.D])
        text = json.dumps(man, indent=0, sort_keys=True)
        if a.out:
            open(a.out, "w").write(text)
        print("%d kernels%s" % (len(man), " -> " + a.out if a.out else ""))
        if not a.out:
            print(text)
    elif a.cmd == "diff":
        A, B = load_manifest(a.a), load_manifest(a.b)
        ch, lay, new, gone = diff(A, B)
        for tag, ks in (("CHANGED", ch), ("LAYOUT-ONLY", lay), ("NEW", new), ("GONE", gone)):
            for k in ks:
                extra = ""
                if k in A and k in B:
                    extra = "  (%d -> %d instructions, vgpr %d -> %d)" % (A[k]["instructions"], B[k]["instructions"], A[k]["vgpr"], B[k]["vgpr"])
                print("%-11s %s%s" % (tag, k, extra))
        print("%d identical, %d changed, %d layout-only, %d new, %d gone" % (
            len([k for k in B if k in A and A[k]["exact"] == B[k]["exact"]]), len(ch), len(lay), len(new), len(gone)))
    elif a.cmd == "check":
        man = load_manifest(a.manifest) if a.manifest else build()
        cl = classify(man)
        optin = optin_patterns(a.opt_in)
        bad = 0
        for tag in ("layout_only", "unverified"):
            for k in cl[tag]:
                o = is_optin(k, optin)
                bad += not o
                print("%-11s %s%s" % (tag.upper(), k, "   [opt-in]" if o else ""))
        print("%d verified, %d layout-only, %d unverified (%d of them not opt-in)" % (
            len(cl["verified"]), len(cl["layout_only"]), len(cl["unverified"]), bad))
        sys.exit(1 if bad else 0)
    else:
        man = load_manifest(a.manifest)
        if a.only:
            seen = {short(ln.strip()) for ln in open(a.only) if ln.strip()}
            man = {k: v for k, v in man.items() if k in seen}
        else:       # (library-level evidence covers what a default call can launch, not the opt-in kernels)
            pats = optin_patterns()
            man = {k: v for k, v in man.items() if not is_optin(k, pats)}
        if a.file != VERIFIED and not os.path.exists(a.file) and os.path.exists(VERIFIED):
            shutil.copy(VERIFIED, a.file)
        ver = bless(man, a.commit, a.evidence, a.file)
        print("%d kernels listed, %d runs" % (len(ver["kernels"]), len(ver["runs"])))


def optin_patterns(path=os.path.join(ROOT, "profiles", "optin_kernels.txt")):
    """Regular expressions (one per line, # comments) of kernels that only run when a caller asks for an opt-in mode."""
    if not os.path.exists(path):
        return []
    pats = [ln.split("#", 1)[0].strip() for ln in open(path)]      # (`#` starts a comment anywhere in a line)
    return [re.compile(p) for p in pats if p]


def is_optin(kernel, pats):
    return any(p.search(kernel) for p in pats)


if __name__ == "__main__":
    main()
