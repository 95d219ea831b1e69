"""
Code-object provenance (pyro_amd/kernel_manifest.py).  Round 5 showed that ONE other register assignment of an
unchanged instruction sequence made a kernel of this library compute random results, and then re-rolled 61 verified kernels by
editing a shared template.  These tests hold the build to the list of code objects that have passed the GPU suite on an MI355X
(profiles/verified_kernels.json): a kernel a default call can launch must be one of them, bit for bit of its ISA; kernels that
are not must be opt-in (profiles/optin_kernels.txt).  After a green GPU run of a changed kernel:
`python tools/kernel_manifest.py bless pyro_amd/kernel_manifest.json --commit <c> --evidence "<what ran>"`.
"""
import json
import os

import pytest

from conftest import ROOT
from pyro_amd import kernel_manifest as KM

ASM = r"""
	.text
	.protected	_Z3foov
	.globl	_Z3foov
	.p2align	8
	.type	_Z3foov,@function
_Z3foov:                                ; @_Z3foov
; %bb.0:
	s_load_dword s4, s[0:1], 0x10
	s_load_dwordx2 s[2:3], s[0:1], 0x8     ; a comment
	s_cbranch_scc1 .LBB7_2
.LBB7_1:
	v_mov_b32_e32 v0, 0
.LBB7_2:
	s_endpgm
	.section	.rodata,"a",@progbits
	.amdhsa_kernel _Z3foov
		.amdhsa_kernarg_size 24
		.amdhsa_next_free_vgpr 1
		.amdhsa_next_free_sgpr 5
	.end_amdhsa_kernel
	.text
.Lfunc_end7:
	.size	_Z3foov, .Lfunc_end7-_Z3foov
"""


def _manifest(tmp_path, text, name="unit.s"):
    p = tmp_path / name
    p.write_text(text)
    return KM.manifest_of([str(p)])


def test_hashes_ignore_position_and_comments_and_see_every_instruction(tmp_path):
    a = _manifest(tmp_path, ASM)
    assert list(a) == ["foo"] and a["foo"]["vgpr"] == 1 and a["foo"]["kernarg"] == 24 and a["foo"]["instructions"] == 5
    # another position in the unit (label numbers), other comments: the same code object
    b = _manifest(tmp_path, ASM.replace(".LBB7_", ".LBB31_").replace("func_end7", "func_end31").replace("; a comment", "; another"))
    assert b["foo"]["exact"] == a["foo"]["exact"]
    # a kernel argument moved: layout-only (loose hash equal, exact hash not)
    c = _manifest(tmp_path, ASM.replace("0x10", "0x18").replace("kernarg_size 24", "kernarg_size 32"))
    assert c["foo"]["exact"] != a["foo"]["exact"] and c["foo"]["loose"] == a["foo"]["loose"]
    # another register in one instruction: a different code object in both senses
    d = _manifest(tmp_path, ASM.replace("v_mov_b32_e32 v0, 0", "v_mov_b32_e32 v1, 0"))
    assert d["foo"]["exact"] != a["foo"]["exact"] and d["foo"]["loose"] != a["foo"]["loose"]
    # ... and so is another register budget in the descriptor
    e = _manifest(tmp_path, ASM.replace("next_free_vgpr 1", "next_free_vgpr 2"))
    assert e["foo"]["exact"] != a["foo"]["exact"]
    ch, lay, new, gone = KM.diff(a, c)
    assert (ch, lay, new, gone) == ([], ["foo"], [], [])
    assert KM.diff(a, d)[0] == ["foo"]


def test_classify_and_bless(tmp_path):
    a = _manifest(tmp_path, ASM)
    path = str(tmp_path / "verified.json")
    assert KM.classify(a, KM.load_verified(path))["unverified"] == ["foo"]
    KM.bless(a, "abc1234", "a test", path)
    ver = KM.load_verified(path)
    assert KM.classify(a, ver)["verified"] == ["foo"] and ver["runs"][0]["commit"] == "abc1234"
    c = _manifest(tmp_path, ASM.replace("0x10", "0x18"))
    assert KM.classify(c, ver)["layout_only"] == ["foo"]
    d = _manifest(tmp_path, ASM.replace("v_mov_b32_e32 v0, 0", "v_mov_b32_e32 v1, 0"))
    assert KM.classify(d, ver)["unverified"] == ["foo"]
    KM.bless(d, "def5678", "another", path)                      # a kernel may be verified in several forms over the rounds
    ver = KM.load_verified(path)
    assert KM.classify(a, ver)["verified"] == ["foo"] and KM.classify(d, ver)["verified"] == ["foo"] and len(ver["runs"]) == 2


def test_opt_in_patterns():
    pats = KM.optin_patterns()
    assert pats, "profiles/optin_kernels.txt"
    for k in ("k_sweep_leanfb<1,unsignedchar,true>", "k_sweep_leanm<4,unsignedshort,false>", "k_sweep_lean4<12,unsignedchar,true,true>",
              "k_lean4_node<12>", "k_sweep3_fast<7,unsignedchar,float*,double>", "k_sweep<12,float,unsignedchar,true,false>",
              "k_sweep_lean4fbc<2,unsignedchar,true,true>"):
        assert KM.is_optin(k, pats), k
    for k in ("k_sweep_lean4<2,unsignedchar,true,true>", "k_sweep_lean4fb<2,unsignedchar,true,true>", "k_sweep3_fast<7,unsignedchar>",
              "k_sweep_lean<1,unsignedchar,true,1,64>", "k_sweep64<3,unsignedchar,true,true,true>", "k_sweep<2,float,unsignedchar,true,false>",
              "k_sweep_tablep<2,float,unsignedchar>", "k_lean_setup<1>", "k_sweep3<11,float,unsignedchar>"):
        assert not KM.is_optin(k, pats), k


def test_every_kernel_a_default_call_can_launch_is_a_verified_code_object():
    """The build's manifest (written by pyro_amd/_build.py from the same compilation as libpyrovi.so) against the verified list."""
    from pyro_amd import _build
    _build.build(verbose=False)                                   # (no-op when the library and its manifest are current)
    man = KM.load_manifest(_build.MANIFEST, library=_build.OUT)      # ({} if the manifest were not this library's)
    assert len(man) > 400
    cl = KM.classify(man)
    pats = KM.optin_patterns()
    bad = [k for k in cl["layout_only"] + cl["unverified"] if not KM.is_optin(k, pats)]
    assert not bad, ("kernels that are neither verified code objects nor opt-in -- run the GPU suite on this build and bless it, or "
                     "restore their ISA (python tools/kernel_manifest.py check): %s" % bad[:10])
    # the headline kernels are among the verified ones, by name
    for k in ("k_sweep_lean4<2,unsignedchar,true,true>", "k_sweep_lean4fb<2,unsignedchar,true,true>", "k_sweep_lean<1,unsignedchar,true,1,64>",
              "k_sweep64<3,unsignedchar,true,true,true>", "k_sweep64m<1,unsignedchar,1>", "k_sweep3_fast<7,unsignedchar>"):
        assert k in cl["verified"], k
    # and the opt-in set is what the opt-in file says, nothing hidden behind it
    optin = [k for k in man if KM.is_optin(k, pats)]
    assert len(optin) < 0.2 * len(man), (len(optin), len(man))
    assert not [k for k in cl["verified"] if KM.is_optin(k, pats)]     # (an opt-in pattern never covers a kernel that is in production)


def test_verified_list_names_its_evidence():
    ver = KM.load_verified()
    assert ver["runs"] and all(r["commit"] and r["evidence"] for r in ver["runs"])
    for k, e in list(ver["kernels"].items())[:50]:
        assert len(e["exact"]) == len(e["loose"]) == len(e["runs"]) and all(0 <= r < len(ver["runs"]) for rs in e["runs"] for r in rs), k


def test_the_manifest_describes_the_device_code_inside_the_library():
    """The manifest is hashed from `hipcc -S --offload-device-only`, the library is linked from `hipcc -c` of the same sources with the
    same flags: two compilations.  This closes the loop -- the code objects are extracted from libpyrovi.so's .hip_fatbin, every
    kernel is disassembled, and its instruction stream must be the assembly's, instruction for instruction (branch targets aside)."""
    from pyro_amd import _build
    _build.build(verbose=False)
    sfiles = [os.path.join(_build.OBJ, "%s.s" % u) for u in _build.UNITS]
    if not all(os.path.exists(f) for f in sfiles) or not os.path.exists(KM.OBJDUMP):
        pytest.skip("no device assembly / llvm-objdump here")
    before = KM.file_id(_build.OUT)
    n, bad = KM.verify_binary(_build.OUT, sfiles)
    assert n > 400 and not bad, (n, bad[:10])
    # ... and looking did not touch it (round 6: `objcopy --dump-section` with one file name re-wrote the library in place -- same
    # code, another layout and sha256, so the manifest no longer described the file that travels to the GPU box)
    assert KM.file_id(_build.OUT) == before
    assert KM.load_manifest(_build.MANIFEST, library=_build.OUT)
