"""CPU-side checks of bench.py: the cpu_baseline leg (oracle C twin, bounded sample) and the helpers around it."""
import contextlib
import io
import os
import sys

import numpy as np
import pytest

from conftest import ROOT


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_cpu_baseline_whole_sweeps_and_slice_modes():
    bench = _bench()
    from oracle import vi_oracle as O
    from pyro_amd import configs
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build("pendulum:61,61:5:float32")
    rec, J, n, rows, _twin = bench.cpu_baseline(cfg, budget_s=3.0)
    assert rec["kind"] == "port" and rec["unit"] == "cells/s" and rec["value"] > 0 and rec["per_core_value"] > 0
    assert 1 <= rec["cores"] <= (os.cpu_count() or 1) and rows is None and n >= 1
    assert "whole sweeps" in rec["sample"] and 0 < rec["efficiency"] < 4
    p = bench.oracle_problem(cfg)
    Jr = O.terminal_cost(p)
    for _ in range(n):
        Jr, _ = O.sweep(p, Jr)
    assert np.array_equal(J, Jr)                         # the timed sweeps are the oracle's sweeps, bit for bit
    # a budget too small for whole sweeps falls back to a slice of one sweep from J0
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build("cartpole:21,21,21,21:7:float32")
    rec, Js, n, rows, _twin = bench.cpu_baseline(cfg, budget_s=0.05)
    assert n == 1 and rows is not None and "nodes [" in rec["sample"]
    p = bench.oracle_problem(cfg)
    Jr, _ = O.sweep(p, O.terminal_cost(p), ids=np.arange(rows[0], rows[1]))
    assert np.array_equal(Js, Jr)


def test_usable_cpus_and_counter_sources():
    bench = _bench()
    assert 1 <= bench.usable_cpus() <= (os.cpu_count() or 1)
    import json
    ctr = json.load(open(os.path.join(ROOT, "profiles", "counters.json")))
    for name, rec in ctr.items():
        if isinstance(rec, dict):
            assert "source" in rec and "profiles/" in rec["source"], name      # counters are labelled, never anonymous


def test_the_driver_parsed_line_is_compact():
    """round 3's 26.5 KB line could not be parsed by the driver.  The headline built from the full records of
    round 3 (N = 1 default run with every secondary, the N > 1 harness lines) must stay below 4 KB, load as JSON and carry
    the contract keys, the roofline and the CPU baseline."""
    import json
    from pyro_amd import benchline
    for name in ("r03_bench_default.json", "r03_bench_world1.json", "r03_bench_world1_c4.json"):
        full = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert len(json.dumps(full)) > 8000 or "world1" in name
        # fields this round adds to the full record
        full["jstar_rel_err_vs_cpu_on"] = "c1 float64 618 sweeps from J0 (solved to tol 0.1): 1.00e-13; " * 3
        full["jstar_ok"] = True
        line = json.dumps(benchline.compact_line(full, "gpurun_out/bench_full.json"))
        assert len(line) < 4096, (name, len(line))
        rec = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline"):
            assert k in rec, (name, k)
        assert rec["config"]["workload"] and "model" not in rec["config"]
        assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(rec["roofline"])
        assert "cands=" not in line
        if name == "r03_bench_default.json":
            assert rec["cpu_baseline"]["kind"] == "port" and rec["cpu_baseline"]["cores"] >= 1 and rec["cpu_baseline"]["sample"]
            assert set(rec["secondary"]) >= {"c1", "c2", "c2p", "c4", "c5"}
            assert all("ms_per_step" in v for v in rec["secondary"].values())


def test_emit_prints_one_line_last_on_stdout(tmp_path, capfd):
    import json
    from pyro_amd import benchline
    full = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_default.json")))
    benchline.emit(full)
    out, err = capfd.readouterr()
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096
    assert json.loads(lines[-1])["metric"] == full["metric"]
    assert "bench.py full record: {" in err                      # everything else: stderr (and gpurun_out/bench_full.json)


def test_counters_are_tied_to_the_kernel_name():
    """the digest names the production kernel (the template instantiation pvi_describe reports), and
    bench.py drops counters whose kernel is not the one the run launched."""
    import json
    bench = _bench()
    assert bench.norm_kernel("void k_sweep64<3, unsigned char, true, true, true>(DevP, double const*)") == \
        "k_sweep64<3,unsignedchar,true,true,true>"
    assert bench.norm_kernel("k_sweep64<3,unsignedchar,true,true,true>") == "k_sweep64<3,unsignedchar,true,true,true>"
    ctr = json.load(open(os.path.join(ROOT, "profiles", "counters.json")))
    # C5 / C5d: the sparse patch walk and the dense patch walk, not set-up's slowest candidate
    assert bench.norm_kernel(ctr["c5"]["kernel"]) == "k_sweep64<3,unsignedchar,true,true,true>"
    assert bench.norm_kernel(ctr["c5d"]["kernel"]) == "k_sweep64<3,unsignedchar,true,true,false>"
    c = dict(ctr["c5"], kernel_path="path=exact-f64v2 mapping=patch8x8 sparse=1", isa_hash=bench.kernel_isa_hash(ctr["c5"]["kernel"]))
    good = "path=exact-f64v2 kernel=k_sweep64<3,unsignedchar,true,true,true> mapping=patch8x8 off32=1 sparse=1 inbox=0.1 note="
    kept, err = bench.check_counters(c, good)
    assert kept and err is None
    kept, err = bench.check_counters(c, good.replace("true,true,true", "true,false,false"))
    assert not kept and "kernel:" in err
    kept, err = bench.check_counters(c, good.replace(" kernel=k_sweep64<3,unsignedchar,true,true,true>", ""))
    assert not kept and "not reported" in err


def test_counters_are_tied_to_the_kernel_code_object(tmp_path, monkeypatch):
    """PMC passes describe a CODE OBJECT.  The digest stores the kernel's ISA hash
    (pyro_amd/kernel_manifest.py, from the device assembly of the build); check_counters compares it with the hash of the same
    kernel in THIS build's manifest: a kernel recompiled to other instructions under the same template name invalidates the
    counters, a comment edit under csrc/ does not, counters without a hash are dropped."""
    import json
    bench = _bench()
    from pyro_amd import _build
    from pyro_amd import kernel_manifest as KM
    lib = tmp_path / "libpyrovi.so"
    lib.write_text("a library")
    man = {"k_sweep_lean4<2,unsignedchar,true,true>": {"exact": "aaaa000011112222", "loose": "x"}, "_library": KM.file_id(str(lib))}
    mpath = tmp_path / "kernel_manifest.json"
    mpath.write_text(json.dumps(man))
    monkeypatch.setattr(_build, "MANIFEST", str(mpath))
    monkeypatch.setattr(_build, "OUT", str(lib))
    assert bench.kernel_isa_hash("void k_sweep_lean4<2, unsigned char, true, true>(DevP, Lean4P)") == "aaaa000011112222"
    desc = "path=lean kernel=k_sweep_lean4<2,unsignedchar,true,true> tile=19x51 win=1 note="
    c = {"kernel": "void k_sweep_lean4<2, unsigned char, true, true>(DevP, Lean4P)", "kernel_path": "path=lean tile=19x51 win=1",
         "source": "profiles/x.json", "isa_hash": "aaaa000011112222", "hbm_bytes_per_launch": 1.0}
    kept, err = bench.check_counters(c, desc)
    assert kept and err is None
    kept, err = bench.check_counters(dict(c, isa_hash=None), desc)           # counters from before the hash existed
    assert not kept and "not recorded" in err
    kept, err = bench.check_counters(dict(c, isa_hash="bbbb000011112222"), desc)   # the kernel was recompiled since
    assert not kept and "code object:" in err and "bbbb000011112222" in err and "aaaa000011112222" in err
    # a manifest written for another library file does not describe this one: nothing is trusted
    lib.write_text("another library")
    os.utime(lib, (os.path.getmtime(lib) + 5,) * 2)
    assert bench.kernel_isa_hash("k_sweep_lean4<2,unsignedchar,true,true>") is None
    kept, err = bench.check_counters(c, desc)
    assert not kept and "no manifest" in err


def test_committed_counters_name_the_code_objects_they_were_taken_on():
    """Every entry of profiles/counters.json carries an ISA hash (or null = not attributable); the entries whose kernel this
    build still compiles to the same instructions are exactly the ones bench.py will keep."""
    import json
    bench = _bench()
    ctr = json.load(open(os.path.join(ROOT, "profiles", "counters.json")))
    man = bench.kernel_manifest()
    if not man:
        pytest.skip("no kernel manifest next to the library (build with pyro_amd/_build.py)")
    live = {}
    for w, c in ctr.items():
        if w.startswith("_"):
            continue
        assert "isa_hash" in c and "csrc_hash" not in c, w
        k = bench.norm_kernel(c["kernel"])
        live[w] = c["isa_hash"] is not None and man.get(k, {}).get("exact") == c["isa_hash"]
    # round 4's passes: the float64 kernels and the n = 3 kernel are unchanged code objects; the float32 window sweeps were
    # rewritten in round 5 (their counters are dropped until a new pass)
    assert live["c5"] and live["c5d"] and live["c1"] and live["h3"], live

