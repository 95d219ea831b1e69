"""
CPU tests of the kernel SOURCES under emulation (tests/emu/): libpyrovi's .hip sources compiled as host C++ against a source-level
emulation of the HIP programming model (work-items as fibers, wave collectives, LDS, buffer loads, atomics, a synchronous runtime),
driven through the same C ABI and the same Python classes as the product, compared with the oracle and the reference goldens on
small grids.  Rounds 5 and 6 had no MI355X for most of their length; this is what can be said about a kernel without one: its source
computes the reference's recursion -- index arithmetic, tile schedules, LDS window addressing, reductions, stop test, refusals --
including the kernels that have never run on hardware (error feedback outside 4-D grids, the swapped cart-pole order, the
corruption detector, the multi-sweep launch of the 2-D float32 sweep).  It says NOTHING about the gfx950 code objects (register
allocation, hazards, occupancy, timing): that is tests -m gpu, profiles/verified_kernels.json and DESIGN.md 4.8.

The emulated library is TEST INFRASTRUCTURE: built here into tests/emu/_build/ (git-ignored), loaded only by the subprocesses below
through PYROVI_LIB; nothing in pyro_amd/ knows it exists, and the product still fails loudly without libpyrovi.so and a HIP device.
"""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

EMU = os.path.join(ROOT, "tests", "emu")


@pytest.fixture(scope="module")
def emu_lib():
    sys.path.insert(0, EMU)
    try:
        import build_emu
    finally:
        sys.path.pop(0)
    if not os.path.exists(build_emu.CLANG):
        pytest.skip("no host clang++ at %s" % build_emu.CLANG)
    return build_emu.build()


def _all_cpus():
    """(preexec) The C oracle's OpenMP settings (OMP_PROC_BIND / OMP_PLACES, set by oracle/c_oracle.py earlier in the same pytest
    session) leave THIS process bound to one core, and children inherit that: give the emulator's host threads the machine back."""
    os.sched_setaffinity(0, range(os.cpu_count() or 1))


def _env(lib, **extra):
    env = {k: v for k, v in os.environ.items() if not k.startswith("OMP_")}
    env.update(PYROVI_LIB=lib, PYTHONPATH=ROOT, **extra)
    return env


def run_check(lib, *names, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(EMU, "checks.py")] + list(names), env=_env(lib), cwd=ROOT, capture_output=True, text=True,
                       timeout=timeout, preexec_fn=_all_cpus)
    sys.stdout.write(r.stdout[-4000:])
    assert r.returncode == 0, (r.stdout[-3000:] + "\n" + r.stderr[-3000:])
    for n in names:
        assert ("== %s passed" % n) in r.stdout
    return r.stdout


def test_emulated_float64_kernels_are_bit_identical_to_the_oracle(emu_lib):
    run_check(emu_lib, "f64_bit_identical")


def test_emulated_float32_families_match_the_oracle_and_each_other(emu_lib):
    run_check(emu_lib, "f32_paths")


def test_emulated_feedback_sweep_and_its_corruption_detector(emu_lib):
    out = run_check(emu_lib, "feedback_4d_and_detector")
    assert "detector fired" in out


def test_emulated_swapped_cartpole_order(emu_lib):
    run_check(emu_lib, "swapped_order")


def test_emulated_feedback_outside_4d_grids(emu_lib):
    run_check(emu_lib, "feedback_2d_explicit_node")


def test_emulated_multi_sweep_launches(emu_lib):
    """Cooperative launches: every workgroup a host thread of its own, the grid barriers spin on real atomics."""
    out = run_check(emu_lib, "multi_sweep_launches")
    assert "kernel=k_sweep_leanm<1,unsignedchar,true>" in out and "kernel=k_sweep_leanm<4,unsignedchar" in out and "kernel=k_sweep64m<" in out


def test_emulated_in_library_rccl_path_with_two_and_three_rank_processes(emu_lib):
    """pvi_shard_* over a stand-in for librccl (shared memory between rank processes): the exchange code with a NEIGHBOUR, which the
    one-GPU test box has never allowed (RCCL refuses two ranks on one device)."""
    out = run_check(emu_lib, "rccl_shards")
    assert "rccl_ranks=3" in out and "send/recv+overlap pieces=2" in out


def test_emulated_slabs_and_the_halo_rule(emu_lib):
    out = run_check(emu_lib, "slabs_and_halo")
    assert "refused (PVI_EHALO)" in out


def test_emulated_tiers_on_the_reference_goldens(emu_lib):
    run_check(emu_lib, "table_tier_spline_rollout")


def test_driver_entry_points_end_to_end_on_the_emulated_library(emu_lib):
    """What the driver runs on the GPU box at round end -- __graft_entry__.smoke() and bench.py's line -- end to end against the
    emulated library (a tiny custom workload): the code paths exist, the JSON line carries every contract key, and a line that
    does not come from pyro_amd/libpyrovi.so says so (`invalid`)."""
    import json
    env = _env(emu_lib)
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600,
                       preexec_fn=_all_cpus)
    assert r.returncode == 0 and "smoke ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([sys.executable, "bench.py", "--workload", "pendulum:41,41:7:float32", "--steps", "3", "--warmup", "1", "--cpu-budget", "1"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=600, preexec_fn=_all_cpus)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().split("\n")[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"]) and line["roofline"]["bound"] == "hbm"
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"]) and line["cpu_baseline"]["kind"] == "port"
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["dtype"] == "f32" and line["vs_baseline"] is None
    assert line["unverified_kernels"] == 0 and line["kernel_verified"] is True          # (this build's manifest against the verified list)
    assert "not the product library" in line["invalid"]
    assert line["step_rel_err_vs_cpu"] <= 1e-5


def test_multi_gpu_bench_harness_end_to_end_on_the_emulated_library(emu_lib):
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` -- the command the driver uses for the scaling runs,
    which has never run with N > 1 on hardware -- against the emulated library and the librccl stand-in: two rank processes,
    the in-library RCCL path (not the torch fall-back), the self-test of a sharded solve against one rank's, ONE JSON line."""
    import json
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = _env(emu_lib, PVI_RCCL_LIB=os.path.join(os.path.dirname(emu_lib), "librccl_emu.so"), PVI_EMU_DEVICES="8", PVI_EMU_THREADS="3")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), "bench.py", "--gpus", "2", "--workload", "cartpole:11,12,9,10:5:float32", "--steps", "3",
                        "--warmup", "1", "--no-cpu", "--selftest-grid", "cartpole:9,7,8,9:5:float32"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900, preexec_fn=_all_cpus)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["rccl_ranks"] == 2 and line["selftest"]["ok"] is True
    assert "RCCL inside libpyrovi" in line["config"]["parallelism"] and "p2p send/recv" in line["config"]["parallelism"]
    assert line["value"] > 0 and line["scaling"] == "strong" and "not the product library" in line["invalid"]
    assert len(line["per_rank"]["sweep_ms"]) == 2


def test_reference_scripts_solve_identically_with_the_reference_and_with_this_package(emu_lib):
    """Where the reference tree is present (the build container; it does not travel, and nothing under -m gpu, smoke() or bench.py
    reads it): two of the reference's own value-iteration scripts, UNMODIFIED, run twice -- with the reference itself and with
    `pyro.*` resolved to `pyro_amd.*` on the emulated library -- leave the same solve behind: the same number of sweeps, J to 1e-9
    of max J, the same action on 99.9 % of the nodes (tools/compare_reference_demos.py; all 22 scripts:
    profiles/r06_reference_demos_compared.log, where every one is identical to 3.7e-15 and on every node)."""
    ref = os.environ.get("PYRO_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "examples", "demos_by_tool", "dynamicprogramming")):
        pytest.skip("the reference tree is not here")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "compare_reference_demos.py"), "pendulum_optimal_swingup_low_def_fast_computation.py",
                        "demos_by_system/car_propulsion/longitudinal_car_braking_value_iteration.py"],
                       env=_env(emu_lib, DEMO_TIMEOUT="600", PYRO_REFERENCE=ref), cwd=ROOT, capture_output=True, text=True, timeout=1500, preexec_fn=_all_cpus)
    sys.stdout.write(r.stdout[-2000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count(" ok  ") == 2 and "differs from the reference's (or could not be compared): 0 / 2" in r.stdout


def test_the_product_never_loads_the_emulated_library():
    """grep: nothing under pyro_amd/, bench.py or __graft_entry__.py names tests/emu or libpyrovi_emu."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "pyro_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".inc", ".h")):
                t = open(os.path.join(base, f), errors="replace").read()
                if "libpyrovi_emu" in t or "tests/emu" in t or "PVI_EMU" in t:
                    bad.append(os.path.join(base, f))
    for f in ("bench.py", "__graft_entry__.py"):
        t = open(os.path.join(ROOT, f)).read()
        if "libpyrovi_emu" in t or "tests/emu" in t:
            bad.append(f)
    assert not bad, bad
