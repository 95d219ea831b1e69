#!/usr/bin/env python3
"""
TEST INFRASTRUCTURE: the checks tests/test_emu_cpu.py runs, each in a subprocess with PYROVI_LIB pointing at the EMULATED build of
libpyrovi (tests/emu/build_emu.py): the product's kernel and host sources compiled for the CPU, driven through the same C ABI and
the same Python classes, compared with the oracle on grids small enough for an emulator.

    PYROVI_LIB=tests/emu/_build/libpyrovi_emu.so python tests/emu/checks.py <check> [...]

A check prints what it measured and raises on failure.  What a green check means and does not mean: tests/emu/README.md.
"""
import contextlib
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import c_oracle as CO  # noqa: E402
from oracle import vi_oracle as O  # noqa: E402
from pyro_amd import _native, configs  # noqa: E402
from pyro_amd.planning import dynamicprogramming as DP  # noqa: E402

assert "emu" in os.path.basename(_native.LIB_PATH), "these checks are for the emulated library (PYROVI_LIB), not the product: %s" % _native.LIB_PATH


# PVI_EMU_FULL=1 (tests/emu/run_asan.sh, the logs under profiles/): every case.  Default (tests/test_emu_cpu.py, part of the CPU
# suite the driver runs every round): a subset of the shapes, the same assertions.
FULL = os.environ.get("PVI_EMU_FULL") == "1"


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def build(name):
    with quiet():
        return configs.build(name)


def make(cfg, dtype, fb=False, order="reference", tune=True, **ov):
    """A DynamicProgramming engine under pvi_override keys.  tune=False adds TUNE=0: pvi_create takes the first tile shape that fits
    instead of timing candidates.  (Measured: the timed set-up pays for itself even here -- it picks the shape the EMULATOR sweeps
    fastest, 42 s instead of 82 s for check_f32_paths.)"""
    if not tune and "TUNE" not in ov:
        ov = dict(ov, TUNE="0")
    with quiet(), _native.overrides(**ov):
        dp = DP.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"], dtype=dtype, f32_feedback=fb, internal_order=order)
    dp.save_time_history = False
    dp.verbose = False
    return dp


def oracle_sweeps(cfg, n, alpha=1.0):
    """(J, pi) after n sweeps of the C twin of the oracle (float64), and J of the sweep before."""
    import bench
    c = CO.CProblem(bench.oracle_problem(cfg))
    p = bench.oracle_problem(cfg)
    J = O.terminal_cost(p)
    Jprev = J
    pi = None
    for _ in range(n):
        Jprev = J
        J, pi = c.sweep(J, alpha)[:2]
    return J, pi, Jprev, c


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def tokens(desc):
    return dict(t.split("=", 1) for t in desc.split() if "=" in t)


# ------------------------------------------------------------------------------------------------------------------------------
def check_f64_bit_identical():
    """The float64 kernels mirror the oracle operation for operation: J and pi bit for bit -- second form (k_sweep64), the
    operation-for-operation kernel (k_sweep), 2-D and 4-D, dense and sparse walks, line and patch mappings."""
    for name, n in (("pendulum:41,41:7:float64", 12), ("cartpole:9,8,11,10:5:float64", 5), ("twolink:7,7,9,9:3,3:float64", 4)):
        cfg = build(name)
        J, pi, _, _ = oracle_sweeps(cfg, n)
        for ov in ({}, {"NO_SWEEP64": "1"}, {"SPARSE": "1"}, {"SPARSE": "0"}, {"PATCH": "1"}, {"PATCH": "0"}):
            if len(cfg["grid_sys"].x_grid_dim) == 2 and ("SPARSE" in ov or "PATCH" in ov):
                continue
            dp = make(cfg, "float64", **ov)
            stats, done = dp._p.sweep(n, 1.0, -1.0)
            assert done == n
            Jg, pig = dp._p.get_J(), dp._p.get_pi()
            print(name, ov, dp._p.describe().split(" note=")[0][:150])
            assert np.array_equal(Jg, J), (name, ov, rel(Jg, J))
            assert np.array_equal(pig, pi), (name, ov, (pig != pi).mean())
            dp._p.close()


def check_f32_paths():
    """Every float32 family against the float64 oracle on one problem per shape: 2-D window sweep (uniform walk, lane split, two
    nodes per thread), the 4-D window sweep with its table / schedule / validity variants (same bits among them), the fall-backs."""
    cfg = build("pendulum:41,41:9:float32")
    n = 15
    J, pi, Jprev, c = oracle_sweeps(cfg, n)
    for ov in ({}, {"LSPLIT": "0"}, {"LSPLIT": "2"}, {"LSPLIT": "0", "NPT": "2"}, {"NO_LEAN": "1"}, {"NO_LEAN": "1", "NO_FAST": "1"}):
        dp = make(cfg, "float32", **ov)
        dp._p.sweep(n, 1.0, -1.0)
        e = rel(dp._p.get_J(), J)
        print("pendulum", ov, "%.2e" % e, dp._p.describe().split(" note=")[0][:120])
        assert e <= 2e-6, (ov, e)
        assert (dp._p.get_pi() != pi).mean() < 0.01
        dp._p.close()
    for name in ("cartpole:13,12,15,14:5:float32", "twolink:7,8,9,10:3,3:float32")[:2 if FULL else 1]:
        cfg = build(name)
        n = 6
        J, pi, Jprev, c = oracle_sweeps(cfg, n)
        outs = {}
        full = name.startswith("cartpole")          # (every variant on the cart-pole; the launch-order ones once is enough)
        for tag, ov in (("default", {}), ("win1", {"WIN": "1"}), ("tab0", {"WIN": "1", "TABLES": "0"}), ("tab1", {"WIN": "1", "TABLES": "1"}),
                        ("tab2", {"WIN": "1", "TABLES": "2"}), ("clamp", {"WIN": "1", "VMASK": "0"}), ("noxcd", {"WIN": "1", "NO_XCD": "1"}),
                        ("bands2", {"WIN": "1", "BANDS": "2"}), ("shape", {"WIN": "1", "TV0": "3", "TV1": "7"}),
                        ("win0", {"WIN": "0"}), ("fast", {"NO_LEAN": "1"}), ("exact32", {"NO_FAST": "1"})):
            if not full and tag in ("tab1", "noxcd", "bands2", "shape", "win0"):
                continue
            dp = make(cfg, "float32", **ov)
            dp._p.sweep(n, 1.0, -1.0)
            outs[tag] = (dp._p.get_J(), dp._p.get_pi(), dp._p.describe())
            e = rel(outs[tag][0], J)
            print(name, tag, "%.2e" % e, outs[tag][2].split(" cands=")[0][:170])
            assert e <= 2e-6, (name, tag, e)
            dp._p.close()
        for tag in ("win1", "tab0", "tab1", "tab2", "clamp", "noxcd", "bands2", "shape"):   # the 4-D window sweep: the same bits whatever the variant
            if tag not in outs:
                continue
            assert "win=1" in outs[tag][2] and "kernel=k_sweep_lean4<" in outs[tag][2], (tag, outs[tag][2])
            assert np.array_equal(outs[tag][0], outs["win1"][0]) and np.array_equal(outs[tag][1], outs["win1"][1]), tag
        assert tokens(outs["tab2"][2])["tables"] == tokens(outs["win1"][2])["tables"]             # declared == found
        assert tokens(outs["win1"][2])["vmask"] == "1" and tokens(outs["clamp"][2])["vmask"] == "0"
        # the policy: float64 Q-regret of the chosen actions on the float64 J of the sweep before
        nodes = np.arange(0, J.size, 3, dtype=np.int64)
        q, qmin = c.q_at(Jprev, nodes, outs["default"][1][nodes])
        ok = np.isfinite(q) & np.isfinite(qmin)
        assert (q[ok] - qmin[ok]).max() <= 1e-5 * np.abs(Jprev).max()


def check_feedback_4d_and_detector():
    """Error-feedback storage on a 4-D grid (k_sweep_lean4fb): closer to float64 than plain storage at every checkpoint; a restart
    clears the residuals; a self check between batches is a dry run.  With FBCHECK=1 (k_sweep_lean4fbc, never run on hardware) the
    detector stays silent and every bit is the same -- and a handle whose residual table is corrupted ... is not what it detects;
    what it detects is exercised by poisoning the float32 loop's coefficient table (PVI_ECORRUPT)."""
    cfg = build("cartpole:13,12,15,14:5:float32")
    d64, d32, dfb = make(cfg, "float64"), make(cfg, "float32"), make(cfg, "float32", True)
    dck = make(cfg, "float32", True, FBCHECK="1")
    worst_fb = worst_plain = 0.0
    for k in range(4):
        for dp in (d64, d32, dfb, dck):
            dp._p.sweep(15, 1.0, -1.0)
        J64 = d64._p.get_J()
        e_fb, e_pl = rel(dfb._p.get_J(), J64), rel(d32._p.get_J(), J64)
        worst_fb, worst_plain = max(worst_fb, e_fb), max(worst_plain, e_pl)
        print("after %d sweeps: feedback %.3e plain %.3e" % (15 * (k + 1), e_fb, e_pl))
        assert e_fb <= 6e-7
        assert np.array_equal(dck._p.get_J(), dfb._p.get_J()) and np.array_equal(dck._p.get_pi(), dfb._p.get_pi())
        rel_sc, mism = dfb._p.self_check(1.0)
        assert rel_sc <= 1e-5
    assert worst_fb < worst_plain
    dfb._p.sweep(1, 1.0, -1.0)       # (describe names the kernel of the LAST launch: the self check's was the plain-gather kernel)
    dck._p.sweep(1, 1.0, -1.0)
    assert "kernel=k_sweep_lean4fb<" in dfb._p.describe() and "kernel=k_sweep_lean4fbc<" in dck._p.describe(), (dfb._p.describe(), dck._p.describe())
    # The firing side.  Device memory is host memory here: find the float32 loop's (position, action) coefficient table of the
    # checked handle by its content -- records of 24 floats whose slots 12..15 are gu dt of four consecutive actions -- and add 3.0
    # to the loop's cost of ONE action group at every position.  The loop's value of those actions is now off by 3, the epilogue
    # (float64 cost from P.gu) is not: wherever one of them still wins, the two disagree -> PVI_ECORRUPT.
    import ctypes
    L = ctypes.CDLL(_native.LIB_PATH)
    L.emu_alloc_count.restype = ctypes.c_long
    g = cfg["grid_sys"]
    R = float(np.asarray(cfg["cf"].R).ravel()[0])
    gudt = np.array([(u * R * u) * g.dt for u in np.asarray(g.u_level[0], dtype=float)], dtype=np.float32)
    before = {}
    for i in range(L.emu_alloc_count()):
        ptr, n = ctypes.c_void_p(), ctypes.c_size_t()
        assert L.emu_alloc_get(ctypes.c_long(i), ctypes.byref(ptr), ctypes.byref(n)) == 0
        before[ptr.value] = n.value
    dck2 = make(cfg, "float32", True, FBCHECK="1")
    dck2._p.sweep(3, 1.0, -1.0)
    hits = 0
    for i in range(L.emu_alloc_count()):
        ptr, n = ctypes.c_void_p(), ctypes.c_size_t()
        L.emu_alloc_get(ctypes.c_long(i), ctypes.byref(ptr), ctypes.byref(n))
        if ptr.value in before or n.value < 96 or n.value % 96:
            continue
        a = np.ctypeslib.as_array((ctypes.c_float * (n.value // 4)).from_address(ptr.value)).reshape(-1, 24)
        if np.array_equal(a[0, 12:16], gudt[:4]) and np.array_equal(a[-2 if len(gudt) > 4 else -1, 12:16], gudt[:4]):
            a[0::2 if len(gudt) > 4 else 1, 12:16] -= 3.0       # (group 0 of every position node: its actions now look 3 cheaper to the loop)
            hits += 1
    assert hits >= 1, "coefficient table not found"
    try:
        dck2._p.sweep(2, 1.0, -1.0)
    except _native.NativeError as e:
        assert e.code == _native.PVI_ECORRUPT and "corruption detector" in str(e), e
        print("detector fired:", str(e)[:150])
    else:
        raise AssertionError("the corruption detector did not fire on a poisoned coefficient table")
    fresh = make(cfg, "float32", True)
    fresh._p.sweep(20, 1.0, -1.0)
    dfb.evaluate_terminal_cost()
    dfb._p.sweep(20, 1.0, -1.0)
    assert np.array_equal(dfb._p.get_J(), fresh._p.get_J())


def check_swapped_order():
    """internal_order="swapped" (PVI_DYN_CARTPOLE_SW, never run on hardware): the same problem -- J within float32 tolerance of the
    float64 solve in the reference's order, with feedback within 6e-7; the policy inside the Q-regret rule; self check; set J."""
    cfg = build("cartpole:11,10,13,12:5:float32")
    J, pi, Jprev, c = oracle_sweeps(cfg, 12)
    for fb in (False, True):
        dsw = make(cfg, "float32", fb, "swapped")
        d32 = make(cfg, "float32", fb)
        assert np.allclose(dsw.J, d32.J, rtol=1e-6, atol=0)
        for dp in (dsw, d32):
            dp._p.sweep(12, 1.0, -1.0)
        desc = dsw._p.describe()
        assert "order=swapped" in desc and "tables=10" in desc and ("kernel=k_sweep_lean4fb<12" if fb else "kernel=k_sweep_lean4<12") in desc, desc
        e = rel(dsw._p.get_J(), J)
        print("swapped fb=%s: %.3e (reference order %.3e)" % (fb, e, rel(d32._p.get_J(), J)))
        assert e <= (6e-7 if fb else 2e-6)
        nodes = np.arange(0, J.size, 3, dtype=np.int64)
        q, qmin = c.q_at(Jprev, nodes, dsw._p.get_pi()[nodes])
        ok = np.isfinite(q) & np.isfinite(qmin)
        assert np.array_equal(np.isfinite(q), np.isfinite(qmin)) and (q[ok] - qmin[ok]).max() <= 1e-5 * np.abs(Jprev).max()
        rel_sc, mism = dsw._p.self_check(1.0)
        assert rel_sc <= 1e-5, (rel_sc, mism)
        for dp in (d32, dsw):
            dp.J = J
            dp._flush()
            dp._p.sweep(1, 1.0, -1.0)
        assert np.abs(dsw._p.get_J() - d32._p.get_J()).max() <= 2e-6 * np.abs(J).max()
    try:
        make(cfg, "float64", False, "swapped")
    except NotImplementedError:
        pass
    else:
        raise AssertionError("float64 with the swapped order must be refused")


def check_feedback_2d_explicit_node():
    """Error-feedback storage outside 4-D grids (never run on hardware; UNPROVEN=1): the 2-D window sweep in its uniform and
    lane-split walks (k_sweep_leanfb), an explicit system (k_sweep3_fast with the feedback arguments), the node-table tier
    (MountainCar) -- every checkpoint within 1e-6 of float64 and no worse than plain storage; restart; the gate without the override."""
    from pyro_amd.analysis import costfunction
    from pyro_amd.dynamic import mountaincar
    from pyro_amd.planning import discretizer
    with quiet():
        s = mountaincar.MountainCar()
        g = discretizer.GridDynamicSystem(s, [41, 41], [5])
        cfm = costfunction.QuadraticCostFunction.from_sys(s)
        cfm.INF = 100
    cases = [("pendulum:41,41:9:float32", build("pendulum:41,41:9:float32"), 60, "k_sweep_leanfb<1"),
             ("pendulum:31,31:101:float32", build("pendulum:31,31:101:float32"), 40, "k_sweep_leanfb<1"),
             ("mountaincar 41x41x5", {"grid_sys": g, "cf": cfm}, 40, "k_sweep_leanfb<4"),
             ("helicopter 21x21x31x7", dict(zip(("sys", "grid_sys", "cf"), configs._helicopter((21, 21, 31), (7,), "float32")[:3])), 30, "k_sweep3_fast<7,unsignedchar,float*,double>")]
    for name, cfg, n, kern in cases:
        try:
            make(cfg, "float32", True)
        except NotImplementedError as e:
            assert "UNPROVEN" in str(e)
        else:
            raise AssertionError("%s: feedback outside 4-D grids must be gated" % name)
        d64, d32, dfb = make(cfg, "float64"), make(cfg, "float32"), make(cfg, "float32", True, UNPROVEN="1")
        worst_fb = worst_plain = 0.0
        for k in range(2):
            for dp in (d64, d32, dfb):
                dp._p.sweep(n // 2, 1.0, -1.0)
            J64 = d64._p.get_J()
            e_fb, e_pl = rel(dfb._p.get_J(), J64), rel(d32._p.get_J(), J64)
            worst_fb, worst_plain = max(worst_fb, e_fb), max(worst_plain, e_pl)
            print("%s after %d sweeps: feedback %.3e plain %.3e" % (name, (n // 2) * (k + 1), e_fb, e_pl))
            assert e_fb <= 1e-6, (name, e_fb)
        assert worst_fb <= worst_plain * 1.05 + 1e-9, (name, worst_fb, worst_plain)
        assert ("kernel=" + kern) in dfb._p.describe(), dfb._p.describe()
        fresh = make(cfg, "float32", True, UNPROVEN="1")
        fresh._p.sweep(10, 1.0, -1.0)
        dfb.evaluate_terminal_cost()
        dfb._p.sweep(10, 1.0, -1.0)
        assert np.array_equal(dfb._p.get_J(), fresh._p.get_J()), name


def check_slabs_and_halo():
    """Two slabs with a host-side halo exchange reproduce the whole-grid sweep bit for bit (float64 2-D, float32 4-D); a halo one
    row short is refused or reported, never clamped silently -- also when the reach is a whole number of cells up to rounding."""
    import test_gpu_parity as T
    from pyro_amd import parallel
    for name, dtype in (("pendulum_demo_51x51x9", "float64"), ("cartpole_11p4x5", "float32")):
        g = T.load(name)
        p = T.oracle_problem(g, *T.CASES[name])
        whole = T.native_problem(p, dtype=dtype)
        whole.terminal_cost()
        N0 = p.dims[0]
        mid, halo = N0 // 2, 6
        slabs = [T.native_problem(p, dtype=dtype, rows=(0, mid), halo=(0, halo)), T.native_problem(p, dtype=dtype, rows=(mid, N0), halo=(halo, 0))]
        for s in slabs:
            s.terminal_cost()
        for _ in range(3):
            whole.sweep(1, 1.0, -1.0)
            for s in slabs:
                s.sweep_async(1.0)
                s.sweep_stats()
            parts = [s.get_J() for s in slabs]
            assert np.array_equal(np.concatenate(parts), whole.get_J()), name
            full = np.concatenate(parts).reshape(N0, -1)
            for s in slabs:
                r0, r1 = s.store_rows
                s.set_J(full[r0:r1].ravel(), r0, r1 - r0)
        print(name, dtype, "slabs == whole grid", slabs[0].describe()[:80])
    # the reach sits on a whole number of cells (+- 1e-6): the rule's halo agrees, one row less is refused / reported
    for system, dims, k, lb, ub in (("pendulum", (31, 29), 2, [-3.2, -6.0], [3.2, 6.0]),
                                    ("cartpole", (13, 9, 11, 10), 2, [-4.0, -3.2, -5.0, -6.0], [4.0, 3.2, 5.0, 6.0])):
      n = len(dims)
      lv = O.make_levels(np.array(lb), np.array(ub), np.array(dims))
      ul = O.make_levels(np.array([-10.0]), np.array([10.0]), np.array([7 if n == 2 else 5]))
      step0 = (ub[0] - lb[0]) / (dims[0] - 1)
      dyn, consts = (O.DYN_PENDULUM, O.pendulum_consts()) if n == 2 else (O.DYN_CARTPOLE, O.cartpole_consts())
      for eps in ((-1.5e-6, -4e-7, 0.0, 4e-7) if FULL else (-4e-7, 4e-7)):
        dt = (k + eps) * step0 / ub[n // 2]
        p = O.Problem(lv, ul, dt, dyn, consts, np.eye(n), np.eye(1), np.zeros((n, n)), np.zeros(n), np.zeros(1), 1e4, 0.2)
        need = parallel._rows_for_reach(k + eps)
        mid = dims[0] // 2
        plane = int(np.prod(dims[1:]))
        whole = T.native_problem(p, dtype="float32")
        whole.terminal_cost()
        whole.sweep(1, 1.0, -1.0)
        ok = T.native_problem(p, dtype="float32", rows=(0, mid), halo=(0, need))
        ok.terminal_cost()
        ok.sweep_async(1.0)
        ok.sweep_stats()
        assert np.array_equal(ok.get_J(), whole.get_J()[:mid * plane]), eps
        outcome = "agrees"
        try:
            short = T.native_problem(p, dtype="float32", rows=(0, mid), halo=(0, need - 1))
            short.terminal_cost()
            short.sweep_async(1.0)
            short.sweep_stats()
            # (not refused and not reported: then every gather stayed inside the stored rows -- x_next exactly on a level takes the
            #  cell below it, whose upper corner is the last stored row -- and the result must be the whole grid's, bit for bit)
            assert np.array_equal(short.get_J(), whole.get_J()[:mid * plane]), (eps, short.describe())
        except _native.NativeError as e:
            outcome = "refused (%s)" % ("PVI_EHALO" if e.code == _native.PVI_EHALO else str(e)[:60])
        print("%s reach %d%+.1e cells: halo %d ok (%s); halo %d %s" % (system, k, eps, need, ok.describe().split()[0], need - 1, outcome))


def check_table_tier_spline_rollout():
    """The other tiers of the class surface on their reference goldens, through the emulated library: table tier (LUT and base
    semantics, obstacles), policy evaluation, nearest interpolation, the bicubic-spline mode, batched rollouts."""
    import test_gpu_parity as T
    for fn, args in ((T.test_table_tier_on_reference_tables, ()), (T.test_config1_solve_f64_matches_reference, ()),
                     (T.test_table_tier_lut_and_base_semantics, ("obstacles_21x21x3x3", 5)),
                     (T.test_table_tier_lut_and_base_semantics, ("reachability_41x41x3", 20)),
                     (T.test_policy_evaluator_classes, ()), (T.test_batched_rollouts_match_reference, ()),
                     (T.test_spline_value_iteration_matches_reference_golden, ("fused",)),
                     (T.test_mountaincar_runs_fused_through_node_tables_and_matches_reference, ()),
                     (T.test_minimum_time_cost_runs_fused_and_matches_reference, ())):
        with quiet():
            fn(*args)
        print(fn.__name__, args, "ok")
    import test_gpu_zz_unproven as Z
    with quiet():
        Z.test_slinear_interpolation_through_the_class_surface_matches_reference_golden()
    print("interpol_method = 'slinear' against the reference's own slinear solves: ok")
    with quiet():
        Z.test_cubic_interpolation_through_the_class_surface_matches_reference_golden()
    print("interpol_method = 'cubic' / 'cubic_legacy' (2-D) against the reference's own cubic_legacy solves: ok")


def check_multi_sweep_launches():
    """Batches as ONE cooperative launch with grid barriers between the sweeps: k_sweep64m (float64; verified on hardware) and
    k_sweep_leanm (the 2-D float32 window sweep; MULTI32=1, never run on hardware) against one launch per sweep -- J, pi, every
    sweep's statistics and the stop sweep are the same bits; odd batch sizes, a tolerance stop inside a batch, a restart."""
    import test_gpu_parity as T
    g = T.load("config1_pendulum_101x101x11")
    p = T.oracle_problem(g, O.DYN_PENDULUM, O.pendulum_consts())
    m = T.native_problem(p)
    m.terminal_cost()
    sm, nm = m.sweep(5000, 1.0, 0.1)
    with _native.overrides(MULTI="0"):          # (the form of a handle's batches is decided at its first sweep)
        s1 = T.native_problem(p)
        s1.terminal_cost()
        ss, ns = s1.sweep(5000, 1.0, 0.1)
    print("k_sweep64m:", m.describe()[:150])
    assert "multi=1" in m.describe() and "kernel=k_sweep64m<" in m.describe() and "multi=0" in s1.describe()
    assert nm == ns == int(g["sweeps"]) == 618
    assert np.array_equal(np.array(sm), np.array(ss)) and np.array_equal(m.get_J(), s1.get_J()) and np.array_equal(m.get_pi(), s1.get_pi())
    assert rel(m.get_J(), g["J"].ravel()) < 1e-12
    from pyro_amd.analysis import costfunction
    from pyro_amd.dynamic import mountaincar
    from pyro_amd.planning import discretizer
    with quiet():
        mc = mountaincar.MountainCar()
        gm = discretizer.GridDynamicSystem(mc, [51, 41], [5])
        cfm = costfunction.QuadraticCostFunction.from_sys(mc)
        cfm.INF = 100
    for name, extra in (("pendulum:61,61:9:float32", {}), ("pendulum:61,61:9:float32", {"LSPLIT": "0"}),
                        ("pendulum:45,75:101:float32", {}), ("mountaincar", {}), ("mountaincar", {"LSPLIT": "0"}))[slice(None) if FULL else slice(1, 4)]:
        cfg = {"grid_sys": gm, "cf": cfm} if name == "mountaincar" else build(name)
        with _native.overrides(MULTI32="1", **extra):
            a = make(cfg, "float32")._p
            sa, na = a.sweep(1, 1.0, -1.0)
        with _native.overrides(MULTI="0", **extra):
            b = make(cfg, "float32")._p
            sb, nb = b.sweep(1, 1.0, -1.0)
        print("k_sweep_leanm:", name, a.describe()[:170])
        assert "multi=1" in a.describe() and "kernel=k_sweep_leanm<" in a.describe(), a.describe()
        assert "multi=0" in b.describe() and "kernel=k_sweep_lean<" in b.describe(), b.describe()
        assert np.array_equal(np.array(sa), np.array(sb))
        for n in (1, 2, 3, 7, 20, 5):
            sa, na = a.sweep(n, 1.0, -1.0)
            sb, nb = b.sweep(n, 1.0, -1.0)
            assert na == nb == n and np.array_equal(np.array(sa), np.array(sb)), (name, n)
            assert np.array_equal(a.get_J(), b.get_J()) and np.array_equal(a.get_pi(), b.get_pi()), (name, n)
        probe, _ = b.sweep(20, 1.0, -1.0)
        a.sweep(20, 1.0, -1.0)
        tol = float(np.array(probe)[-1, 3]) * 0.7
        sa, na = a.sweep(300, 1.0, tol)
        sb, nb = b.sweep(300, 1.0, tol)
        assert na == nb and 0 < na < 300, (name, na, nb, tol)
        assert np.array_equal(np.array(sa), np.array(sb)) and np.array_equal(a.get_J(), b.get_J()) and np.array_equal(a.get_pi(), b.get_pi())
        assert np.array_equal(a.get_J(prev=True), b.get_J(prev=True))
        for h in (a, b):
            h.terminal_cost()
        sa, _ = a.sweep(15, 1.0, -1.0)
        sb, _ = b.sweep(15, 1.0, -1.0)
        assert np.array_equal(np.array(sa), np.array(sb)) and np.array_equal(a.get_J(), b.get_J())
        print("   stop sweep %d of 300 at tol %.4g: identical" % (na, tol))


def check_rccl_shards():
    """The in-library multi-GPU path (pvi_shard_*: axis-0 slabs, halo exchange by ncclSend / ncclRecv in one group, all-gather by
    broadcasts, three-double all-reduce, boundary-first overlap schedule) with 1, 2 and 3 rank PROCESSES over the stand-in for
    librccl (tests/emu/rccl_emu.cpp: shared memory) -- on hardware two ranks have never shared the test box's one GPU (RCCL
    refuses), so this is the first time the exchange code runs with a neighbour: the concatenated slabs equal the whole-grid
    handle bit for bit, statistics and stop sweep included."""
    import subprocess
    import tempfile
    import test_gpu_parity as T
    rccl = os.path.join(os.path.dirname(_native.LIB_PATH), "librccl_emu.so")
    assert os.path.exists(rccl), rccl
    env = dict(os.environ, PVI_RCCL_LIB=rccl, PVI_EMU_THREADS="3")
    for case, world, overlap, tol in (("pendulum:61,61:9:float64", 1, False, 8.0), ("pendulum:61,61:9:float64", 2, False, 8.0),
                                      ("cartpole:11,12,9,10:5:float32", 2, True, 60.0), ("cartpole:11,12,9,10:5:float32", 3, True, 60.0),
                                      ("cartpole:11,12,9,10:5:float32", 3, False, 60.0)):
        with tempfile.TemporaryDirectory() as tmp:
            script = os.path.join(tmp, "rccl_rank.py")
            open(script, "w").write(T._RCCL_RANK % dict(root=ROOT))
            idfile = os.path.join(tmp, "comm.id")
            procs = [subprocess.Popen([sys.executable, script, str(r), str(world), case, idfile, os.path.join(tmp, "r%d.npz" % r), str(int(overlap)), repr(tol)],
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
            outs = [(p.wait(timeout=600), p.stdout.read()) for p in procs]
            assert all(rc == 0 and ("RCCL-RANK-OK %d" % r) in o for r, (rc, o) in enumerate(outs)), "\n".join(o[-1500:] for _, o in outs)
            parts = [np.load(os.path.join(tmp, "r%d.npz" % r)) for r in range(world)]
        cfg = build(case)
        with quiet():
            h = cfg["grid_sys"]._device_problem(cost=cfg["cf"].device_cost(), dtype=cfg["dtype"])
        h.terminal_cost()
        st5, _ = h.sweep(5, 1.0, -1.0)
        st, n = h.sweep(400, 1.0, tol)
        desc = str(parts[0]["desc"])
        print("%s world %d overlap %s: stop after %d sweeps; %s" % (case, world, overlap, n, desc[:110]))
        assert "comm=rccl" in desc and ("rank=0/%d" % world) in desc, desc
        assert 0 < n < 400 and all(int(p["n"]) == n for p in parts), (n, [int(p["n"]) for p in parts])
        assert np.array_equal(np.concatenate([p["J"] for p in parts]), h.get_J())
        assert np.array_equal(np.concatenate([p["pi"] for p in parts]), h.get_pi())
        for p in parts:
            assert np.allclose(p["st5"], st5[-1], rtol=1e-12) and np.allclose(p["st"], st[-1], rtol=1e-12)
        h.close()


CHECKS = {k[6:]: v for k, v in globals().items() if k.startswith("check_")}

if __name__ == "__main__":
    for name in sys.argv[1:] or list(CHECKS):
        print("== " + name, flush=True)
        CHECKS[name]()
        import ctypes
        L = ctypes.CDLL(_native.LIB_PATH)
        L.emu_launch_count.restype = L.emu_inactive_lane_reads.restype = ctypes.c_ulonglong
        print("== %s passed (%d kernel launches, %d reads of inactive lanes)" % (name, L.emu_launch_count(), L.emu_inactive_lane_reads()), flush=True)
