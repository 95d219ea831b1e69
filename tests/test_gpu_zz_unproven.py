"""
GPU tests of code paths that were written while no MI355X was reachable (round 5) and had not executed when they were
committed: the swapped internal order of the float32 cart-pole, error-feedback storage outside the 4-D window sweep
(2-D grids, explicit n = 3 systems, the node-table tier).  They sit in a file that collects LAST (the driver runs
`pytest -x`): a failure here must not hide the verified rows of tests/test_gpu_parity.py.  Same checkers, same tolerances.
"""
import contextlib
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["cartpole_11p4x5", "twolink_11p4x3x3", "doublependulum_13x11x13x11x3x3"])
def test_declared_invariance_and_clamped_validity_variants_agree(name):
    """Host logic of round 5 that tests/test_gpu_parity.py::test_f32_kernel_variants_agree does not cover in its last-run form: the
    4-D window sweep with the invariant axes of the displacement FOUND by k_lean4_invariance (TABLES=2) instead of declared by the
    cart-pole's closed form, and with cell validity by clamp-and-compare (VMASK=0) instead of set-up's bit per action -- the same
    bits as the default; the declared axes are the found ones; the handle says which form it runs."""
    from pyro_amd import _native
    from test_gpu_parity import CASES, load, native_problem, oracle_problem
    g = load(name)
    p = oracle_problem(g, *CASES[name])
    alpha = float(g["alpha"]) if "alpha" in g.files else 1.0
    outs = {}
    for tag, kv in (("lean", {"WIN": "1"}), ("tab2", {"WIN": "1", "TABLES": "2"}), ("clamp", {"WIN": "1", "VMASK": "0"})):
        with _native.overrides(**kv):
            h = native_problem(p, dtype="float32")
        h.terminal_cost()
        h.sweep(6, alpha, -1.0)
        outs[tag] = (h.get_J(), h.get_pi(), h.describe())
        h.close()
    tok = {t: dict(x.split("=", 1) for x in outs[t][2].split() if "=" in x) for t in outs}
    for t in ("tab2", "clamp"):
        assert outs[t][2].split()[0] == "path=lean" and tok[t]["win"] == "1", outs[t][2]
        assert np.array_equal(outs[t][0], outs["lean"][0]) and np.array_equal(outs[t][1], outs["lean"][1]), (t, outs[t][2])
    assert tok["tab2"]["tables"] == tok["lean"]["tables"], (outs["tab2"][2], outs["lean"][2])          # declared == found
    assert tok["lean"]["vmask"] == "1" and tok["clamp"]["vmask"] == "0", (outs["lean"][2], outs["clamp"][2])


def test_feedback_refusals_and_the_unproven_gate():
    """PVI_FLAG_F32_FEEDBACK where it is not offered: the table tier (class surface), an explicit system whose mask sweep is
    switched off (library), and -- until their tests have run on hardware -- 2-D grids and explicit systems without
    pvi_override("UNPROVEN", "1") (class surface: NotImplementedError; library: PVI_EINVAL naming the override)."""
    import table_case
    from pyro_amd import configs, _native
    from pyro_amd.planning import dynamicprogramming as DP
    with contextlib.redirect_stdout(io.StringIO()):
        tc = table_case.table_case()
        h3 = configs.build("h3s")
        c2 = configs.build("pendulum:41,41:5:float32")
    with pytest.raises(NotImplementedError):
        DP.DynamicProgrammingWithLookUpTable(tc["grid_sys"], tc["cf"], dtype="float32", f32_feedback=True)
    for cfg in (c2, h3):
        with pytest.raises(NotImplementedError):
            DP.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"], dtype="float32", f32_feedback=True)
        with pytest.raises(_native.NativeError) as ei:
            cfg["grid_sys"]._device_problem(cost=DP.device_cost_of(cfg["cf"], cfg["sys"]), dtype="float32", flags=_native.FLAG_F32_FEEDBACK)
        assert "UNPROVEN" in str(ei.value) and "PVI_FLAG_F32_FEEDBACK" in str(ei.value), ei.value
    with _native.overrides(NO_FAST="1", UNPROVEN="1"):
        with pytest.raises(_native.NativeError) as ei:
            h3["grid_sys"]._device_problem(cost=DP.device_cost_of(h3["cf"], h3["sys"]), dtype="float32", flags=_native.FLAG_F32_FEEDBACK)
    assert "PVI_FLAG_F32_FEEDBACK" in str(ei.value) and "UNPROVEN" not in str(ei.value), ei.value


def test_error_feedback_over_python_driven_slabs(tmp_path):
    """Round 5: error-feedback storage over the Python-driven slabs (three ranks on one GPU, every piece keeps the residuals of its
    rows, nothing about them is exchanged): the bits of the one-handle feedback solve."""
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    from pyro_amd import configs
    from pyro_amd.planning import dynamicprogramming
    import test_gpu_parity as T
    case, world = "cartpole:21,21,21,21:7:float32", 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "res.npz")
    code = T._WORLD2.replace('overlap=%s)', 'overlap=%s, f32_feedback=True)') % (ROOT, port, world, case, True)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    logs = [pr.communicate(timeout=900)[0] for pr in procs]
    assert all("WORLD2-OK" in lg for lg in logs), "\n".join(logs)
    r = np.load(out)
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build(case)
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"], dtype="float32", f32_feedback=True)
    stats, _ = dp._p.sweep(7, 1.0, -1.0)
    assert np.array_equal(r["J"], dp._p.get_J()) and np.array_equal(r["pi"], dp._p.get_pi())
    np.testing.assert_allclose(r["stats"][:4], stats[:4], rtol=1e-12)


def test_closed_loop_trajectory_of_a_solve_matches_the_reference():
    """Round 6 (the caller after the path): what the reference's scripts end with -- `cl = ctl + sys; cl.x0 = ...;
    cl.compute_trajectory(3.0, 121, 'euler')` with ctl = dp.get_lookup_table_controller() -- is ONE row of the rollout kernel
    (pvi_rollout; the code object of dp.simulate_closed_loop) plus the host bookkeeping of simulation.CLosedLoopSimulator: every
    field of the reference's Trajectory (tests/golden/trajectory_pendulum_21x21x5.npz).  A controller edited after the solve
    falls back to the host loop and gives the same numbers."""
    from conftest import GOLDEN
    import os
    from pyro_amd.analysis import costfunction
    from pyro_amd.control import controller
    from pyro_amd.dynamic import pendulum
    from pyro_amd.planning import discretizer, dynamicprogramming
    g = np.load(os.path.join(GOLDEN, "trajectory_pendulum_21x21x5.npz"))
    with contextlib.redirect_stdout(io.StringIO()):
        s = pendulum.SinglePendulum()
        grid = discretizer.GridDynamicSystem(s, [21, 21], [5])
        q = costfunction.QuadraticCostFunction.from_sys(s)
        q.xbar, q.INF = np.array([-3.14, 0.0]), 300
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(grid, q)
        dp.save_time_history = False
        dp.compute_steps(40)
        dp.clean_infeasible_set()
    assert np.array_equal(dp.pi, g["pi"])
    ctl = dp.get_lookup_table_controller()
    cl = ctl + s
    assert isinstance(cl, controller.ClosedLoopSystem)
    calls = []
    real = dp.simulate_closed_loop
    dp.simulate_closed_loop = lambda *a, **k: (calls.append(k.get("device_only")), real(*a, **k))[1]
    for k in (0, 1):
        cl.x0 = g["cl%d_x0" % k]
        tr = cl.compute_trajectory(3.0, 121, "euler")
        assert tr is cl.traj and calls[-1] is True                      # (the rollout kernel computed x and u)
        for name in ("t", "x", "u", "y", "r"):
            np.testing.assert_allclose(getattr(tr, name), g["cl%d_%s" % (k, name)], rtol=1e-8, atol=1e-8, err_msg=name)
        np.testing.assert_allclose(tr.dx, g["cl%d_dx" % k], rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(tr.dJ, g["cl%d_dJ" % k], rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(tr.J, g["cl%d_J" % k], rtol=1e-7, atol=1e-7)
    # a controller that is no longer the solve's: the host loop (the reference's own), the same trajectory
    n_dev = len(calls)
    ctl2 = dp.get_lookup_table_controller()
    ctl2._dp = None
    cl2 = controller.ClosedLoopSystem(s, ctl2)
    cl2.x0 = g["cl0_x0"]
    th = cl2.compute_trajectory(3.0, 121, "euler")
    assert len(calls) == n_dev
    for name in ("x", "u", "dx", "y", "r", "J", "dJ"):
        np.testing.assert_allclose(getattr(th, name), g["cl0_" + name], rtol=1e-10, atol=1e-10, err_msg=name)


def test_cubic_interpolation_through_the_class_surface_matches_reference_golden():
    """Round 6 (VERDICT r5 missing #5): dp.interpol_method = 'cubic' / 'cubic_legacy' on 2-D grids -- RegularGridInterpolator's
    order-3 spline, zero outside the box -- served by the spline sweep of the table tier (the code objects of
    DynamicProgramming2DRectBivariateSpline; new is only the class-side plumbing: the in-box mask as validity table).  Against the
    reference's own 'cubic_legacy' solves (SciPy's exact fit) at 1e-10 after sweeps 1, 2, 8, LUT class and cell-by-cell base class;
    the reference's 'cubic' solves (SciPy >= 1.13: iterative fit) are as far from this build as from 'cubic_legacy'."""
    from conftest import GOLDEN
    import os
    from pyro_amd.analysis import costfunction
    from pyro_amd.dynamic import pendulum
    from pyro_amd.planning import discretizer
    from pyro_amd.planning import dynamicprogramming as DP
    g = np.load(os.path.join(GOLDEN, "cubic_pendulum.npz"))
    for tag in ("a", "b"):
        with contextlib.redirect_stdout(io.StringIO()):
            s = pendulum.SinglePendulum()
            s.x_lb, s.x_ub, s.u_lb, s.u_ub = g[tag + "_x_lb"], g[tag + "_x_ub"], g[tag + "_u_lb"], g[tag + "_u_ub"]
            grid = discretizer.GridDynamicSystem(s, [int(d) for d in g[tag + "_dims"]], [int(d) for d in g[tag + "_udims"]], float(g[tag + "_dt"]))
            cf = costfunction.QuadraticCostFunction.from_sys(s)
            cf.Q, cf.R, cf.S, cf.xbar, cf.ubar = g[tag + "_Q"], g[tag + "_R"], g[tag + "_S"], g[tag + "_xbar"], g[tag + "_ubar"]
            cf.INF, cf.EPS = float(g[tag + "_INF"]), float(g[tag + "_EPS"])
            dp = DP.DynamicProgrammingWithLookUpTable(grid, cf)
            dp.save_time_history = False
            assert dp.tier == "fused"
            dp.interpol_method = "cubic_legacy" if tag == "a" else "cubic"
            assert dp.tier == "table" and dp._p.describe().startswith("path=spline")
            for k in range(1, 9):
                dp.compute_steps(1)
                if k in (1, 2, 8):
                    Jg = g["%s_J_%d" % (tag, k)]
                    assert np.abs(dp.J - Jg).max() <= 1e-10 * np.abs(Jg).max(), (tag, k)
                    clear = g["%s_gap_%d" % (tag, k)] > 1e-8
                    assert np.array_equal(dp.pi[clear], g["%s_pi_%d" % (tag, k)][clear])
                if k == 4:              # the other name of the same engine: nothing is rebuilt
                    engine = dp._p
                    dp.interpol_method = "cubic" if tag == "a" else "cubic_legacy"
                    assert dp._p is engine
            Ji, Jg = g[tag + "_J_8_iter"], g[tag + "_J_8"]
            assert np.abs(dp.J - Ji).max() <= 1.0001 * np.abs(Ji - Jg).max() + 1e-10 * np.abs(Jg).max()
            # back to the linear sweeps: the fused tier again, the solve goes on from the same J
            J8 = dp.J.copy()
            dp.interpol_method = "linear"
            assert dp.tier == "fused" and np.array_equal(dp.J, J8)
            if tag == "b":              # float32 storage of J around the same float64 fit and evaluation
                d32 = DP.DynamicProgrammingWithLookUpTable(grid, cf, dtype="float32")
                d32.save_time_history = False
                d32.interpol_method = "cubic"
                d32.compute_steps(8)
                assert "float" in d32._p.describe().split("kernel=")[1].split()[0]
                assert np.abs(d32.J - Jg).max() <= 1e-6 * np.abs(Jg).max()
            if tag == "a":
                db = DP.DynamicProgramming(grid, cf)
                db.save_time_history = False
                db.interpol_method = "cubic"
                db.compute_steps(2)
                assert np.abs(db.J - g["a_base_J_2"]).max() <= 1e-10 * np.abs(db.J).max()
                assert (db.pi != g["a_base_pi_2"]).mean() < 0.02
                with pytest.raises(NotImplementedError):
                    db.interpol_method = "quintic"
                assert db.interpol_method == "cubic"


def test_slinear_interpolation_through_the_class_surface_matches_reference_golden():
    """Round 6 (VERDICT r5 missing #5): dp.interpol_method = 'slinear' -- scipy's order-1 spline, the linear interpolant by another
    code path -- is served by the linear sweeps of the fused tier without rebuilding the engine; against the reference's own
    'slinear' solves (tests/golden/slinear_*.npz) at the float64 tolerance; 'quintic' / 'pchip' (and 'cubic' on the 4-D grid) raise."""
    from conftest import GOLDEN
    import os
    from pyro_amd.analysis import costfunction
    from pyro_amd.dynamic import cartpole, pendulum
    from pyro_amd.planning import discretizer
    from pyro_amd.planning import dynamicprogramming as DP
    for name, key, n in (("slinear_pendulum_31x21x5", "12", 12), ("slinear_cartpole_7x9x7x9x3", "4", 4)):
        g = np.load(os.path.join(GOLDEN, name + ".npz"))
        with contextlib.redirect_stdout(io.StringIO()):
            s = pendulum.SinglePendulum() if "pendulum" in name else cartpole.CartPole()
            s.x_lb, s.x_ub, s.u_lb, s.u_ub = g["x_lb"], g["x_ub"], g["u_lb"], g["u_ub"]
            grid = discretizer.GridDynamicSystem(s, [int(d) for d in g["dims"]], [int(d) for d in g["udims"]], float(g["dt"]))
            cf = costfunction.QuadraticCostFunction.from_sys(s)
            cf.Q, cf.R, cf.S, cf.xbar, cf.ubar, cf.INF, cf.EPS = g["Q"], g["R"], g["S"], g["xbar"], g["ubar"], float(g["INF"]), float(g["EPS"])
            dp = DP.DynamicProgrammingWithLookUpTable(grid, cf)
            dp.save_time_history = False
            dp.interpol_method = "slinear"
            path = dp._p.describe().split()[0]
            dp.compute_steps(n // 2)
            dp.interpol_method = "linear"                       # (the same engine serves both: no rebuild, the solve goes on)
            dp.interpol_method = "slinear"
            dp.compute_steps(n - n // 2)
        assert dp.interpol_method == "slinear" and dp.tier == "fused" and dp._p.describe().split()[0] == path
        Jg = g["J%s_slinear" % key]
        assert np.abs(dp.J - Jg).max() <= 1e-12 * np.abs(Jg).max(), name
        assert (dp.pi != g["pi%s_slinear" % key]).mean() < 1e-3
        for bad in ("quintic", "pchip") + (("cubic",) if "cartpole" in name else ()):      # ('cubic': 2-D grids only, its own test)
            with pytest.raises(NotImplementedError):
                dp.interpol_method = bad
            assert dp.interpol_method == "slinear"


@pytest.mark.parametrize("dims,nact,fb", [((31, 29, 27, 25), 21, False), ((41, 41, 41, 41), 21, False), ((31, 29, 27, 25), 9, True)])
def test_swapped_internal_order_matches_the_reference_order(dims, nact, fb):
    """Round 5 (opt-in): DynamicProgramming(internal_order="swapped") solves the float32 cart-pole with q = (theta, x) inside the
    engine (pyro_amd/planning/permuted.py; Dyn<PVI_DYN_CARTPOLE_SW>, a dynamics id of its own) so that the lanes of the 4-D window sweep
    run along the axis the displacement does not depend on.  Same problem: J within the float32 tolerance of the float64 solve in
    the reference's order at every checkpoint (with error feedback: 1e-6), the policy within the float64 Q-regret rule, the
    statistics of a sweep within 1e-5; the engine says what it is (order=swapped, the displacement table over axes 0 and 2, the
    window kernel)."""
    from oracle import c_oracle as CO
    from pyro_amd import configs
    from pyro_amd.planning import dynamicprogramming as DP
    import bench
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build("cartpole:%s:%d:float32" % (",".join(str(d) for d in dims), nact))
    g, cf = cfg["grid_sys"], cfg["cf"]

    def make(dt, order="reference"):
        with contextlib.redirect_stdout(io.StringIO()):
            dp = DP.DynamicProgrammingWithLookUpTable(g, cf, dtype=dt, f32_feedback=fb and dt == "float32", internal_order=order)
        dp.save_time_history = False
        dp.verbose = False
        return dp
    d64, d32, dsw = make("float64"), make("float32"), make("float32", "swapped")
    desc = dsw._p.describe()
    assert "order=swapped" in desc and "tables=10" in desc and "win=1" in desc, desc      # (`kernel=` names the LAST launch: checked below)
    assert np.allclose(dsw.J, d32.J, rtol=1e-6, atol=0.0)            # the terminal cost, transposed back
    worst = 0.0
    for k in range(5):
        st = {}
        for name, dp in (("f64", d64), ("f32", d32), ("sw", dsw)):
            stats, n = dp._p.sweep(60, 1.0, -1.0)
            st[name] = np.array(stats[-1])
        J64 = d64._p.get_J()
        m = np.abs(J64).max()
        e_sw, e_32 = np.abs(dsw._p.get_J() - J64).max() / m, np.abs(d32._p.get_J() - J64).max() / m
        worst = max(worst, e_sw)
        print("after %d sweeps: swapped %.3e reference order %.3e" % (60 * (k + 1), e_sw, e_32))
        assert e_sw <= (1e-6 if fb else 1e-5), (k, e_sw, e_32)
        assert np.allclose(st["sw"], st["f64"], rtol=1e-5, atol=1e-5 * m), (st["sw"], st["f64"])
    assert ("kernel=k_sweep_lean4fb<12," if fb else "kernel=k_sweep_lean4<12,") in dsw._p.describe(), dsw._p.describe()
    c = CO.CProblem(bench.oracle_problem(cfg))
    Jprev = d64._p.get_J(prev=True)
    nodes = np.arange(0, g.nodes_n, 5, dtype=np.int64)
    q, qmin = c.q_at(Jprev, nodes, dsw._p.get_pi()[nodes])
    ok = np.isfinite(q) & np.isfinite(qmin)
    assert np.array_equal(np.isfinite(q), np.isfinite(qmin)) and (q[ok] - qmin[ok]).max() <= 1e-5 * np.abs(Jprev).max(), (q[ok] - qmin[ok]).max()
    assert (dsw._p.get_pi() != d32._p.get_pi()).mean() < 0.02       # (ties and float32 roundings of another summation order)
    # a cost-to-go set through the class surface lands transposed: the next sweep of both orders agrees
    J0 = d64._p.get_J()
    for dp in (d32, dsw):
        dp.J = J0
        dp._flush()
        dp._p.sweep(1, 1.0, -1.0)
    assert np.abs(dsw._p.get_J() - d32._p.get_J()).max() <= 2e-6 * np.abs(J0).max()
    rel, mism = dsw._p.self_check(1.0)                               # the window sweep against the plain-gather kernel, swapped dynamics in both
    assert rel <= 1e-5, (rel, mism)
    with pytest.raises(NotImplementedError):
        make("float64", "swapped")
    with pytest.raises(NotImplementedError):
        dsw._p.rollout(np.zeros((1, 4)), 10, 0.1)
    for dp in (d64, d32, dsw):
        dp._p.close()


@pytest.mark.parametrize("name,sweeps,every", [("pendulum:401,401:21:float32", 600, 100), ("pendulum:201,201:201:float32", 300, 100),
                                              ("pendulum:1001,1001:51:float32", 300, 100)])
def test_f32_error_feedback_storage_on_2d_grids(name, sweeps, every):
    """VERDICT r4 missing #3 / next #5: error-feedback storage for the 2-D float32 sweep (k_sweep_leanfb: uniform walk and the
    lane-split walk of grids with many actions; C2 = BASELINE configs[1] at full size).  Every checkpoint within 1e-6 of the
    float64 iterates and no worse than plain float32 storage; the same bits from a restart; the policy within the Q-regret rule
    of the float32 path."""
    from pyro_amd import configs
    from pyro_amd.planning import dynamicprogramming as DP
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build(name)
    g, cf = cfg["grid_sys"], cfg["cf"]

    def make(dt, fb=False):
        from pyro_amd import _native
        with contextlib.redirect_stdout(io.StringIO()), _native.overrides(**({"UNPROVEN": "1"} if fb else {})):
            dp = DP.DynamicProgrammingWithLookUpTable(g, cf, dtype=dt, f32_feedback=fb)
        dp.save_time_history = False
        dp.verbose = False
        return dp
    d64, d32, dfb = make("float64"), make("float32"), make("float32", True)
    assert "feedback=1" in dfb._p.describe() and "feedback=0" in d32._p.describe(), dfb._p.describe()
    worst_fb = worst_plain = 0.0
    for k in range(sweeps // every):
        for dp in (d64, d32, dfb):
            dp._p.sweep(every, 1.0, -1.0)
        J64 = d64._p.get_J()
        m = np.abs(J64).max()
        e_fb, e_plain = np.abs(dfb._p.get_J() - J64).max() / m, np.abs(d32._p.get_J() - J64).max() / m
        worst_fb, worst_plain = max(worst_fb, e_fb), max(worst_plain, e_plain)
        print("%s after %d sweeps: feedback %.3e plain %.3e" % (name, every * (k + 1), e_fb, e_plain))
        assert e_fb <= 1e-6, (k, e_fb)
    assert "kernel=k_sweep_leanfb<" in dfb._p.describe(), dfb._p.describe()
    assert worst_fb <= worst_plain * 1.05 + 1e-9, (worst_fb, worst_plain)
    # the policy: float64 Q-regret of the feedback handle's actions on the float64 J of the sweep before
    Jprev, pi = d64._p.get_J(prev=True), dfb._p.get_pi()
    from oracle import c_oracle as CO
    import bench
    c = CO.CProblem(bench.oracle_problem(cfg))
    nodes = np.arange(0, g.nodes_n, max(1, g.nodes_n // 100000), dtype=np.int64)
    q, qmin = c.q_at(Jprev, nodes, pi[nodes])
    ok = np.isfinite(q) & np.isfinite(qmin)
    assert (q[ok] - qmin[ok]).max() <= 1e-5 * np.abs(Jprev).max()
    # a new terminal cost clears the residuals: the same bits as a fresh handle
    dfb.evaluate_terminal_cost()
    dfb._p.sweep(40, 1.0, -1.0)
    fresh = make("float32", True)
    fresh._p.sweep(40, 1.0, -1.0)
    assert np.array_equal(dfb._p.get_J(), fresh._p.get_J()) and np.array_equal(dfb._p.get_pi(), fresh._p.get_pi())


def test_f32_error_feedback_storage_of_an_explicit_system():
    """... and for the n = 3 systems (k_sweep3_fast: the reference's helicopter demo on a 101 x 101 x 201 grid, obstacles and
    domain-check cost): every checkpoint within 1e-6 of float64, no worse than plain float32 storage, a restart clears the
    residuals."""
    from pyro_amd import configs
    from pyro_amd.planning import dynamicprogramming as DP
    with contextlib.redirect_stdout(io.StringIO()):
        s, g, cf, _ = configs._helicopter((101, 101, 201), (11,), "float32")

    def make(dt, fb=False):
        from pyro_amd import _native
        with contextlib.redirect_stdout(io.StringIO()), _native.overrides(**({"UNPROVEN": "1"} if fb else {})):
            dp = DP.DynamicProgrammingWithLookUpTable(g, cf, dtype=dt, f32_feedback=fb)
        dp.save_time_history = False
        dp.verbose = False
        return dp
    d64, d32, dfb = make("float64"), make("float32"), make("float32", True)
    assert "path=fast3" in dfb._p.describe() and "feedback=1" in dfb._p.describe() and "feedback=0" in d32._p.describe(), dfb._p.describe()
    worst_fb = worst_plain = 0.0
    for k in range(4):
        for dp in (d64, d32, dfb):
            dp._p.sweep(100, 1.0, -1.0)
        J64 = d64._p.get_J()
        m = np.abs(J64).max()
        e_fb, e_plain = np.abs(dfb._p.get_J() - J64).max() / m, np.abs(d32._p.get_J() - J64).max() / m
        worst_fb, worst_plain = max(worst_fb, e_fb), max(worst_plain, e_plain)
        print("helicopter after %d sweeps: feedback %.3e plain %.3e" % (100 * (k + 1), e_fb, e_plain))
        assert e_fb <= 1e-6, (k, e_fb)
    assert worst_fb <= worst_plain * 1.05 + 1e-9
    dfb.evaluate_terminal_cost()
    dfb._p.sweep(30, 1.0, -1.0)
    fresh = make("float32", True)
    fresh._p.sweep(30, 1.0, -1.0)
    assert np.array_equal(dfb._p.get_J(), fresh._p.get_J())


def test_f32_error_feedback_storage_through_the_node_table_tier():
    """... and for a one-degree-of-freedom system of the generic mechanical tier (MountainCar: Dyn<PVI_DYN_NODE_1x1>)."""
    from pyro_amd.analysis import costfunction
    from pyro_amd.dynamic import mountaincar
    from pyro_amd.planning import discretizer
    from pyro_amd.planning import dynamicprogramming as DP
    with contextlib.redirect_stdout(io.StringIO()):
        s = mountaincar.MountainCar()
        g = discretizer.GridDynamicSystem(s, [201, 201], [11])
        cf = costfunction.QuadraticCostFunction.from_sys(s)
        cf.INF = 100
        from pyro_amd import _native
        with _native.overrides(UNPROVEN="1"):
            dps = {k: DP.DynamicProgrammingWithLookUpTable(g, cf, dtype=dt, f32_feedback=fb)
                   for k, dt, fb in (("f64", "float64", False), ("f32", "float32", False), ("fb", "float32", True))}
    for dp in dps.values():
        dp.save_time_history = False
        dp.verbose = False
    assert dps["fb"].tier == "fused" and "feedback=1" in dps["fb"]._p.describe()
    for k in range(4):
        for dp in dps.values():
            dp._p.sweep(100, 1.0, -1.0)
        J64 = dps["f64"]._p.get_J()
        m = np.abs(J64).max()
        e_fb, e_plain = (np.abs(dps[k_]._p.get_J() - J64).max() / m for k_ in ("fb", "f32"))
        print("mountain car after %d sweeps: feedback %.3e plain %.3e" % (100 * (k + 1), e_fb, e_plain))
        assert e_fb <= 1e-6


def _dp(name, dtype=None, fb=False):
    from pyro_amd import configs
    from pyro_amd.planning import dynamicprogramming as DP
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build(name)
        dp = DP.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"], dtype=dtype or cfg["dtype"], f32_feedback=fb)
    dp.save_time_history = False
    dp.verbose = False
    return dp


@pytest.mark.parametrize("name", ["cartpole:41,41,41,41:21:float32", "cartpole:31,33,35,37:31:float32", "twolink:21,21,21,21:5,5:float32"])
def test_feedback_corruption_detector_is_silent_on_a_sound_build_and_changes_no_bit(name):
    """Round 6 (VERDICT r5 next #3d): k_sweep_lean4fbc = the error-feedback sweep of 4-D grids with a detector in its epilogue -- the
    float32 action loop's value of the winner against the float64 re-evaluation of the same backup, |diff| <= 1e-4 (1 + |J|), else
    pvi_sweep returns PVI_ECORRUPT.  On a sound build it never fires (full-occupancy grids, a solve's worth of sweeps), J, pi, the
    residual-driven iterates and the statistics are the bits of k_sweep_lean4fb, and the handle says which kernel it runs.  (That it
    FIRES on the build of commit 4e5b14a: tools/r05_hunt/run_r06_detector.sh, which needs that build.)"""
    from pyro_amd import _native
    with _native.overrides(FBCHECK="1"):
        c = _dp(name, fb=True)._p
    p = _dp(name, fb=True)._p
    for n in (1, 40, 7, 120):
        sc, nc = c.sweep(n, 1.0, -1.0)                      # (PVI_ECORRUPT would raise NativeError here)
        sp, npl = p.sweep(n, 1.0, -1.0)
        assert nc == npl == n and np.array_equal(np.array(sc), np.array(sp)), n
        assert np.array_equal(c.get_J(), p.get_J()) and np.array_equal(c.get_pi(), p.get_pi()), n
    assert "kernel=k_sweep_lean4fbc<" in c.describe() and "kernel=k_sweep_lean4fb<" in p.describe(), c.describe()
    assert np.isfinite(c.get_J()).all()
    c.close()
    p.close()


@pytest.mark.parametrize("system,dims,nact,k", [("pendulum", (61, 57), 9, 2), ("pendulum", (61, 57), 9, 5), ("cartpole", (21, 19, 23, 17), 5, 3)])
@pytest.mark.parametrize("eps", [-1.5e-6, -4e-7, 0.0, 4e-7])
def test_float32_slabs_refuse_or_agree_when_the_reach_sits_on_a_whole_number_of_cells(system, dims, nact, k, eps):
    """the time step is chosen so that the largest axis-0 displacement of an in-box cell is k + eps cells, eps
    within +-1.5e-6 (the float32 kernels form x_next in float32: an ulp of a ten-cell displacement is 1e-6 cells).  With the halo
    the host rule agrees on (parallel._rows_for_reach: floor(d + 2e-6) + 1 rows) the two slabs must reproduce the whole-grid float32
    sweep bit for bit; with ONE ROW LESS the handle must be refused (set-up) or the sweep must end in PVI_EHALO -- never a clamped
    gather that passes silently."""
    from oracle import vi_oracle as O
    from pyro_amd import _native, parallel
    from test_gpu_parity import native_problem
    n0 = dims[0]
    if system == "pendulum":
        lb, ub, dyn, consts = [-3.2, -6.0], [3.2, 6.0], O.DYN_PENDULUM, O.pendulum_consts()
    else:
        lb, ub, dyn, consts = [-4.0, -3.2, -5.0, -6.0], [4.0, 3.2, 5.0, 6.0], O.DYN_CARTPOLE, O.cartpole_consts()
    n = len(dims)
    lv = O.make_levels(np.array(lb), np.array(ub), np.array(dims))
    ul = O.make_levels(np.array([-10.0]), np.array([10.0]), np.array([nact]))
    step0 = (ub[0] - lb[0]) / (n0 - 1)
    vmax = ub[n // 2]                                           # x_next_0 = x_0 + dq_0 dt: the reach is max |dq_0| dt
    dt = (k + eps) * step0 / vmax
    Q, R = np.eye(n), np.eye(1)
    p = O.Problem(lv, ul, dt, dyn, consts, Q, R, np.zeros((n, n)), np.zeros(n), np.zeros(1), 1e4, 0.2)
    need = parallel._rows_for_reach(k + eps)
    assert need == (k + 1 if eps >= -2e-6 else k)
    mid = n0 // 2
    whole = native_problem(p, dtype="float32")
    whole.terminal_cost()
    slabs = [native_problem(p, dtype="float32", rows=(0, mid), halo=(0, need)), native_problem(p, dtype="float32", rows=(mid, n0), halo=(need, 0))]
    for s in slabs:
        s.terminal_cost()
    for _ in range(3):
        whole.sweep(1, 1.0, -1.0)
        for s in slabs:
            s.sweep_async(1.0)
            s.sweep_stats()
        parts = [s.get_J() for s in slabs]
        assert np.array_equal(np.concatenate(parts), whole.get_J())
        full = np.concatenate(parts).reshape(n0, -1)
        for s in slabs:
            r0, r1 = s.store_rows
            s.set_J(full[r0:r1].ravel(), r0, r1 - r0)
    for s in slabs:
        s.close()
    # one row less than the rule: refused at create, reported by the sweep, or -- where every gather stays inside the rows that ARE
    # stored -- still the whole grid's bits; a silent difference is the failure
    whole.terminal_cost()
    whole.sweep(1, 1.0, -1.0)
    try:
        short = native_problem(p, dtype="float32", rows=(0, mid), halo=(0, need - 1))
    except _native.NativeError as e:
        assert "halo" in str(e).lower(), e
        whole.close()
        return
    short.terminal_cost()
    try:
        short.sweep_async(1.0)
        short.sweep_stats()
    except _native.NativeError as e:
        assert e.code == _native.PVI_EHALO, e
    else:
        # neither refused nor reported: every gather stayed inside the rows that ARE stored (x_next exactly on a level takes the
        # cell below it, whose upper corner is the last stored row) -- then the result must be the whole grid's, bit for bit
        assert np.array_equal(short.get_J(), whole.get_J()[:mid * int(np.prod(dims[1:]))]), (system, k, eps, short.describe())
    short.close()
    whole.close()


# last of all: the one piece that could leave a GPU spinning if it were wrong (its grid barrier gives up after about a second)
@pytest.mark.parametrize("name,lsplit", [("pendulum:201,201:201:float32", "1"), ("pendulum:201,201:201:float32", "0"),
                                         ("pendulum:201,201:21:float32", None), ("pendulum:101,101:11:float32", None),
                                         ("pendulum:301,151:51:float32", "0")])
def test_multi_sweep_launch_of_the_2d_float32_sweep_is_bit_identical(name, lsplit):
    """Round 5: batches of the 2-D float32 window sweep as ONE cooperative launch (k_sweep_leanm: write-through
    J, fence-free grid barrier, statistics records folded by every workgroup) against one launch per sweep (pvi_override MULTI=0):
    J, pi, every sweep's statistics and the stop sweep are the same bits -- fixed counts in odd batch sizes, then a tolerance
    stop in the middle of a batch, then a restart."""
    from pyro_amd import _native
    # `lsplit`: lanes per node = 2^lsplit.  The multi-sweep launch needs every workgroup resident and at most MULTI_MAX_WG = 512 of
    # them (multi32_applies, lean.hip): set-up's own choice for 201 x 201 x 201 actions is 8 lanes per node -- 1 407 workgroups,
    # the tiling of the round-4 bench line too -- which does not qualify; 2 lanes (316 workgroups) and 1 lane (158) do.
    ls = {} if lsplit is None else {"LSPLIT": lsplit}
    with _native.overrides(MULTI32="1", **ls):                # (the form of a handle's batches is decided at its first sweep)
        m = _dp(name)._p
        sm, nm = m.sweep(1, 1.0, -1.0)
    with _native.overrides(MULTI="0", **ls):
        s = _dp(name)._p
        ss, ns = s.sweep(1, 1.0, -1.0)
    tok = dict(t.split("=", 1) for t in m.describe().split() if "=" in t)
    if "multi=0" in m.describe():
        # more workgroups than MULTI_MAX_WG or than are resident at once: the multi-sweep launch does not apply to this handle and
        # one launch per sweep is what runs -- nothing to compare (the reason is part of the skip message)
        pytest.skip("grid of %s workgroups x %s threads: the multi-sweep launch does not apply (%s)" % (tok["grid"], tok["block"], m.describe()[:120]))
    assert "multi=1" in m.describe() and "kernel=k_sweep_leanm<" in m.describe(), m.describe()
    assert "multi=0" in s.describe() and "kernel=k_sweep_lean<" in s.describe(), s.describe()
    assert np.array_equal(np.array(sm), np.array(ss))
    for n in (1, 2, 3, 7, 40, 5, 64, 9):
        sm, nm = m.sweep(n, 1.0, -1.0)
        ss, ns = s.sweep(n, 1.0, -1.0)
        assert nm == ns == n and np.array_equal(np.array(sm), np.array(ss)), n
        assert np.array_equal(m.get_J(), s.get_J()) and np.array_equal(m.get_pi(), s.get_pi()), n
    # a stop in the middle of a batch: the tolerance is the delta a scout run of the one-launch-per-sweep handle reaches 45 sweeps
    # from here (round 6, found under emulation: "0.7 x the current delta" is never met within 400 sweeps on the 101 x 101 grid,
    # whose delta sits on a plateau for hundreds of sweeps -- BASELINE configs[0] needs 618)
    J_here = s.get_J()
    assert np.array_equal(J_here, m.get_J())
    scout, _ = s.sweep(45, 1.0, -1.0)
    tol = float(np.array(scout)[-1, 3]) * 1.0000001
    s.set_J(J_here)
    sm, nm = m.sweep(400, 1.0, tol)
    ss, ns = s.sweep(400, 1.0, tol)
    assert nm == ns and 0 < nm <= 45, (nm, ns, tol)
    assert np.array_equal(np.array(sm), np.array(ss))
    assert np.array_equal(m.get_J(), s.get_J()) and np.array_equal(m.get_pi(), s.get_pi())
    assert np.array_equal(m.get_J(prev=True), s.get_J(prev=True))
    for h in (m, s):
        h.terminal_cost()
    sm, _ = m.sweep(25, 1.0, -1.0)
    ss, _ = s.sweep(25, 1.0, -1.0)
    assert np.array_equal(np.array(sm), np.array(ss)) and np.array_equal(m.get_J(), s.get_J())
    m.close()
    s.close()
