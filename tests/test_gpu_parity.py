"""
GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C ABI
(pyro_amd._native -> libpyrovi.so); the oracle and the golden fixtures are only the checkers.

Tolerances
  float64 path: the kernels mirror the oracle operation for operation (same trig tables, no
      implicit FMA), so J is compared at 1e-12 relative and is normally bit-identical; pi exact.
  float32 path: max|J_gpu - J_ref| / max|J_ref| <= 1e-5 (BASELINE.json north_star), pi judged
      by Q-regret (float64 Q of the chosen action minus the float64 minimum, on the GPU's own previous J,
      <= 1e-5 max|J|) because f32 rounding legitimately flips near-ties.
"""
import contextlib
import io
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import vi_oracle as O

pytestmark = pytest.mark.gpu

TWOLINK = dict(l1=0.5, lc1=0.2, lc2=0.1, m1=1, I1=0, m2=1, I2=0, gravity=9.81, d1=0.5, d2=0.5)
DOUBLEP = dict(l1=1, lc1=1, lc2=1, m1=1, I1=0, m2=1, I2=0, gravity=9.81, d1=0, d2=0)
REL_F32 = 1e-5


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def oracle_problem(g, dyn, consts):
    lv = O.make_levels(g["x_lb"], g["x_ub"], g["dims"])
    ul = O.make_levels(g["u_lb"], g["u_ub"], g["udims"])
    return O.Problem(lv, ul, float(g["dt"]), dyn, consts, g["Q"], g["R"], g["S"], g["xbar"], g["ubar"],
                     float(g["INF"]), float(g["EPS"]))


def trig_for(p):
    t = p.trig_tables()
    if p.dyn_id == O.DYN_PENDULUM:
        return (t["s0"],)
    if p.dyn_id == O.DYN_CARTPOLE:
        return (t["c1"], t["s1"])
    return (t["s0"], t["c1"], t["s1"], t["s01"])


def native_problem(p, dtype="float64", **kw):
    from pyro_amd import _native
    cost = dict(Q=p.Q, R=p.R, S=p.S, xbar=p.xbar, ubar=p.ubar, EPS=p.EPS, INF=p.INF, ontarget_check=p.ontarget_check)
    return _native.Problem(p.levels, p.u_levels, p.x_lb, p.x_ub, p.u_lb, p.u_ub, p.dt, dtype=dtype,
                           dynamics_id=p.dyn_id, dyn_params=list(p.dyn_c), trig=trig_for(p), cost=cost, **kw)


class _Overrides:
    """monkeypatch-like front end of pvi_override: setenv("PVI_X", v) pins variant key X, delenv removes it."""

    def setenv(self, key, value):
        from pyro_amd import _native
        _native.override(key[4:] if key.startswith("PVI_") else key, value)

    def delenv(self, key, raising=False):
        from pyro_amd import _native
        _native.override(key[4:] if key.startswith("PVI_") else key, None)


@pytest.fixture
def variants():
    from pyro_amd import _native
    _native.override()
    yield _Overrides()
    _native.override()


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


CASES = {
    "pendulum_21x21x5": (O.DYN_PENDULUM, O.pendulum_consts()),
    "pendulum_demo_51x51x9": (O.DYN_PENDULUM, O.pendulum_consts()),
    "pendulum_lowdef_41x21x3": (O.DYN_PENDULUM, O.pendulum_consts()),
    "cartpole_11p4x5": (O.DYN_CARTPOLE, O.cartpole_consts()),
    "twolink_11p4x3x3": (O.DYN_TWOLINK, O.twolink_consts(**TWOLINK)),
    "twolink_11p4x3x3_dt01": (O.DYN_TWOLINK, O.twolink_consts(**TWOLINK)),
    "doublependulum_13x11x13x11x3x3": (O.DYN_TWOLINK, O.twolink_consts(**DOUBLEP)),
}


# ------------------------------------------------------------------------------------------------ f
@pytest.mark.parametrize("key,dyn,consts", [
    ("pendulum", O.DYN_PENDULUM, O.pendulum_consts()),
    ("inverted", O.DYN_PENDULUM, O.pendulum_consts(inverted=True)),
    ("cartpole", O.DYN_CARTPOLE, O.cartpole_consts()),
    ("twolink", O.DYN_TWOLINK, O.twolink_consts(**TWOLINK)),
    ("doublependulum", O.DYN_TWOLINK, O.twolink_consts(**DOUBLEP)),
])
def test_eval_f_matches_reference_kat(key, dyn, consts):
    from pyro_amd import _native
    g = load("f_kat")
    dX = _native.eval_f(dyn, consts, g[key + "_X"], g[key + "_U"])
    ref = g[key + "_dX"]
    assert (np.abs(dX - ref) / np.maximum(1.0, np.abs(ref))).max() < 1e-11   # device sin/cos: ~1 ulp


# ------------------------------------------------------------------------------------- tables (a3/a4/a8)
@pytest.mark.parametrize("name", list(CASES))
def test_tables_match_oracle_bitwise(name):
    g = load(name)
    p = oracle_problem(g, *CASES[name])
    h = native_problem(p)
    xn, xok, aok, G = h.build_tables()
    oxn, oxok, oaok, oG = O.cells(p, np.arange(p.nodes_n))
    assert np.array_equal(xn, oxn)
    assert np.array_equal(xok, oxok) and np.array_equal(aok, oaok)
    assert np.array_equal(G, oG)
    h.close()


def test_tables_match_reference_golden():
    g = load("pendulum_21x21x5")
    p = oracle_problem(g, *CASES["pendulum_21x21x5"])
    h = native_problem(p)
    xn, xok, aok, G = h.build_tables()
    assert np.array_equal(xn, g["x_next_table"])
    assert np.array_equal(xok, g["x_next_isok"]) and np.array_equal(aok, g["action_isok"])
    np.testing.assert_allclose(G, g["G"], rtol=1e-14, atol=1e-14)
    h.close()


# ------------------------------------------------------------------------------------- sweeps, float64
@pytest.mark.parametrize("name", list(CASES))
def test_sweeps_f64_match_oracle(name):
    g = load(name)
    p = oracle_problem(g, *CASES[name])
    alpha = float(g["alpha"]) if "alpha" in g.files else 1.0
    nsw = min(int(g["sweeps"]), 10) if "sweeps" in g.files else 10
    h = native_problem(p)
    h.terminal_cost()
    J = O.terminal_cost(p)
    assert np.array_equal(h.get_J(), J)
    for k in range(nsw):
        Jn, pi = O.sweep(p, J, alpha)
        st, delta = O.sweep_stats(Jn, J)
        stats, n = h.sweep(1, alpha, -1.0)
        assert n == 1
        Jg, pig = h.get_J(), h.get_pi()
        assert relerr(Jg, Jn) < 1e-12, "sweep %d" % (k + 1)
        bad = pig != pi
        if bad.any():
            assert O.q_regret(p, J, pig, alpha)[bad].max() < 1e-9
        np.testing.assert_allclose(stats[0], list(st) + [delta], rtol=1e-12, atol=1e-12)
        assert np.array_equal(h.get_J(prev=True), J) or relerr(h.get_J(prev=True), J) < 1e-12
        J = Jg            # continue from the device state (keeps the comparison one-step)
    h.close()


def test_config1_solve_f64_matches_reference():
    """BASELINE configs[0]: 101x101x11 pendulum to tol 0.1 -> 618 sweeps, J* and pi* of the reference."""
    g = load("config1_pendulum_101x101x11")
    p = oracle_problem(g, O.DYN_PENDULUM, O.pendulum_consts())
    h = native_problem(p)
    h.terminal_cost()
    stats, n = h.sweep(5000, 1.0, 0.1)
    assert n == int(g["sweeps"]) == 618
    J, pi = h.get_J(), h.get_pi()
    assert relerr(J, g["J"]) < 1e-12
    bad = pi != g["pi"]
    assert bad.mean() < 1e-3
    if bad.any():
        assert O.q_regret(p, g["J_prev"], pi)[bad].max() < 1e-9
    assert stats[-1, 3] <= 0.1 < stats[-2, 3]
    h.close()


# ------------------------------------------------------------------------------------- sweeps, float32
def test_config1_solve_f32_within_tolerance():
    g = load("config1_pendulum_101x101x11")
    p = oracle_problem(g, O.DYN_PENDULUM, O.pendulum_consts())
    h = native_problem(p, dtype="float32")
    h.terminal_cost()
    stats, n = h.sweep(5000, 1.0, 0.1)
    assert abs(n - 618) <= 2
    J, pi = h.get_J(), h.get_pi()
    assert relerr(J, g["J"]) <= REL_F32
    # policy: float64 Q-regret of the GPU's action on the GPU's own previous cost-to-go (f32 rounding flips near-ties)
    reg = O.q_regret(p, h.get_J(prev=True), pi)
    assert reg.max() <= 1e-5 * np.abs(g["J"]).max()
    h.close()


def test_cartpole_21p4_f32_golden():
    """Cart-pole 21^4 x 7, 20 sweeps against the reference's own J (stored as f32)."""
    g = load("cartpole_21p4x7")
    p = oracle_problem(g, O.DYN_CARTPOLE, O.cartpole_consts())
    for dtype, tol in (("float64", 1e-6), ("float32", REL_F32)):
        h = native_problem(p, dtype=dtype)
        h.terminal_cost()
        stats, n = h.sweep(int(g["sweeps"]), 1.0, -1.0)
        assert n == 20
        J, pi = h.get_J(), h.get_pi()
        assert relerr(J, g["J"].astype(np.float64)) <= tol
        # the policy by its float64 Q-regret on the GPU's own previous cost-to-go (no mismatch allowance --
        # an action is right when no other is better by more than the tolerance, whatever ties rounding flips)
        from oracle import c_oracle as CO
        nodes = np.arange(pi.size, dtype=np.int64)
        q, qmin = CO.CProblem(p).q_at(h.get_J(prev=True), nodes, pi.astype(np.int64))
        ok = np.isfinite(q) & np.isfinite(qmin)
        assert np.array_equal(np.isfinite(q), np.isfinite(qmin))
        assert (q[ok] - qmin[ok]).max() <= (1e-9 if dtype == "float64" else 1e-5) * np.abs(g["J"]).max(), (dtype, (q[ok] - qmin[ok]).max())
        if dtype == "float64":
            assert (pi != g["pi"]).mean() < 1e-4          # (ties at the last bit of LAPACK's inverse in the reference's own run)
        h.close()


@pytest.mark.parametrize("name", ["cartpole_11p4x5", "twolink_11p4x3x3", "doublependulum_13x11x13x11x3x3"])
def test_4d_f32_within_tolerance(name):
    g = load(name)
    p = oracle_problem(g, *CASES[name])
    alpha = float(g["alpha"])
    h = native_problem(p, dtype="float32")
    h.terminal_cost()
    h.sweep(int(g["sweeps"]), alpha, -1.0)
    assert relerr(h.get_J(), g["J"]) <= REL_F32
    h.close()


# ------------------------------------------------------------------------------------- tier B (tables)
def test_table_tier_on_reference_tables():
    """dynamicprogramming.py:564-570 on the reference's own x_next_table / G."""
    from pyro_amd import _native
    g = load("pendulum_21x21x5")
    lv = O.make_levels(g["x_lb"], g["x_ub"], g["dims"])
    ul = O.make_levels(g["u_lb"], g["u_ub"], g["udims"])
    h = _native.Problem(lv, ul, g["x_lb"], g["x_ub"], g["u_lb"], g["u_ub"], float(g["dt"]),
                        dynamics_id=_native.DYN_TABLE)
    h.set_tables(g["x_next_table"], g["G"])
    h.set_J(g["J0"])
    for k in range(1, 11):
        h.sweep(1, 1.0, -1.0)
        if k in (1, 2, 10):
            assert relerr(h.get_J(), g["J_%d" % k]) < 1e-14
            assert np.array_equal(h.get_pi(), g["pi_%d" % k])
    h.close()


# ------------------------------------------------------------------------------------- slabs (multi-GPU building block)
@pytest.mark.parametrize("name,dtype", [("pendulum_demo_51x51x9", "float64"), ("cartpole_11p4x5", "float32")])
def test_slab_sweeps_equal_whole_grid(name, dtype):
    """Two handles owning disjoint row slabs (+halo) reproduce the whole-grid sweep bit for bit."""
    g = load(name)
    p = oracle_problem(g, *CASES[name])
    whole = native_problem(p, dtype=dtype)
    whole.terminal_cost()
    N0 = p.dims[0]
    mid, halo = N0 // 2, 6
    slabs = [native_problem(p, dtype=dtype, rows=(0, mid), halo=(0, halo)),
             native_problem(p, dtype=dtype, rows=(mid, N0), halo=(halo, 0))]
    for s in slabs:
        s.terminal_cost()
    for _ in range(4):
        whole.sweep(1, 1.0, -1.0)
        Jw = whole.get_J()
        for s in slabs:
            s.sweep_async(1.0)
            s.sweep_stats()
        parts = [s.get_J() for s in slabs]
        assert np.array_equal(np.concatenate(parts), Jw)
        # "halo exchange" through the host: every slab receives the rows it stores but does not own
        full = np.concatenate(parts).reshape(N0, -1)
        for s in slabs:
            r0, r1 = s.store_rows
            s.set_J(full[r0:r1].ravel(), r0, r1 - r0)
        assert np.array_equal(np.concatenate([s.get_pi() for s in slabs]), whole.get_pi())
    for s in slabs + [whole]:
        s.close()


def test_halo_too_small_is_reported():
    from pyro_amd import _native
    g = load("pendulum_demo_51x51x9")
    p = oracle_problem(g, *CASES["pendulum_demo_51x51x9"])
    s = native_problem(p, rows=(0, 25), halo=(0, 0))
    s.terminal_cost()
    s.sweep_async(1.0)
    with pytest.raises(_native.NativeError) as e:
        s.sweep_stats()
    assert e.value.code == _native.PVI_EHALO
    s.close()


# ------------------------------------------------------------------------------------- class surface
def test_class_surface_config1():
    """The drop-in classes, driven like examples/demos_by_tool/lqr_vs_valueiteration_for_a_simple_pendulum.py."""
    from pyro_amd.analysis import costfunction
    from pyro_amd.dynamic import pendulum
    from pyro_amd.planning import discretizer, dynamicprogramming
    g = load("config1_pendulum_101x101x11")
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        sys_ = pendulum.SinglePendulum()
        sys_.xbar = np.array([-3.14, 0])
        qcf = costfunction.QuadraticCostFunction.from_sys(sys_)
        qcf.INF = 300
        grid_sys = discretizer.GridDynamicSystem(sys_, [101, 101], [11])
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(grid_sys, qcf)
        dp.save_time_history = False
        dp.solve_bellman_equation()
    assert dp.k == 618
    assert "618 t:-30.90" in buf.getvalue()
    assert relerr(dp.J, g["J"]) < 1e-12
    assert (dp.pi != g["pi"]).mean() < 1e-3
    assert relerr(dp.J_next, g["J_prev"]) < 1e-12
    assert dp.J.dtype == np.float64 and dp.pi.dtype == np.int64


def test_class_surface_small_with_history_and_controller():
    from pyro_amd.analysis import costfunction
    from pyro_amd.dynamic import pendulum
    from pyro_amd.planning import discretizer, dynamicprogramming
    g = load("pendulum_21x21x5")
    with contextlib.redirect_stdout(io.StringIO()):
        sys_ = pendulum.SinglePendulum()
        grid_sys = discretizer.GridDynamicSystem(sys_, [21, 21], [5])
        qcf = costfunction.QuadraticCostFunction.from_sys(sys_)
        qcf.xbar = np.array([-3.14, 0.0])
        qcf.INF = 300
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(grid_sys, qcf)
        assert np.array_equal(grid_sys.x_next_table, g["x_next_table"])
        assert np.array_equal(grid_sys.x_next_isok, g["x_next_isok"])
        assert np.array_equal(grid_sys.action_isok, g["action_isok"])
        np.testing.assert_allclose(dp.G, g["G"], rtol=1e-14, atol=1e-14)
        dp.compute_steps(10)
        assert len(dp.J_list) == 11 and len(dp.pi_list) == 11
        np.testing.assert_allclose(dp.J_list[1], g["J_1"], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(dp.J, g["J_10"], rtol=1e-13, atol=1e-13)
        assert np.array_equal(dp.pi, g["pi_10"])
        dp.clean_infeasible_set()
        np.testing.assert_allclose(dp.J, g["J_clean"], rtol=1e-13)
        assert np.array_equal(dp.pi, g["pi_clean"])
        import matplotlib
        matplotlib.use("Agg")
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            dp.plot_cost2go(); dp.update_cost2go_plot(); dp.plot_policy(); dp.update_policy_plot(); dp.plot_cost2go_3D()
            ani = dp.animate_cost2go(show=False)
        assert len(dp.J_list) == 11
        import matplotlib.pyplot as plt
        plt.close("all")
        dp.J, dp.pi = g["J_clean"].copy(), g["pi_clean"].copy()
        ctl = dp.get_lookup_table_controller()
        u = np.array([ctl.c(x, 0) for x in g["ctl_x"]])
    np.testing.assert_allclose(u, g["ctl_u"], rtol=1e-12, atol=1e-12)


def test_class_surface_generic_system_uses_table_tier():
    """A system the kernels do not know (plain Python f) still runs: tables on the host like the
    reference, sweeps on the GPU."""
    from pyro_amd.analysis import costfunction
    from pyro_amd.dynamic import system
    from pyro_amd.planning import discretizer, dynamicprogramming

    class Integrator(system.ContinuousDynamicSystem):
        def __init__(self):
            super().__init__(2, 1, 2)
            self.name = "Double integrator"
            self.x_ub, self.x_lb = np.array([2.0, 2.0]), np.array([-2.0, -2.0])

        def f(self, x, u, t=0):
            return np.array([x[1], u[0]])

    with contextlib.redirect_stdout(io.StringIO()):
        s = Integrator()
        grid = discretizer.GridDynamicSystem(s, [21, 21], [3], dt=0.1)
        cf = costfunction.QuadraticCostFunction.from_sys(s)
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(grid, cf)
        assert dp.tier == "table"
        dp.compute_steps(5)
    J = np.array([cf.h(x, 0) for x in grid.state_from_node_id], dtype=float)
    for _ in range(5):
        J, pi, _ = O.sweep_lut(grid.x_level, grid.x_next_table, dp.G, J)
    assert relerr(dp.J, J) < 1e-13
    assert np.array_equal(dp.pi, pi)


# ------------------------------------------------------------------------------------- f32 kernel variants
@pytest.mark.parametrize("name", ["pendulum_demo_51x51x9", "cartpole_11p4x5", "twolink_11p4x3x3", "doublependulum_13x11x13x11x3x3"])
def test_f32_kernel_variants_agree(name, variants):
    monkeypatch = variants
    """lean (LDS window, precomputed coefficients), tile, fast and action-split variants compute the same
    recursion; they may differ in float32 summation order only."""
    g = load(name)
    p = oracle_problem(g, *CASES[name])
    alpha = float(g["alpha"]) if "alpha" in g.files else 1.0
    nsw = 6
    ref = O.terminal_cost(p)
    for _ in range(nsw):
        ref, _ = O.sweep(p, ref, alpha)
    outs = {}
    four_d = len(p.levels) == 4
    for tag, env in [("lean", {}),
                     # the generic LDS-window kernel (2-D grids; 4-D grids only with WIN=0): lanes per node, nodes per thread
                     ("lean_split", {"PVI_LSPLIT": "2", "PVI_WIN": "0"}), ("lean_nosplit", {"PVI_LSPLIT": "0", "PVI_WIN": "0"}),
                     ("lean_npt2", {"PVI_LSPLIT": "0", "PVI_NPT": "2", "PVI_WIN": "0"}),
                     # the 4-D kernel of round 3 (position-paired window, split displacement): forced on, with per-node and with
                     # factorised coefficient tables, with the plain launch order, with another tile shape
                     ("lean_win1", {"PVI_WIN": "1"}), ("lean_tab0", {"PVI_WIN": "1", "PVI_TABLES": "0"}),
                     ("lean_tab1", {"PVI_WIN": "1", "PVI_TABLES": "1"}), ("lean_noxcd", {"PVI_WIN": "1", "PVI_NO_XCD": "1"}),
                     ("lean_shape", {"PVI_WIN": "1", "PVI_TV0": "3", "PVI_TV1": "7"}),
                     # round 4: other bands of the launch order (the persistent and quad-window kernels of this round were
                     # measured, lost and removed: DESIGN.md 4.2b)
                     ("lean_bands2", {"PVI_WIN": "1", "PVI_BANDS": "2"}),
                     ("fast", {"PVI_NO_LEAN": "1"}),
                     ("exact32", {"PVI_NO_FAST": "1"})]:
        for k in ("PVI_LSPLIT", "PVI_NO_LEAN", "PVI_NO_FAST", "PVI_NPT", "PVI_WIN", "PVI_TABLES", "PVI_NO_XCD", "PVI_TV0", "PVI_TV1",
                  "PVI_BANDS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        h = native_problem(p, dtype="float32")
        h.terminal_cost()
        h.sweep(nsw, alpha, -1.0)
        desc = h.describe()                              # (after the sweeps: `kernel=` names the kernel of the last launch)
        outs[tag] = (h.get_J(), h.get_pi(), desc)
        h.close()
    if four_d:
        # the tiling a create chose, taken back by L4PIN (what the counter passes do: tools/tools_counters.sh): the same launch
        tok = dict(t.split("=", 1) for t in outs["lean"][2].split() if "=" in t)
        assert tok.get("choice", "-") != "-", outs["lean"][2]
        for k in ("PVI_LSPLIT", "PVI_NO_LEAN", "PVI_NO_FAST", "PVI_NPT", "PVI_WIN", "PVI_TABLES", "PVI_NO_XCD", "PVI_TV0", "PVI_TV1",
                  "PVI_BANDS"):
            monkeypatch.delenv(k, raising=False)
        monkeypatch.setenv("PVI_L4PIN", tok["choice"])
        h = native_problem(p, dtype="float32")
        h.terminal_cost()
        h.sweep(nsw, alpha, -1.0)
        pinned = dict(t.split("=", 1) for t in h.describe().split() if "=" in t)
        assert [pinned[k] for k in ("tile", "grid", "block", "lds_bytes", "choice")] == [tok[k] for k in ("tile", "grid", "block", "lds_bytes", "choice")]
        assert np.array_equal(h.get_J(), outs["lean"][0]) and np.array_equal(h.get_pi(), outs["lean"][1])
        h.close()
        monkeypatch.delenv("PVI_L4PIN", raising=False)
    for tag, (J, pi, desc) in outs.items():
        assert relerr(J, ref) <= REL_F32, (tag, desc)
        assert relerr(J, outs["exact32"][0]) <= 2e-6, (tag, desc)
    def path_of(desc):
        return desc.split()[0]
    # every variant must have taken the path its switches select (the default two-link arm is the one fixture whose
    # float32 displacement operands cancel: it is routed to float64 dynamics on purpose, see DESIGN.md numerics)
    if name == "twolink_11p4x3x3":
        # the default two-link arm: its float32 displacement operands reach thousands of cells and cancel, so the generic
        # kernel (float32 ta + tB u) hands it to float64 dynamics on purpose (DESIGN.md numerics) ...
        for tag in ("lean_split", "lean_nosplit", "lean_npt2"):
            assert path_of(outs[tag][2]) == "path=exact-f32" and "float64 dynamics" in outs[tag][2], outs[tag][2]
    else:
        assert path_of(outs["lean_split"][2]) == "path=lean" and "lsplit=2" in outs["lean_split"][2]
        assert path_of(outs["lean_nosplit"][2]) == "path=lean" and "lsplit=0" in outs["lean_nosplit"][2]
        assert "win=0" in outs["lean_nosplit"][2], outs["lean_nosplit"][2]
        # two nodes per thread (2-D grids; bands of the tile walked one after the other): the same arithmetic per node
        assert ("npt=%d" % (2 if not four_d else 1)) in outs["lean_npt2"][2], outs["lean_npt2"][2]
        assert np.array_equal(outs["lean_npt2"][0], outs["lean_nosplit"][0]) and np.array_equal(outs["lean_npt2"][1], outs["lean_nosplit"][1])
        assert path_of(outs["fast"][2]) == "path=fast"
    if four_d:
        # ... while the round-3 kernel splits every operand into integer + fraction in float64 at set-up and takes the arm too.
        # Whatever the coefficient tables, the launch order or the tile shape: the same bits.
        for tag in ("lean", "lean_win1", "lean_tab0", "lean_tab1", "lean_noxcd", "lean_shape", "lean_bands2"):
            assert path_of(outs[tag][2]) == "path=lean" and "win=1" in outs[tag][2], (tag, outs[tag][2])
            assert np.array_equal(outs[tag][0], outs["lean_win1"][0]) and np.array_equal(outs[tag][1], outs["lean_win1"][1]), (tag, outs[tag][2])
        assert "tables=0" in outs["lean_tab0"][2] and "gx=node" in outs["lean_tab0"][2], outs["lean_tab0"][2]
        assert "tile=3x" in outs["lean_shape"][2], outs["lean_shape"][2]       # (columns are evened out over the tiles)
        assert "kernel=k_sweep_lean4<" in outs["lean"][2], outs["lean"][2]
        # its split displacement is the more accurate float32 form: at least as close to the float64 oracle as the others
        assert relerr(outs["lean_win1"][0], ref) <= max(relerr(outs["lean_nosplit"][0], ref), 5e-7), \
            (relerr(outs["lean_win1"][0], ref), relerr(outs["lean_nosplit"][0], ref))
    else:
        assert path_of(outs["lean"][2]) == "path=lean" and "win=0" in outs["lean"][2], outs["lean"][2]
    assert path_of(outs["exact32"][2]) == "path=exact-f32"


def test_exact_f32_sparse_walk_is_bit_identical(variants):
    monkeypatch = variants
    """float32 storage with float64 dynamics (systems whose float32 displacement cancels: the two-link arm): the walk over
    the per-node validity masks against the dense action loop of the same kernel -- J, pi and statistics bit for bit,
    whole grid and a slab with halos."""
    for name in ("twolink_11p4x3x3", "doublependulum_13x11x13x11x3x3"):
        g = load(name)
        p = oracle_problem(g, *CASES[name])
        alpha = float(g["alpha"]) if "alpha" in g.files else 1.0
        outs = {}
        for tag, env in (("dense", {"PVI_SPARSE": "0"}), ("sparse", {"PVI_SPARSE": "1"})):
            monkeypatch.setenv("PVI_NO_FAST", "1")           # the exact float32 path whatever the displacement magnitude
            monkeypatch.setenv("PVI_SPARSE", env["PVI_SPARSE"])
            h = native_problem(p, dtype="float32")
            desc = h.describe()
            assert desc.startswith("path=exact-f32") and ("sparse=1" if tag == "sparse" else "sparse=0") in desc, desc
            h.terminal_cost()
            stats, n = h.sweep(6, alpha, -1.0)
            outs[tag] = [h.get_J(), h.get_pi(), stats.copy()]
            h.close()
            N0 = p.dims[0]
            r0, r1 = N0 // 3, max(N0 // 3 + 2, (2 * N0) // 3)
            hs = native_problem(p, dtype="float32", rows=(r0, r1), halo=(r0, N0 - r1))
            hs.terminal_cost()
            hs.sweep_async(alpha)
            hs.sweep_stats()
            outs[tag] += [hs.get_J(), hs.get_pi()]
            hs.close()
        for a, b in zip(outs["dense"], outs["sparse"]):
            assert np.array_equal(a, b), name
    monkeypatch.delenv("PVI_NO_FAST", raising=False)
    monkeypatch.delenv("PVI_SPARSE", raising=False)


@pytest.mark.parametrize("name", list(CASES) + ["edge:" + k for k in ("A300_u16_policy", "minimal_2x2_A1", "cartpole_ragged_fancy_cost", "doublependulum_72_actions", "everything_out_of_bounds", "tiny_box")])
def test_f64_second_form_is_bit_identical(name, variants):
    monkeypatch = variants
    """k_sweep64 (tabulated-reciprocal fractions, hoisted position weights, one validity compare per bound, skipped
    action loops where the position row leaves the box) against k_sweep, which mirrors the oracle operation for
    operation: the same J, pi and statistics, bit for bit -- whole grids and slabs with halos."""
    if name.startswith("edge:"):
        key = name[5:]
        p, alpha = _custom_problem(**dict(EDGE_CASES[key]))
        nsw = 4
    else:
        g = load(name)
        p = oracle_problem(g, *CASES[name])
        alpha = float(g["alpha"]) if "alpha" in g.files else 1.0
        nsw = 6
    outs = {}
    # (4-D grids: "sparse" walks a per-node validity mask built at set-up instead of all A actions -- forced on and off
    #  here, chosen by the share of cells in the box otherwise)
    for tag, env in (("v1", {"PVI_NO_SWEEP64": "1"}), ("v2", {"PVI_PATCH": "1", "PVI_SPARSE": "0"}),
                     ("v2line", {"PVI_PATCH": "0", "PVI_SPARSE": "0"}), ("v2sparse", {"PVI_PATCH": "1", "PVI_SPARSE": "1"}),
                     ("v2sparseline", {"PVI_PATCH": "0", "PVI_SPARSE": "1"}), ("v2auto", {})):
        for k in ("PVI_NO_SWEEP64", "PVI_PATCH", "PVI_SPARSE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        h = native_problem(p)
        assert h.describe().split()[0] == ("path=exact-f64" if tag == "v1" else "path=exact-f64v2")
        if p.n == 4 and tag in ("v2", "v2line"):
            assert ("mapping=patch8x8" if tag == "v2" else "mapping=line64") in h.describe()
        if tag.startswith(("v2sparse", "v2union")):
            A = int(np.prod([len(u) for u in p.u_levels]))
            want = env["PVI_SPARSE"] if (p.n == 4 and A <= 128) else "0"
            assert ("sparse=" + want) in h.describe(), h.describe()
        h.terminal_cost()
        stats, n = h.sweep(nsw, alpha, -1.0)
        outs[tag] = (h.get_J(), h.get_pi(), stats.copy())
        h.close()
        # a slab with halos (multi-GPU building block): rows [r0, r1) of axis 0
        N0 = p.dims[0]
        r0, r1 = N0 // 3, max(N0 // 3 + 2, (2 * N0) // 3)
        hs = native_problem(p, rows=(r0, r1), halo=(r0, N0 - r1))
        hs.terminal_cost()
        hs.sweep_async(alpha)
        hs.sweep_stats()
        outs[tag + "_slab"] = (hs.get_J(), hs.get_pi())
        hs.close()
    for k in ("PVI_NO_SWEEP64", "PVI_PATCH", "PVI_SPARSE"):
        monkeypatch.delenv(k, raising=False)
    for b in ("v2", "v2line", "v2sparse", "v2sparseline", "v2auto"):
        for a, c in (("v1", b), ("v1_slab", b + "_slab")):
            assert np.array_equal(outs[a][0], outs[c][0]) and np.array_equal(outs[a][1], outs[c][1]), (name, c)
        assert np.array_equal(outs["v1"][2], outs[b][2])


# ------------------------------------------------------------------------------------- full size (BASELINE configs[1])
def _c2():
    from pyro_amd import configs
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build("c2")
    s, g, cf = cfg["sys"], cfg["grid_sys"], cfg["cf"]
    dyn_id, params = s.device_dynamics()
    p = O.Problem(g.x_level, g.u_level, g.dt, dyn_id, np.array(params), cf.Q, cf.R, cf.S, cf.xbar, cf.ubar,
                  float(cf.INF), float(cf.EPS))
    return p


def test_full_size_c2_against_c_oracle():
    """1001 x 1001 x 51 (51.1 M cells per sweep): three sweeps, f32 within 1e-5 and f64 within 1e-12 of the
    oracle's C twin (itself bit-identical to the NumPy oracle)."""
    from oracle import c_oracle as CO
    p = _c2()
    c = CO.CProblem(p)
    J = c.terminal_cost()
    for _ in range(3):
        J, pi = c.sweep(J)
    from pyro_amd import _native
    nodes = np.arange(0, pi.size, 7, dtype=np.int64)
    outs = {}
    # look-up-table class and base class (PVI_FLAG_HARD_INF: a rejected cell costs exactly INF, dynamicprogramming.py:225-233)
    for dtype, tol, flags in (("float32", REL_F32, 0), ("float64", 1e-12, 0), ("float32", REL_F32, _native.FLAG_HARD_INF),
                              ("float64", 1e-12, _native.FLAG_HARD_INF)):
        h = native_problem(p, dtype=dtype, flags=flags)
        h.terminal_cost()
        stats, n = h.sweep(3, 1.0, -1.0)
        Jg, pig = h.get_J(), h.get_pi()
        assert relerr(Jg, J) <= tol
        if dtype == "float64":
            assert np.array_equal(pig, pi)
        else:   # float32: the float64 Q-regret of the GPU's action on the GPU's own previous J (no mismatch allowance)
            q, qmin = c.q_at(h.get_J(prev=True), nodes, pig[nodes].astype(np.int64))
            assert (q - qmin).max() <= 1e-5 * np.abs(J).max(), (q - qmin).max()
        assert abs(stats[-1, 0] - J.max()) <= 1e-5 * J.max()
        outs[(dtype, flags)] = (Jg, pig)
        h.close()
    # the pendulum's box IS its grid and every input is valid: a rejected cell lands outside the grid, where the interpolant
    # fills 0 -- the two classes must agree bit for bit
    for dtype in ("float32", "float64"):
        a, b = outs[(dtype, 0)], outs[(dtype, _native.FLAG_HARD_INF)]
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), dtype


def test_full_size_north_star_grid_against_c_oracle():
    """The north-star grid of BASELINE.json (C2': 201 x 201 nodes x 201 actions, swing-up demo bounds and weights): the
    float32 path there is the ACTION-SPLIT lean kernel (several lanes per node, cross-lane argmin with the first-minimum
    rule), float64 the second-form exact kernel.  Whole grid, 1, 3 and 200 sweeps against the oracle's C twin."""
    from oracle import c_oracle as CO
    from pyro_amd import configs
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build("c2p")
    s, g, cf = cfg["sys"], cfg["grid_sys"], cfg["cf"]
    dyn_id, params = s.device_dynamics()
    p = O.Problem(g.x_level, g.u_level, g.dt, dyn_id, np.array(params), cf.Q, cf.R, cf.S, cf.xbar, cf.ubar,
                  float(cf.INF), float(cf.EPS), x_lb=s.x_lb, x_ub=s.x_ub, u_lb=s.u_lb, u_ub=s.u_ub)
    c = CO.CProblem(p)
    ref, J = {}, c.terminal_cost()
    for k in range(1, 201):
        J, pi = c.sweep(J)
        if k in (1, 3, 200):
            ref[k] = (J.copy(), pi.copy())
    for dtype, tol in (("float32", REL_F32), ("float64", 1e-12)):
        h = native_problem(p, dtype=dtype)
        desc = h.describe()
        if dtype == "float32":
            assert desc.startswith("path=lean ") and "lsplit=0" not in desc, desc      # more than one lane per node
        else:
            assert desc.startswith("path=exact-f64v2"), desc
        h.terminal_cost()
        done = 0
        for k in (1, 3, 200):
            h.sweep(k - done, 1.0, -1.0)
            done = k
            Jg, pig = h.get_J(), h.get_pi()
            assert relerr(Jg, ref[k][0]) <= tol, (dtype, k)
            if dtype == "float64":
                assert np.array_equal(pig, ref[k][1]), k
            else:   # float32 rounding flips near-ties: float64 Q-regret of the GPU's action on the GPU's own previous J
                nodes = np.arange(pig.size, dtype=np.int64)
                q, qmin = c.q_at(h.get_J(prev=True), nodes, pig.astype(np.int64))
                assert (q - qmin).max() <= 1e-5 * max(1.0, np.abs(ref[k][0]).max()), (k, (q - qmin).max())
        h.close()


def test_full_size_c2_bellman_operator_properties():
    """Size-independent properties of the backup T at full size (f32 production path):
    monotone (J <= J' => TJ <= TJ'), constant shift (T(J+c) <= TJ + alpha*c, equality where the minimiser
    stays in bounds), and the device-side stop equals the host-side rule."""
    p = _c2()
    h = native_problem(p, dtype="float32")
    h.terminal_cost()
    h.sweep(40, 1.0, -1.0)
    J = h.get_J()
    h.sweep(1, 1.0, -1.0)
    TJ = h.get_J()
    c = 2.5
    h.set_J(J + c)
    h.sweep(1, 1.0, -1.0)
    TJc = h.get_J()
    assert (TJc >= TJ - 1e-3).all()                      # monotone
    assert (TJc <= TJ + c + 1e-3).all()                  # shift upper bound
    # in-bounds actions shift by exactly alpha*c (weights sum to 1), out-of-bounds ones stay at INF:
    # T(J+c) = min(TJ + c, INF) wherever INF is not already the minimum
    inside = TJ < float(p.INF) - c - 1.0
    assert inside.sum() > 100000
    assert np.abs(TJc[inside] - TJ[inside] - c).max() < 2e-3
    # stop rule: first sweep with delta <= tol ends the batch, later launches are no-ops
    h.set_J(J)
    stats, n = h.sweep(1000, 1.0, 0.3)
    assert 1 <= n < 1000 and stats[n - 1, 3] <= 0.3 and (n == 1 or stats[n - 2, 3] > 0.3)
    h.close()


def test_reference_side_stub_on_reference_tables():
    """INTEGRATION.md route B, executed: the binding a pyro maintainer would add
    (examples/reference_side_stub/dynamicprogramming_hip.py: its own pvi_desc mirror, pvi_create / pvi_set_tables /
    pvi_set_J / pvi_sweep / pvi_get_J / pvi_get_pi, nothing from pyro_amd) driven with the tables, J0 and results the
    REFERENCE produced (pendulum 21x21x5 and the obstacle case with its INF + alpha*J semantics)."""
    import importlib.util
    from pyro_amd import _native
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location(
        "dynamicprogramming_hip", os.path.join(root, "examples", "reference_side_stub", "dynamicprogramming_hip.py"))
    stub = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(stub)
    lib = stub.load_library(_native.LIB_PATH)
    for name, checks in (("pendulum_21x21x5", (1, 2, 10)), ("obstacles_21x21x3x3", (1, 5))):
        g = load(name)
        lv = O.make_levels(g["x_lb"], g["x_ub"], g["dims"])
        ul = O.make_levels(g["u_lb"], g["u_ub"], g["udims"])
        eng = stub.HipSweepEngine(lib, lv, ul, g["x_lb"], g["x_ub"], g["u_lb"], g["u_ub"], float(g["dt"]),
                                  g["x_next_table"], g["G"], float(g["INF"]))
        eng.set_J(g["J0"])
        k = 0
        for target in checks:
            stats, n = eng.sweep(target - k)
            k += n
            assert k == target
            np.testing.assert_allclose(eng.get_J(), g["J_%d" % target], rtol=1e-12, atol=1e-12)
            assert np.array_equal(eng.get_pi(), g["pi_%d" % target])
        eng.close()


# ------------------------------------------------------------------------------- full size, sampled (configs[2..4])
def _full_problem(name):
    from pyro_amd import configs
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build(name)
    s, g, cf = cfg["sys"], cfg["grid_sys"], cfg["cf"]
    dyn_id, params = s.device_dynamics()
    p = O.Problem(g.x_level, g.u_level, g.dt, dyn_id, np.array(params), cf.Q, cf.R, cf.S, cf.xbar, cf.ubar,
                  float(cf.INF), float(cf.EPS), x_lb=s.x_lb, x_ub=s.x_ub, u_lb=s.u_lb, u_ub=s.u_ub)
    return cfg, p


def _sample_blocks(dims, nrand=160, blen=1024, seed=7):
    """Node ranges for the sampled oracle check: the first and last rows of axis 0 (slab / halo edges), one whole
    velocity plane in the middle of the grid (every tile seam of the (i2, i3) tiling), and seeded blocks spread
    over the grid (position rows near and far from the box faces, in- and out-of-box frontier)."""
    N = int(np.prod(dims))
    vplane = int(dims[-1] * dims[-2])
    blocks = [(0, min(N, 3 * vplane)), (N - 3 * vplane, N)]
    mid = (dims[0] // 2) * int(np.prod(dims[1:])) + (dims[1] // 3) * vplane
    blocks.append((mid, mid + vplane))
    rng = np.random.default_rng(seed)
    for s0 in rng.integers(0, N - blen, nrand):
        blocks.append((int(s0), int(s0) + blen))
    return blocks


_FULL = {
    # name: (expected path, dma16, J tolerance, regret tolerance relative to max|J|)
    "c3": ("lean", 0, REL_F32, 1e-5),
    "c4": ("lean", 0, REL_F32, 1e-5),
    "c5": ("exact-f64v2", 0, 1e-12, 1e-6),
    # SURVEY 8(d)'s dense variant of configs[4]: dt = 0.01 keeps about half of the cells in the box, so the in-kernel
    # H(q)^-1 dynamics and the 16-corner interpolation run for most of them
    "c5d": ("exact-f64v2", 0, 1e-12, 1e-6),
}


@pytest.mark.parametrize("name", list(_FULL))
def test_full_size_sampled_against_c_oracle(name):
    """BASELINE configs[2], [3] (on one GPU) and [4] at FULL size: J_{k+1} = T J_k for k in {1, 3} and at depth (k = 200
    for the float32 cart-pole grids, 20 for the float64 two-link arm) is compared with
    the oracle's C twin on >= 2e5 sampled nodes, starting from the GPU's own J_k (so every production-size code
    path -- 16-byte window DMA, tuned tile shapes, XCD remap with ~1e6 tiles, int32 offsets at 5e8 nodes -- meets
    the oracle), pi by the float64 Q-regret of the GPU's action."""
    from oracle import c_oracle as CO
    path, dma16, tol, rtol = _FULL[name]
    cfg, p = _full_problem(name)
    g = cfg["grid_sys"]
    with contextlib.redirect_stdout(io.StringIO()):
        h = g._device_problem(cost=cfg["cf"].device_cost(), dtype=cfg["dtype"])
    desc = h.describe()
    fields = dict(kv.split("=", 1) for kv in desc.split(" note=")[0].split())
    assert fields["path"] == path, desc
    if name == "c5":        # 93 % of the two-link cells leave the box: set-up timing must pick the patch mapping
        assert fields["mapping"] == "patch8x8" and fields["off32"] == "1", desc
    if name == "c5d":       # ... and about half of them stay inside with the shorter step
        assert 0.35 < float(fields["inbox"]) < 0.75, desc
    if path == "lean":
        tv0, tv1 = (int(v) for v in fields["tile"].split("x"))
        assert int(fields["dma16"]) == dma16 and int(fields["lsplit"]) == 0 and int(fields["tb_tile"]) == 1, desc
        # the round-3 kernel: position-paired window, the cart-pole's displacement table spans (theta, dtheta) only
        # (axes 0 and 2 dropped: bits 0 and 2), g_x from per-axis terms, step-aligned row pieces, banded XCD schedule
        assert fields["win"] == "1" and fields["tables"] == "5" and fields["gx"] == "axes", desc
        assert 128 <= int(fields["block"]) <= 512 and tv0 >= 2 and tv1 >= 12 and int(fields["lds_bytes"]) <= 80 * 1024, desc
        assert int(fields["grid"].split("x")[0]) >= p.dims[0] * p.dims[1] * 6, desc        # tiles
    c = CO.CProblem(p)
    f32 = cfg["dtype"] == "float32"
    blocks = _sample_blocks(p.dims)
    assert sum(b - a for a, b in blocks) >= 200000
    h.terminal_cost()
    done = 0
    # k = 1, 3: a cost bowl with INF cliffs; then a DEVELOPED cost-to-go (VERDICT r4 weak #1 ii): depth 200 for the cart-pole
    # grids (the frontier of finite J has crossed the grid, the interpolant is smooth and steep in turns), 20 for the two-link
    # arm (float64, 15-40 ms per sweep) -- one backup of the GPU's own deep J against the C twin on the same sampled nodes
    deep = 200 if f32 else 20
    for k in (1, 3, deep):
        h.sweep(k - done, 1.0, -1.0)
        Jk = h.get_J()
        stats, _ = h.sweep(1, 1.0, -1.0)
        done = k + 1
        Jk1, pik1 = h.get_J(), h.get_pi()
        scale = max(np.abs(Jk1).max(), 1e-300)
        assert abs(stats[-1, 0] - Jk1.max()) <= 1e-6 * scale
        worst, nodes = 0.0, []
        for lo, hi in blocks:
            Jo, pio = c.sweep(Jk, 1.0, lo, hi, f32=f32)
            worst = max(worst, np.abs(Jk1[lo:hi] - Jo).max() / scale)
            nodes.append(np.arange(lo, hi))
        assert worst <= tol, (name, k, worst, desc)
        nodes = np.concatenate(nodes)
        q, qmin = c.q_at(Jk, nodes, pik1[nodes])
        assert (q - qmin).max() <= rtol * scale, (name, k, (q - qmin).max(), scale)
        del Jk, Jk1, pik1
    h.close()


def test_c3_full_size_solved_to_tolerance_f32_matches_f64():
    """north_star: "J* match to the CPU reference within 1e-5 relative".  BASELINE configs[2] at FULL size solved with
    solve_bellman_equation's stop rule (tol 0.1) twice on the GPU: the float32 production path and the float64 path
    (which the tests above pin to the oracle one backup at a time).  Same sweep count, and max |J32 - J64| / max |J64|
    <= 1e-5 at the end; the drift after every 100 sweeps is printed (alpha = 1 is only non-expansive: rounding errors
    are not contracted away, SURVEY 7 hard part 3)."""
    import bench
    from pyro_amd import configs
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build("c3")
    cv = bench.converged_check(cfg, tol=0.1, every=100)
    print("C3 solved to tol 0.1: %d sweeps (float32), %d (float64), max J %.3f, %.1f s"
          % (cv["sweeps_f32"], cv["sweeps_f64"], cv["max_J"], cv["seconds"]))
    for k, e in cv["drift_curve"]:
        print("  after %5d sweeps: max|J32 - J64| / max|J64| = %.3e" % (k, e))
    assert "path=lean" in cv["paths"]["float32"] and "path=exact-f64v2" in cv["paths"]["float64"], cv["paths"]
    assert cv["sweeps_f32"] == cv["sweeps_f64"], cv
    assert cv["rel_err"] <= REL_F32, cv["drift_curve"]                  # J*: the north-star claim (measured: 2.1e-6)
    # The iterates on the way differ by more (measured peak: 1.55e-5 around sweep 1500) and that is float32 STORAGE, not
    # the kernel's arithmetic: the oracle's C twin in float64 arithmetic with J rounded to float32 after every sweep
    # drifts the same way (profiles/r03_drift_f32_storage_cpu.log), and two different float32 kernels give the same
    # curve to three digits.  Increments below half an ulp of J (3e-5 at J = 1000) are lost while a node still ramps up;
    # the fixed point pulls them back.  A float32 + residual (compensated) cost-to-go would remove it (hard
    # part 3); J* meets the bound without, so the transient is bounded here as a regression guard, not at 1e-5.
    assert max(e for _, e in cv["drift_curve"]) <= 2.5e-5, cv["drift_curve"]


def test_c3_full_size_f32_error_feedback_stays_within_1e6_of_f64_at_every_checkpoint():
    """VERDICT r3 weak #1 / missing #6: the float32 iterates of BASELINE configs[2] drift to 1.55e-5 of max J mid-solve (plain
    float32 storage; the test above bounds that at 2.5e-5).  With error-feedback storage (PVI_FLAG_F32_FEEDBACK,
    f32_feedback=True: k_sweep_lean4fb) the whole solve stays within 1e-6 of the float64 iterates -- every checkpoint, not
    only J* (measured: 4.0e-7 worst over 1 915 sweeps) -- with the same stop sweep."""
    import bench
    from pyro_amd import configs
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build("c3")
    cv = bench.converged_check(cfg, tol=0.1, every=100, feedback=True)
    fb = cv["feedback"]
    for (k, e), (_, ep) in zip(fb["drift_curve"], cv["drift_curve"]):
        print("  after %5d sweeps: feedback %.3e   plain %.3e" % (k, e, ep))
    print("ms per sweep: feedback %.3f, plain %.3f, float64 %.3f" % (fb["ms_per_sweep"], fb["ms_per_sweep_plain"], fb["ms_per_sweep_f64"]))
    assert "feedback=1" in cv["paths"]["float32fb"] and "kernel=k_sweep_lean4fb<" in cv["paths"]["float32fb"], cv["paths"]
    assert "feedback=0" in cv["paths"]["float32"] and "kernel=k_sweep_lean4<" in cv["paths"]["float32"], cv["paths"]
    assert fb["sweeps"] == cv["sweeps_f64"], (fb["sweeps"], cv["sweeps_f64"])
    assert fb["max_transient_rel_err"] <= 1e-6, fb["drift_curve"]
    assert fb["rel_err"] <= 1e-6, fb["drift_curve"]


def test_f32_error_feedback_storage_small():
    """PVI_FLAG_F32_FEEDBACK on a small cart-pole: (i) the iterates stay within 6e-7 of the float64 ones at every checkpoint
    and end closer than plain float32 storage does; (ii) the policy passes the float64 Q-regret rule; (iii) a self check in
    the middle of a run is a dry run (residuals untouched: the same bits with and without it); (iv) a new terminal cost
    clears the residuals (the same bits as a fresh handle); (v) the sweep a handle cannot take it on refuses the flag."""
    from oracle import c_oracle as CO
    from pyro_amd import configs, _native
    from pyro_amd.planning import dynamicprogramming as DP
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build("cartpole:25,25,25,25:9:float32")
    g, cf = cfg["grid_sys"], cfg["cf"]

    def make(dt, fb=False):
        with contextlib.redirect_stdout(io.StringIO()):
            dp = DP.DynamicProgrammingWithLookUpTable(g, cf, dtype=dt, f32_feedback=fb)
        dp.save_time_history = False
        dp.verbose = False
        return dp
    d64, d32, dfb, dfb2 = make("float64"), make("float32"), make("float32", True), make("float32", True)
    assert "feedback=1" in dfb._p.describe() and "feedback=0" in d32._p.describe()
    worst_fb = worst_plain = 0.0
    for k in range(4):
        for dp in (d64, d32, dfb):
            dp._p.sweep(100, 1.0, -1.0)
        dfb2._p.sweep(37, 1.0, -1.0)
        rel, mism = dfb2._p.self_check(1.0)                  # (iii) a dry run between two batches of other sizes
        assert rel <= 1e-5
        dfb2._p.sweep(63, 1.0, -1.0)
        J64 = d64._p.get_J()
        m = np.abs(J64).max()
        e_fb, e_plain = np.abs(dfb._p.get_J() - J64).max() / m, np.abs(d32._p.get_J() - J64).max() / m
        worst_fb, worst_plain = max(worst_fb, e_fb), max(worst_plain, e_plain)
        print("after %d sweeps: feedback %.3e plain %.3e" % (100 * (k + 1), e_fb, e_plain))
        assert e_fb <= 6e-7, (k, e_fb)
    assert "kernel=k_sweep_lean4fb<" in dfb._p.describe(), dfb._p.describe()
    assert worst_fb < worst_plain, (worst_fb, worst_plain)
    assert np.array_equal(dfb._p.get_J(), dfb2._p.get_J()) and np.array_equal(dfb._p.get_pi(), dfb2._p.get_pi())
    # (ii) the policy against the float64 twin of the oracle on the float64 J of the sweep before
    import bench
    c = CO.CProblem(bench.oracle_problem(cfg))
    Jprev = d64._p.get_J(prev=True)
    nodes = np.arange(0, g.nodes_n, 7, dtype=np.int64)
    q, qmin = c.q_at(Jprev, nodes, dfb._p.get_pi()[nodes])
    assert (q - qmin).max() <= 1e-5 * np.abs(Jprev).max(), (q - qmin).max()
    # (iv) restart: terminal cost again, 50 sweeps == a fresh handle's 50 sweeps
    dfb.evaluate_terminal_cost()
    dfb._p.sweep(50, 1.0, -1.0)
    fresh = make("float32", True)
    fresh._p.sweep(50, 1.0, -1.0)
    assert np.array_equal(dfb._p.get_J(), fresh._p.get_J())
    # ... and so does a cost-to-go set from the host
    J0 = fresh._p.get_J()
    dfb._p.set_J(J0)
    d32b = make("float32", True)
    d32b._p.set_J(J0)
    dfb._p.sweep(20, 1.0, -1.0)
    d32b._p.sweep(20, 1.0, -1.0)
    assert np.array_equal(dfb._p.get_J(), d32b._p.get_J())
    # (v) refusals: float64, a 2-D grid (class surface), and the library itself on a handle without the 4-D window sweep
    with pytest.raises(NotImplementedError):
        make("float64", True)
    with contextlib.redirect_stdout(io.StringIO()):
        c2 = configs.build("pendulum:41,41:5:float32")
    with pytest.raises(NotImplementedError):
        DP.DynamicProgrammingWithLookUpTable(c2["grid_sys"], c2["cf"], dtype="float32", f32_feedback=True)
    with pytest.raises(_native.NativeError) as ei:
        c2["grid_sys"]._device_problem(cost=DP.device_cost_of(c2["cf"], c2["sys"]), dtype="float32", flags=_native.FLAG_F32_FEEDBACK)
    assert "PVI_FLAG_F32_FEEDBACK" in str(ei.value)


_WORLD1 = r"""
import contextlib, io, sys
import numpy as np
import torch                      # torch first: it brings its own HIP runtime, libpyrovi then shares it
import torch.distributed as dist
sys.path.insert(0, %r)
from pyro_amd import configs, parallel
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1)
with contextlib.redirect_stdout(io.StringIO()):
    cfg = configs.build("cartpole:21,21,21,21:7:float32")
vi = parallel.ShardedValueIteration(cfg["grid_sys"], cfg["cf"], dist, dtype="float32")
for _ in range(5):
    st = vi.sweep(1.0)
J, pi = vi.gather()
from pyro_amd.planning import dynamicprogramming
with contextlib.redirect_stdout(io.StringIO()):
    dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"], dtype="float32")
stats, _ = dp._p.sweep(5, 1.0, -1.0)
assert np.array_equal(J, dp._p.get_J()) and np.array_equal(pi, dp._p.get_pi())
assert np.allclose(st, stats[-1], rtol=1e-12)
np.save(%r, J)
dist.destroy_process_group()
print("WORLD1-OK")
"""


def test_sharded_driver_world1_on_gpu(tmp_path):
    """pyro_amd.parallel with the product HipSlab (torch buffers handed to the library as ext_J / ext_pi, kernels
    on torch's stream) at world size 1; checked against the reference golden as well.  Runs in a fresh
    interpreter because torch must initialise its bundled HIP runtime before libpyrovi is loaded."""
    import subprocess
    import sys
    from conftest import ROOT
    out = str(tmp_path / "J.npy")
    r = subprocess.run([sys.executable, "-c", _WORLD1 % (ROOT, out)], capture_output=True, text=True, timeout=900)
    assert "WORLD1-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    g = load("cartpole_21p4x7")
    p = oracle_problem(g, O.DYN_CARTPOLE, O.cartpole_consts())
    J = O.terminal_cost(p)
    for _ in range(5):
        J, _ = O.sweep(p, J)
    assert relerr(np.load(out), J) <= REL_F32


# ------------------------------------------------------------------------------------- tier B on the reference's tables
@pytest.mark.parametrize("name,sweeps", [("obstacles_21x21x3x3", 5), ("helicopter_11x11x11x5", 5), ("reachability_41x41x3", 20)])
def test_table_tier_lut_and_base_semantics(name, sweeps):
    """Obstacles (2-D, m=2), domain-check cost (3-D) and reachability cost through pvi_set_tables: the
    look-up-table class semantics (ok = NULL) and the base-class semantics (ok mask) against the reference."""
    from pyro_amd import _native
    g = load(name)
    lv = O.make_levels(g["x_lb"], g["x_ub"], g["dims"])
    ul = O.make_levels(g["u_lb"], g["u_ub"], g["udims"])
    alpha = 0.999 if "helicopter" in name else 1.0
    ok = g["x_next_isok"] & g["action_isok"]
    for mask, key, pkey in ((None, "J_%d", "pi_%d"), (ok, "Jbase_%d", "pibase_%d")):
        h = _native.Problem(lv, ul, g["x_lb"], g["x_ub"], g["u_lb"], g["u_ub"], float(g["dt"]),
                            dynamics_id=_native.DYN_TABLE, table_inf=float(g["INF"]))
        h.set_tables(g["x_next_table"], g["G"], mask)
        h.set_J(g["J0"])
        for k in range(1, sweeps + 1):
            h.sweep(1, alpha, -1.0)
            if k in (1, sweeps):
                assert relerr(h.get_J(), g[key % k]) < 1e-13
                assert np.array_equal(h.get_pi(), g[pkey % k])
        h.close()
        # float32 handle: the tables are packed into (offset, fractions, G) records; same semantics within 1e-5
        h = _native.Problem(lv, ul, g["x_lb"], g["x_ub"], g["u_lb"], g["u_ub"], float(g["dt"]), dtype="float32",
                            dynamics_id=_native.DYN_TABLE, table_inf=float(g["INF"]))
        h.set_tables(g["x_next_table"], g["G"], mask)
        assert "path=table-packed" in h.describe()
        h.set_J(g["J0"])
        h.sweep(sweeps, alpha, -1.0)
        assert relerr(h.get_J(), g[key % sweeps]) < REL_F32
        if "reachability" not in name:        # (0/1 costs: exact ties between actions, broken by float32 rounding)
            assert (h.get_pi() != g[pkey % sweeps]).mean() < 0.02
        h.close()


def test_policy_evaluator_classes():
    """PolicyEvaluatorWithLookUpTable driven like examples/.../policy_evaluator_with_computed_torque.py, with a
    stand-in controller that replays the reference controller's inputs (U of the golden)."""
    from pyro_amd.analysis import costfunction
    from pyro_amd.control import controller
    from pyro_amd.dynamic import pendulum
    from pyro_amd.planning import discretizer, dynamicprogramming
    g = load("policy_eval_41x41")

    with contextlib.redirect_stdout(io.StringIO()):
        s = pendulum.SinglePendulum()
        s.x_ub, s.x_lb = g["x_ub"].copy(), g["x_lb"].copy()
        s.u_ub, s.u_lb = g["u_ub"].copy(), g["u_lb"].copy()
        grid = discretizer.GridDynamicSystem(s, [41, 41], [11], 0.05, False)

        class Replay(controller.StaticController):
            def __init__(self):
                super().__init__(1, 1, 2)
                self.rbar = np.array([-3.14])

            def c(self, y, r, t=0):
                return g["U"][grid.get_nearest_node_id_from_state(y)]

        q = costfunction.QuadraticCostFunction.from_sys(s)
        q.xbar, q.INF = g["xbar"].copy(), float(g["INF"])
        q.S = g["S"].copy()
        ev = dynamicprogramming.PolicyEvaluatorWithLookUpTable(Replay(), grid, q)
        ev.save_time_history = False
        np.testing.assert_allclose(ev.x_next_table, g["x_next_table"], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(ev.G, g["G"], rtol=1e-13, atol=1e-13)
        ev.compute_steps(10)
    assert relerr(ev.J, g["J_10"]) < 1e-12
    assert (ev.pi == 0).all()


def test_policy_evaluator_tables_are_built_on_the_gpu():
    """examples/demos_by_tool/dynamicprogramming/policy_evaluator_with_computed_torque.py without its O(N)
    Python loop.  With the reference's ComputedTorqueController (pyro/control/nonlinear.py:23-116, mirrored in
    pyro_amd.control.nonlinear) on a closed-form system the control law, f, the validity tests and g all run in one kernel
    (pvi_policy_tables); the inputs equal the reference controller's bit for bit (golden U), the tables its x_next_table / G,
    J after 10 sweeps its J.  Any other controller: ctl.c on the host, the rest on the GPU; a system without a closed form:
    the reference's loop."""
    from pyro_amd.analysis import costfunction
    from pyro_amd.control import controller, nonlinear
    from pyro_amd.dynamic import manipulator, mountaincar, pendulum
    from pyro_amd.planning import discretizer, dynamicprogramming
    g = load("policy_eval_41x41")
    with contextlib.redirect_stdout(io.StringIO()):
        s = pendulum.SinglePendulum()
        s.x_ub, s.x_lb = g["x_ub"].copy(), g["x_lb"].copy()
        s.u_ub, s.u_lb = g["u_ub"].copy(), g["u_lb"].copy()
        grid = discretizer.GridDynamicSystem(s, [41, 41], [11], 0.05, False)
        q = costfunction.QuadraticCostFunction.from_sys(s)
        q.xbar, q.INF = g["xbar"].copy(), float(g["INF"])
        q.S = g["S"].copy()
        ctl = nonlinear.ComputedTorqueController(s)
        ctl.rbar = np.array([-3.14])
        ev = dynamicprogramming.PolicyEvaluatorWithLookUpTable(ctl, grid, q)
        ev.save_time_history = False
    assert ev.tables_on == "gpu: controller + dynamics + cost"
    assert np.array_equal(ev.U, g["U"])                                     # the reference controller's inputs, bit for bit
    np.testing.assert_allclose(ev.x_next_table, g["x_next_table"], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(ev.G, g["G"], rtol=1e-13, atol=1e-13)
    assert np.array_equal(ev.J, g["J0"])
    with contextlib.redirect_stdout(io.StringIO()):
        ev.compute_steps(10)
    assert relerr(ev.J, g["J_10"]) < 1e-12 and (ev.pi == 0).all()

    # an arbitrary Python controller on the same system: inputs on the host, tables on the GPU, same tables
    class Wrapped(controller.StaticController):
        def __init__(self):
            super().__init__(1, 1, 2)
            self.rbar = np.array([-3.14])

        def c(self, y, r, t=0):
            return ctl.c(y, r, t)
    with contextlib.redirect_stdout(io.StringIO()):
        ev2 = dynamicprogramming.PolicyEvaluatorWithLookUpTable(Wrapped(), grid, q)
    assert ev2.tables_on == "gpu: dynamics + cost (controller on the host)"
    assert np.array_equal(ev2.x_next_table, ev.x_next_table) and np.array_equal(ev2.G, ev.G)

    # the two-link arm (dof = m = 2): kernel against the mirror of the reference's controller + its own loop
    with contextlib.redirect_stdout(io.StringIO()):
        s2 = manipulator.TwoLinkManipulator()
        s2.u_ub, s2.u_lb = np.array([400.0, 400.0]), np.array([-400.0, -400.0])
        g2 = discretizer.GridDynamicSystem(s2, [7, 7, 7, 7], [3, 3], 0.01)
        q2 = costfunction.QuadraticCostFunction.from_sys(s2)
        c2 = nonlinear.ComputedTorqueController(s2)
        c2.rbar = np.array([0.3, -0.2]); c2.w0, c2.zeta = 2, 0.9
        ev3 = dynamicprogramming.PolicyEvaluator(c2, g2, q2)
    assert ev3.tables_on == "gpu: controller + dynamics + cost"
    X = g2.state_from_node_id
    Uh = np.array([c2.c(X[i], c2.rbar, 0) for i in range(g2.nodes_n)])
    assert np.abs(ev3.U - Uh).max() <= 1e-12 * np.abs(Uh).max()
    Xn = np.array([s2.f(X[i], Uh[i]) * g2.dt + X[i] for i in range(g2.nodes_n)])
    assert np.abs(ev3.x_next_table - Xn).max() <= 1e-10 * np.abs(Xn).max()

    # no closed form (generic mechanical tier): the reference's loop, as before
    with contextlib.redirect_stdout(io.StringIO()):
        s4 = mountaincar.MountainCar()
        g4 = discretizer.GridDynamicSystem(s4, [21, 21], [3])
        ev4 = dynamicprogramming.PolicyEvaluator(Wrapped(), g4, costfunction.QuadraticCostFunction.from_sys(s4))
    assert ev4.tables_on == "host"


# ------------------------------------------------------------------------------------- edge cases
def _custom_problem(kind, dims, udims, dt=0.05, alpha=1.0, seed=0, fancy_cost=False, INF=500.0, EPS=1e-3, bounds=None):
    rng = np.random.default_rng(seed)
    if kind == "pendulum":
        dyn, c, n, m = O.DYN_PENDULUM, O.pendulum_consts(d1=0.3), 2, 1
        lb, ub, ulb, uub = [-2 * np.pi] * 2, [2 * np.pi] * 2, [-5.0], [5.0]
    elif kind == "cartpole":
        dyn, c, n, m = O.DYN_CARTPOLE, O.cartpole_consts(), 4, 1
        lb, ub, ulb, uub = [-3.0, -np.pi, -4.0, -6.0], [2.0, 2 * np.pi, 5.0, 7.0], [-8.0], [10.0]
    else:
        dyn, c, n, m = O.DYN_TWOLINK, O.twolink_consts(**DOUBLEP), 4, 2
        lb, ub, ulb, uub = [-5.0, -1.5, -4.0, -4.0], [0.5, 4.0, 5.5, 7.0], [-12.0, -10.0], [12.0, 11.0]
    if bounds is not None:
        lb, ub = bounds
    lv = O.make_levels(lb, ub, dims)
    ul = O.make_levels(ulb, uub, udims)
    Q, R, S = np.eye(n), np.eye(m), np.zeros((n, n))
    xbar, ubar = np.zeros(n), np.zeros(m)
    if fancy_cost:
        A = rng.normal(size=(n, n)); Q = A @ A.T / n
        B = rng.normal(size=(m, m)); R = B @ B.T + 0.1 * np.eye(m)
        C = rng.normal(size=(n, n)); S = 3.0 * C @ C.T / n
        xbar = 0.3 * rng.normal(size=n); ubar = 0.5 * rng.normal(size=m)
    return O.Problem(lv, ul, dt, dyn, c, Q, R, S, xbar, ubar, INF, EPS), alpha


EDGE_CASES = {
    "A300_u16_policy": dict(kind="pendulum", dims=(64, 37), udims=(300,)),
    "minimal_2x2_A1": dict(kind="pendulum", dims=(2, 2), udims=(1,)),
    "cartpole_ragged_fancy_cost": dict(kind="cartpole", dims=(9, 8, 7, 6), udims=(70,), alpha=0.9, fancy_cost=True, EPS=0.8),
    "doublependulum_72_actions": dict(kind="twolink", dims=(8, 8, 8, 8), udims=(9, 8), dt=0.1, alpha=0.97),
    "everything_out_of_bounds": dict(kind="pendulum", dims=(15, 15), udims=(5,), dt=50.0),
    "tiny_box": dict(kind="pendulum", dims=(33, 65), udims=(7,), bounds=([-0.2, -0.1], [0.3, 0.4])),
}


@pytest.mark.parametrize("name", list(EDGE_CASES))
def test_edge_cases_f64_exact_and_f32_within_tolerance(name):
    kw = dict(EDGE_CASES[name])
    p, alpha = _custom_problem(**kw)
    J = O.terminal_cost(p)
    Jprev = J
    for _ in range(4):
        Jprev = J
        J, pi = O.sweep(p, J, alpha)
    for dtype, tol in (("float64", 1e-12), ("float32", REL_F32)):
        h = native_problem(p, dtype=dtype)
        h.terminal_cost()
        stats, n = h.sweep(4, alpha, -1.0)
        Jg, pig = h.get_J(), h.get_pi()
        assert relerr(Jg, J) <= tol, (name, dtype, h.describe())
        bad = pig != pi
        if dtype == "float64":
            assert not bad.any() or O.q_regret(p, Jprev, pig, alpha)[bad].max() < 1e-9
        else:
            assert O.q_regret(p, h.get_J(prev=True), pig, alpha).max() <= 1e-5 * max(1.0, np.abs(J).max())
        assert pig.min() >= 0 and pig.max() < p.actions_n
        h.close()
    if name == "everything_out_of_bounds":      # dt = 50 s: only nodes at rest with zero net torque stay inside
        assert (J == p.INF).mean() > 0.9 and (pi[J == p.INF] == 0).all()


def test_c2_solve_to_convergence_f32_vs_f64():
    """BASELINE configs[1] solved to tol 0.1 in both precisions: same sweep count, J* within 1e-5."""
    p = _c2()
    out = {}
    for dtype in ("float64", "float32"):
        h = native_problem(p, dtype=dtype)
        h.terminal_cost()
        stats, n = h.sweep(20000, 1.0, 0.1)
        out[dtype] = (h.get_J(), n)
        h.close()
    assert out["float64"][1] == out["float32"][1] and 100 < out["float64"][1] < 20000
    assert relerr(out["float32"][0], out["float64"][0]) <= REL_F32


# ------------------------------------------------------------------------------------- batched rollouts (next #2)
def test_batched_rollouts_match_reference():
    """pvi_rollout: LookUpTableController interpolation + f + Euler for a batch of initial states, against the
    reference's closed loop (golden) and through the class surface (policy edited on the host, re-uploaded)."""
    from pyro_amd.analysis import costfunction
    from pyro_amd.dynamic import pendulum
    from pyro_amd.planning import discretizer, dynamicprogramming
    g = load("rollout_pendulum_21x21x5")
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
    t, X, U = dp.simulate_closed_loop(g["X0"], tf=float(g["tf"]), n=int(g["npts"]))
    np.testing.assert_allclose(U, g["U"], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(X, g["X"], rtol=1e-8, atol=1e-8)
    # a larger batch: end states only
    rng = np.random.default_rng(0)
    X0 = rng.uniform(s.x_lb, s.x_ub, size=(4096, 2))
    Xe = dp._p.rollout(X0, 121, 0.025, trajectory=False)
    p = oracle_problem(g, O.DYN_PENDULUM, O.pendulum_consts())
    Xo, _ = O.rollout(p, dp.pi, X0[:64], 121, 3.0)
    np.testing.assert_allclose(Xe[:64], Xo[:, -1], rtol=1e-7, atol=1e-7)


def test_class_surface_host_edits_and_three_step_api(tmp_path):
    """dp.J assigned on the host is uploaded before the next sweep; the initialize/compute/finalize triple and
    save_latest / load_J_next behave like the reference (dynamicprogramming.py:175-261, :481-499)."""
    from pyro_amd.analysis import costfunction
    from pyro_amd.dynamic import pendulum
    from pyro_amd.planning import discretizer, dynamicprogramming
    g = load("pendulum_21x21x5")
    p = oracle_problem(g, O.DYN_PENDULUM, O.pendulum_consts())
    with contextlib.redirect_stdout(io.StringIO()):
        s = pendulum.SinglePendulum()
        grid = discretizer.GridDynamicSystem(s, [21, 21], [5])
        q = costfunction.QuadraticCostFunction.from_sys(s)
        q.xbar, q.INF = np.array([-3.14, 0.0]), 300
        dp = dynamicprogramming.DynamicProgramming(grid, q)        # base class: same recursion for box validity
        rng = np.random.default_rng(3)
        J0 = rng.uniform(0, 50, grid.nodes_n)
        dp.J = J0
        dp.alpha = 0.9
        dp.initialize_backward_step()
        dp.compute_backward_step()
        delta = dp.finalize_backward_step()
    Jn, pi = O.sweep(p, J0, 0.9)
    assert relerr(dp.J, Jn) < 1e-13 and np.array_equal(dp.pi, pi)
    assert abs(delta - O.sweep_stats(Jn, J0)[1]) < 1e-9
    assert relerr(dp.J_next, J0) < 1e-15 and dp.k == 1 and abs(dp.t + 0.05) < 1e-12
    name = str(tmp_path / "vi")
    dp.save_latest(name)
    assert np.array_equal(np.load(name + "_J_inf.npy"), dp.J_next)             # J_next, not J (:484)
    assert np.array_equal(np.load(name + "_pi_inf.npy"), dp.pi)
    dp.load_J_next(name)
    assert relerr(dp.J_next, J0) < 1e-15
    with pytest.raises(ValueError):
        dp.J = np.zeros(7)


_WORLD2 = r"""
import contextlib, io, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %r)
from pyro_amd import configs, parallel
rank, out = int(sys.argv[1]), sys.argv[2]
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d", rank=rank, world_size=%d)
with contextlib.redirect_stdout(io.StringIO()):
    cfg = configs.build("%s")
vi = parallel.ShardedValueIteration(cfg["grid_sys"], cfg["cf"], dist, dtype="float32", device=0, overlap=%s)
stats = [vi.sweep(1.0) for _ in range(4)]
last = vi.run(3, 1.0, -1.0)
J, pi = vi.gather()
if rank == 0:
    np.savez(out, J=J, pi=pi, stats=np.array(stats + [last]), p2p=vi.p2p, halo=vi.halo)
np.save(out + ".pieces%%d.npy" %% rank, len(vi.slab.handles))
dist.destroy_process_group()
print("WORLD2-OK", rank)
"""


@pytest.mark.parametrize("case,world,overlap", [("cartpole:21,21,21,21:7:float32", 2, True),
                                                ("cartpole:21,21,21,21:7:float32", 2, False),
                                                ("pendulum:101,101:11:float32", 2, True),
                                                ("cartpole:21,21,21,21:7:float32", 3, True)])
def test_two_ranks_share_one_gpu(tmp_path, case, world, overlap):
    """The sharded driver with the product HipSlab on real hardware: two processes (both on GPU 0), halo rows moved
    by a gloo process group through host staging, statistics all-reduced -- must equal the single-handle result
    bit for bit (same kernels, same arithmetic per node).  overlap=True is the boundary-first schedule: separate
    handles for the rows next to the neighbour and for the interior over the same buffers, the exchange on a second
    stream while the interior kernel runs.  RCCL itself is the only part not exercised here."""
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "res.npz")
    code = _WORLD2 % (ROOT, port, world, case, overlap)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    logs = [p.communicate(timeout=900)[0] for p in procs]
    assert all("WORLD2-OK" in l for l in logs), "\n".join(l[-1500:] for l in logs)
    r = np.load(out)
    from pyro_amd import configs
    from pyro_amd.planning import dynamicprogramming
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build(case)
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"], dtype="float32")
    stats, _ = dp._p.sweep(7, 1.0, -1.0)
    assert np.array_equal(r["J"], dp._p.get_J()) and np.array_equal(r["pi"], dp._p.get_pi())
    np.testing.assert_allclose(r["stats"][:4], stats[:4], rtol=1e-12)
    np.testing.assert_allclose(r["stats"][4], stats[6], rtol=1e-12)
    assert bool(r["p2p"])
    pieces = [int(np.load(out + ".pieces%d.npy" % k)) for k in range(world)]
    assert pieces == ([1] * world if not overlap else [2] + [3] * (world - 2) + [2])


_BASE2 = r"""
import contextlib, io, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %r)
from pyro_amd import parallel
from pyro_amd.analysis import costfunction
from pyro_amd.dynamic import vehicle_steering
from pyro_amd.planning import discretizer, dynamicprogramming
rank, out = int(sys.argv[1]), sys.argv[2]
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d", rank=rank, world_size=2)
with contextlib.redirect_stdout(io.StringIO()):
    s = vehicle_steering.HolonomicMobileRobotwithObstacles()
    g = discretizer.GridDynamicSystem(s, [41, 41], [3, 3])
    cf = costfunction.QuadraticCostFunction.from_sys(s)
    cf.INF = 100.0
    dp = dynamicprogramming.DynamicProgramming(g, cf, dtype="float64", device=0, comm=parallel.TorchDistComm(dist))
    dp.save_time_history = False
    assert dp.sharded and dp.HARD_INF
    dp.compute_steps(6)
    J, pi = dp.J.copy(), dp.pi.copy()
if rank == 0:
    np.savez(out, J=J, pi=pi)
dist.destroy_process_group()
print("BASE2-OK", rank)
"""


@pytest.mark.gpu
def test_base_class_with_obstacles_over_torch_dist_comm(tmp_path):
    """the base class (a rejected cell costs exactly INF, dynamicprogramming.py:225-233) over the
    Python-driven sharded schedule used to run the look-up-table recursion (INF + alpha J) -- HipSlab never received
    PVI_FLAG_HARD_INF.  Two ranks on GPU 0 (gloo, host-staged halo) on a system with obstacles must give the single-GPU
    base class bit for bit, which differs from the look-up-table class."""
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    from pyro_amd.analysis import costfunction
    from pyro_amd.dynamic import vehicle_steering
    from pyro_amd.planning import discretizer, dynamicprogramming
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    out = str(tmp_path / "base.npz")
    code = _BASE2 % (ROOT, port)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    logs = [p.communicate(timeout=900)[0] for p in procs]
    assert all("BASE2-OK" in l for l in logs), "\n".join(l[-1500:] for l in logs)
    r = np.load(out)
    with contextlib.redirect_stdout(io.StringIO()):
        s = vehicle_steering.HolonomicMobileRobotwithObstacles()
        g = discretizer.GridDynamicSystem(s, [41, 41], [3, 3])
        cf = costfunction.QuadraticCostFunction.from_sys(s)
        cf.INF = 100.0
        one = dynamicprogramming.DynamicProgramming(g, cf, dtype="float64")
        one.save_time_history = False
        one.compute_steps(6)
        lut = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, cf, dtype="float64")
        lut.save_time_history = False
        lut.compute_steps(6)
    assert np.array_equal(r["J"], one.J) and np.array_equal(r["pi"], one.pi)
    assert not np.array_equal(one.J, lut.J)                     # the two recursions really differ on this system


@pytest.mark.gpu
@pytest.mark.parametrize("name,tol,extra", [("config1_pendulum_101x101x11", 0.1, {}),
                                            ("pendulum_demo_51x51x9", -1.0, {}),
                                            ("pendulum_lowdef_41x21x3", -1.0, {}),      # a zone of on-target nodes (EPS = 1)
                                            ("pendulum_21x21x5", -1.0, {}),
                                            ("cartpole_11p4x5", -1.0, {"PVI_PATCH": "0", "PVI_SPARSE": "0"})])
def test_multi_sweep_launch_is_bit_identical(name, tol, extra, variants):
    """a batch of sweeps as ONE cooperative launch (k_sweep64m: per-node state in registers, ping-pong J, grid
    barrier, device-side stop) against one launch per sweep (MULTI=0): J, pi, every sweep's statistics and the stop sweep
    are the same bits -- BASELINE config 1 stops after the reference's 618 sweeps either way."""
    g = load(name)
    case = CASES.get(name, (O.DYN_PENDULUM, O.pendulum_consts()))
    p = oracle_problem(g, *case)
    alpha = float(g["alpha"]) if "alpha" in g.files else 1.0
    outs = {}
    # "multi": on 2-D grids with few actions the per-action cells stay in registers and J travels write-through (REGTAB);
    # "multi_fenced": the same launch recomputing them every sweep, with the fence-based barrier
    for tag, env in (("multi", {}), ("multi_fenced", {"PVI_REGTAB": "0"}), ("single", {"PVI_MULTI": "0"})):
        for k in ("PVI_MULTI", "PVI_PATCH", "PVI_SPARSE", "PVI_REGTAB"):
            variants.delenv(k)
        for k, v in {**extra, **env}.items():
            variants.setenv(k, v)
        h = native_problem(p, dtype="float64")
        h.terminal_cost()
        stats, n = h.sweep(700 if tol > 0 else 9, alpha, tol)
        outs[tag] = (h.get_J(), h.get_pi(), stats, n, h.describe(), h.get_J(prev=True))
        h.close()
    assert "multi=1" in outs["multi"][4] and "kernel=k_sweep64m<" in outs["multi"][4], outs["multi"][4]
    assert "multi=0" in outs["single"][4] and "kernel=k_sweep64<" in outs["single"][4], outs["single"][4]
    two_d = len(p.levels) == 2
    assert ("regtab=1" if two_d else "regtab=0") in outs["multi"][4], outs["multi"][4]
    if two_d:   # the stop test rides one sweep behind in the register-table form: every stop position, also inside short batches
        for k in ("PVI_MULTI", "PVI_PATCH", "PVI_SPARSE", "PVI_REGTAB"):
            variants.delenv(k)
        for tol_k in (1e30, 60.0, 20.0, 5.0, 2.0, 0.7):
            for cap in (1, 2, 3, 40):
                res = []
                for env in ({}, {"PVI_MULTI": "0"}):
                    variants.delenv("PVI_MULTI")
                    for k, v in env.items():
                        variants.setenv(k, v)
                    h = native_problem(p, dtype="float64")
                    h.terminal_cost()
                    st, n = h.sweep(cap, alpha, tol_k)
                    st2, n2 = h.sweep(cap, alpha, tol_k)          # a second batch on the same handle
                    res.append((n, n2, h.get_J(), h.get_pi(), np.array(st), np.array(st2)))
                    h.close()
                variants.delenv("PVI_MULTI")
                assert res[0][0] == res[1][0] and res[0][1] == res[1][1], (tol_k, cap, res[0][:2], res[1][:2])
                for i in (2, 3, 4, 5):
                    assert np.array_equal(res[0][i], res[1][i]), (tol_k, cap, i)
    if two_d:   # the write-through hand-off between the sweeps, again and again, with a second handle sweeping beside it
        for k in ("PVI_MULTI", "PVI_PATCH", "PVI_SPARSE", "PVI_REGTAB"):
            variants.delenv(k)
        noise = native_problem(p, dtype="float32")
        noise.terminal_cost()
        for rep in range(6):
            h = native_problem(p, dtype="float64")
            h.terminal_cost()
            for _ in range(100):                     # (its own stream: small kernels coming and going on other CUs)
                noise.sweep_async(alpha)
            stats, n = h.sweep(700 if tol > 0 else 9, alpha, tol)
            assert n == outs["single"][3] and np.array_equal(h.get_J(), outs["single"][0]) and np.array_equal(stats, outs["single"][2]), rep
            h.close()
        noise.synchronize()
        noise.close()
    assert "multi=1" in outs["multi_fenced"][4] and "regtab=0" in outs["multi_fenced"][4], outs["multi_fenced"][4]
    for tag in ("multi", "multi_fenced"):
        assert outs[tag][3] == outs["single"][3] == (618 if tol > 0 else 9), tag
        for i in (0, 1, 5):
            assert np.array_equal(outs[tag][i], outs["single"][i]), (tag, i)
        assert np.array_equal(outs[tag][2], outs["single"][2]), tag          # the statistics of every sweep


@pytest.mark.gpu
@pytest.mark.parametrize("dims,udim", [((201, 201), 21), ((101, 101), 21), ((57, 301), 24), ((333, 45), 13), ((181, 181), 7)])
def test_wide_multi_sweep_launch_is_bit_identical(dims, udim, variants):
    """The reference's usual demo sizes in float64 (201 x 201 x 21: 158 workgroups): the WIDE register-table form of the multi-sweep
    launch (up to 24 actions, 512 workgroups) against one launch per sweep -- J, pi, the statistics of every sweep, the stop."""
    from pyro_amd.analysis import costfunction
    from pyro_amd.dynamic import pendulum
    from pyro_amd.planning import discretizer, dynamicprogramming
    outs = {}
    with contextlib.redirect_stdout(io.StringIO()):
        s = pendulum.SinglePendulum()
        grid = discretizer.GridDynamicSystem(s, list(dims), [udim])
        cf = costfunction.QuadraticCostFunction.from_sys(s)
        cf.xbar, cf.INF = np.array([-3.14, 0.0]), 300.0
        for tag, env in (("multi", {}), ("single", {"PVI_MULTI": "0"})):
            variants.delenv("PVI_MULTI")
            for k, v in env.items():
                variants.setenv(k, v)
            dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(grid, cf, dtype="float64")
            dp.save_time_history = False
            st1, n1 = dp._p.sweep(37, 1.0, -1.0)
            st2, n2 = dp._p.sweep(500, 1.0, 2.0)           # a stop inside the batch
            outs[tag] = (dp._p.get_J(), dp._p.get_pi(), np.array(st1), np.array(st2), n1, n2, dp._p.describe())
            dp._p.close()
        variants.delenv("PVI_MULTI")
    assert "multi=1" in outs["multi"][6] and "regtab=1" in outs["multi"][6] and "kernel=k_sweep64m<1,unsignedchar,2>" in outs["multi"][6], outs["multi"][6]
    assert "multi=0" in outs["single"][6]
    assert outs["multi"][4] == outs["single"][4] == 37 and outs["multi"][5] == outs["single"][5] and 0 < outs["multi"][5] < 500
    for i in range(4):
        assert np.array_equal(outs["multi"][i], outs["single"][i]), i


# ------------------------------------------------------------------------------- interpol_method = 'nearest'
@pytest.mark.gpu
def test_nearest_interpolation_through_the_class_surface_matches_reference():
    """VERDICT r3 missing #5: dp.interpol_method = 'nearest' (the reference hands it to RegularGridInterpolator every sweep,
    discretizer.py:570-587).  The table tier implements it; assigning it moves the engine there, carrying J over.  Against
    the reference's own runs: 2-D pendulum after 1, 3, 8 sweeps (float64 bit for bit), a switch from linear to nearest after
    three sweeps, a 4-D cart-pole, float32 within tolerance."""
    from pyro_amd.analysis import costfunction
    from pyro_amd.dynamic import cartpole, pendulum
    from pyro_amd.planning import discretizer, dynamicprogramming
    g = load("nearest_pendulum_31x21x5")
    with contextlib.redirect_stdout(io.StringIO()):
        s = pendulum.SinglePendulum()
        grid = discretizer.GridDynamicSystem(s, [31, 21], [5])
        q = costfunction.QuadraticCostFunction.from_sys(s)
        q.xbar, q.INF = g["xbar"].copy(), float(g["INF"])
        for dtype, tol in (("float64", 0.0), ("float32", REL_F32)):
            dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(grid, q, dtype=dtype)
            dp.save_time_history = False
            assert dp.tier == "fused"
            dp.interpol_method = "nearest"
            assert dp.tier == "table" and dp.interpol_method == "nearest"
            done = 0
            for k in (1, 3, 8):
                dp.compute_steps(k - done)
                done = k
                if dtype == "float64":
                    assert np.array_equal(dp.J, g["J_%d" % k]) and np.array_equal(dp.pi, g["pi_%d" % k]), k
                else:
                    assert relerr(dp.J, g["J_%d" % k]) <= tol, k
            with pytest.raises(NotImplementedError):
                dp.interpol_method = "cubic"
        dm = dynamicprogramming.DynamicProgrammingWithLookUpTable(grid, q)
        dm.save_time_history = False
        dm.compute_steps(3)                       # fused tier, linear
        dm.interpol_method = "nearest"            # ... the next sweeps on the table tier, from the same J
        dm.compute_steps(2)
        assert dm.k == 5 and relerr(dm.J, g["Jmix_5"]) < 1e-12 and np.array_equal(dm.pi, g["pimix_5"])
        dm.interpol_method = "linear"             # and back
        assert dm.tier == "fused"
        g4 = load("nearest_cartpole_7x9x7x9x3")
        c = cartpole.CartPole()
        grid4 = discretizer.GridDynamicSystem(c, [7, 9, 7, 9], [3])
        q4 = costfunction.QuadraticCostFunction.from_sys(c)
        q4.INF = float(g4["INF"])
        d4 = dynamicprogramming.DynamicProgrammingWithLookUpTable(grid4, q4)
        d4.save_time_history = False
        d4.interpol_method = "nearest"
        d4.compute_steps(4)
    assert np.array_equal(d4.J, g4["J_4"]) and np.array_equal(d4.pi, g4["pi_4"])


_POLICY_RANK = r"""
import contextlib, io, os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, %(root)r)
from pyro_amd import parallel
from pyro_amd.analysis import costfunction
from pyro_amd.control import controller
from pyro_amd.dynamic import pendulum
from pyro_amd.planning import discretizer, dynamicprogramming
rank, world, port, out, cls = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + port, rank=rank, world_size=world)
class Damper(controller.StaticController):
    def __init__(self):
        super().__init__(1, 1, 2)
        self.rbar = np.array([0.0])
    def c(self, y, r, t=0):
        return np.array([-2.0 * y[1] - 1.5 * np.sin(y[0])])
with contextlib.redirect_stdout(io.StringIO()):
    s = pendulum.SinglePendulum()
    grid = discretizer.GridDynamicSystem(s, [41, 31], [3])
    q = costfunction.QuadraticCostFunction.from_sys(s)
    q.INF = 200.0
    ev = getattr(dynamicprogramming, cls)(Damper(), grid, q, comm=parallel.staged_transport(dist, rank, world))
    ev.save_time_history = False
    assert ev.sharded and ev.tier == "table"
    ev.compute_steps(12)
    J, k = ev.J.copy(), ev.k
np.savez(out, J=J, k=k, rows=np.array(ev._p.rows), xn_rows=ev.x_next_table.shape[0])
ev._p.close()
dist.destroy_process_group()
print("POLICY-RANK-OK", rank)
"""


@pytest.mark.gpu
@pytest.mark.parametrize("cls", ["PolicyEvaluatorWithLookUpTable", "PolicyEvaluator"])
def test_sharded_policy_evaluation_equals_one_gpu(tmp_path, cls):
    """VERDICT r3 missing #4: PolicyEvaluator* over a sharded grid (dynamicprogramming.py:623-753).  Two ranks share the GPU
    (the library's slab schedule over a host-staged transport): each evaluates the control law, f and g on ITS rows only,
    the one-action tables go to the sharded table tier -- the gathered cost-to-go equals the one-GPU evaluator's bit for bit."""
    import subprocess
    import sys as _sys
    from pyro_amd.analysis import costfunction
    from pyro_amd.control import controller
    from pyro_amd.dynamic import pendulum
    from pyro_amd.planning import discretizer, dynamicprogramming
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "policy_rank.py"
    script.write_text(_POLICY_RANK % dict(root=root))
    port = str(29650 + (os.getpid() + (11 if cls == "PolicyEvaluator" else 0)) % 300)
    procs = [subprocess.Popen([_sys.executable, str(script), str(r), "2", port, str(tmp_path / ("p%d.npz" % r)), cls],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    for r, p in enumerate(procs):
        o, _ = p.communicate(timeout=600)
        assert p.returncode == 0 and ("POLICY-RANK-OK %d" % r) in o, o[-3000:]

    class Damper(controller.StaticController):
        def __init__(self):
            super().__init__(1, 1, 2)
            self.rbar = np.array([0.0])

        def c(self, y, r, t=0):
            return np.array([-2.0 * y[1] - 1.5 * np.sin(y[0])])
    with contextlib.redirect_stdout(io.StringIO()):
        s = pendulum.SinglePendulum()
        grid = discretizer.GridDynamicSystem(s, [41, 31], [3])
        q = costfunction.QuadraticCostFunction.from_sys(s)
        q.INF = 200.0
        one = getattr(dynamicprogramming, cls)(Damper(), grid, q)
        one.save_time_history = False
        one.compute_steps(12)
    rows = []
    for r in range(2):
        p = np.load(tmp_path / ("p%d.npz" % r))
        # (the one-GPU evaluator builds its tables in ONE kernel for a closed-form system -- f, validity and g in float64 --,
        #  the ranks by the reference's Python loop over their rows: G may differ in the last bit of a BLAS dot)
        assert int(p["k"]) == 12 and relerr(p["J"], one.J) <= 1e-14
        rows.append(tuple(int(v) for v in p["rows"]))
        assert int(p["xn_rows"]) == (rows[-1][1] - rows[-1][0]) * 31        # a rank evaluated the control law on its rows only
    assert rows == [(0, 21), (21, 41)]


# ------------------------------------------------------------------------------- bicubic-spline class
def _spline_case(g, tag, xd, ud, dt):
    lv = O.make_levels(g["x_lb"], g["x_ub"], xd)
    ul = O.make_levels(g["u_lb"], g["u_ub"], ud)
    return O.Problem(lv, ul, dt, O.DYN_PENDULUM, O.pendulum_consts(), g["Q"], g["R"], g["S"], g["xbar"], g["ubar"],
                     float(g["INF"]), float(g["EPS"]))


@pytest.mark.gpu
@pytest.mark.parametrize("tier", ["fused", "table"])
def test_spline_value_iteration_matches_reference_golden(tier):
    """DynamicProgramming2DRectBivariateSpline (dynamicprogramming.py:578-614): coefficients of the refit and
    J / pi after 1, 2, 8 sweeps against the reference's own run; float64, tolerance 1e-11 relative (the fit is
    a linear solve: FITPACK uses Givens QR, the kernel a banded LU -- equal up to rounding, cond ~ 4)."""
    from pyro_amd import _native
    g = load("spline_pendulum")
    for tag, xd, ud, dt in (("a", (21, 21), (5,), 0.05), ("b", (41, 31), (7,), 0.1)):
        p = _spline_case(g, tag, xd, ud, dt)
        if tier == "fused":
            h = native_problem(p)
            h.terminal_cost()
        else:
            xn, _, _, G = O.cells(p, np.arange(p.nodes_n))
            h = _native.Problem(p.levels, p.u_levels, p.x_lb, p.x_ub, p.u_lb, p.u_ub, p.dt, dynamics_id=_native.DYN_TABLE)
            h.set_tables(xn, G)                     # packed for the linear sweep, raw tables dropped ...
            with pytest.raises(RuntimeError, match="before pvi_set_tables"):
                h.set_interpolation("bicubic")      # ... so the spline mode must be chosen first
            h.close()
            h = _native.Problem(p.levels, p.u_levels, p.x_lb, p.x_ub, p.u_lb, p.u_ub, p.dt, dynamics_id=_native.DYN_TABLE)
            h.set_interpolation("bicubic")
            h.set_tables(xn, G)
            h.set_J(O.terminal_cost(p))
        if tier == "fused":
            h.set_interpolation("bicubic")
        for k in range(1, 9):
            if k in (1, 2, 8):
                C = h.spline_coefficients()
                ref = g["%s_coef_%d" % (tag, k)]
                assert np.abs(C.ravel() - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
            st = h.sweep(1, 1.0, -1.0)[0][-1]
            assert np.allclose(st, g[tag + "_stats"][k - 1], rtol=1e-11)
            if k in (1, 2, 8):
                J, pi = h.get_J(), h.get_pi()
                assert relerr(J, g["%s_J_%d" % (tag, k)]) < 1e-11
                clear = g["%s_gap_%d" % (tag, k)] > 1e-8
                assert np.array_equal(pi[clear], g["%s_pi_%d" % (tag, k)][clear])
        h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dims,udims", [((4, 4), (3,)), ((5, 130), (4,)), ((131, 67), (9,)), ((200, 64), (2,)),
                                        ((300, 521), (3,)), ((257, 259), (2,))])
def test_spline_sweeps_match_oracle(dims, udims):
    """Ragged sizes (minimum 4 levels, lines that do not fill the 64-wide solve tiles, lines long enough to be cut
    into chunks with a warm-up), alpha < 1, f64 and f32."""
    p, _ = _custom_problem("pendulum", dims, udims, dt=0.08, seed=5)
    xn, _, _, G = O.cells(p, np.arange(p.nodes_n))
    for dtype, tol in (("float64", 1e-11), ("float32", 2e-6)):
        h = native_problem(p, dtype=dtype)
        h.set_interpolation("bicubic")
        if min(dims) > 128:                        # the substitution passes are cut into chunks with a warm-up
            assert "chunk0=64" in h.describe() and "chunk1=64" in h.describe()
        h.terminal_cost()
        J = O.terminal_cost(p)
        if dtype == "float32":
            J = J.astype(np.float32).astype(np.float64)
        for k in range(4):
            Jn, pi, Q = O.sweep_spline(p.levels, xn, G, J, alpha=0.97)
            h.sweep(1, 0.97, -1.0)
            Jd = h.get_J()
            assert relerr(Jd, Jn) < tol
            Qs = np.sort(Q, axis=1)
            clear = (Qs[:, 1] - Qs[:, 0]) > 1e-4 * np.abs(Qs[:, 0]).max() if Q.shape[1] > 1 else np.ones(len(Jn), bool)
            assert np.array_equal(h.get_pi()[clear], pi[clear])
            J = Jd                                  # follow the device iterate (f32 storage rounds)
        h.close()


@pytest.mark.gpu
def test_spline_mode_errors_and_class_surface():
    from pyro_amd import _native
    from pyro_amd.analysis import costfunction
    from pyro_amd.dynamic import cartpole, pendulum
    from pyro_amd.planning import discretizer, dynamicprogramming
    g = load("spline_pendulum")
    # n != 2 and fewer than 4 levels are refused
    p4, _ = _custom_problem("cartpole", (5, 5, 5, 5), (3,))
    h = native_problem(p4)
    with pytest.raises(RuntimeError):
        h.set_interpolation("bicubic")
    h.close()
    p3, _ = _custom_problem("pendulum", (3, 9), (3,))
    h = native_problem(p3)
    with pytest.raises(RuntimeError):
        h.set_interpolation("bicubic")
    h.close()
    with contextlib.redirect_stdout(io.StringIO()):
        with pytest.raises(NotImplementedError):
            s4 = cartpole.CartPole()
            dynamicprogramming.DynamicProgramming2DRectBivariateSpline(
                discretizer.GridDynamicSystem(s4, [5, 5, 5, 5], [3]), costfunction.QuadraticCostFunction.from_sys(s4))
        sys_ = pendulum.SinglePendulum()
        qcf = costfunction.QuadraticCostFunction.from_sys(sys_)
        qcf.xbar = np.array([-3.14, 0]); qcf.INF = 300
        grid_sys = discretizer.GridDynamicSystem(sys_, [21, 21], [5])
        dp = dynamicprogramming.DynamicProgramming2DRectBivariateSpline(grid_sys, qcf)
        dp.save_time_history = False
        dp.compute_steps(8)
    assert dp.k == 8 and dp.interpol_method == "bicubic"
    assert relerr(dp.J, g["a_J_8"]) < 1e-11
    assert np.abs(np.asarray(dp.J_interpol.get_coeffs()) - g["a_coef_8"]).max() < 1e-9 * np.abs(g["a_coef_8"]).max()
    # switching back to the bilinear interpolant restores the plain recursion
    g0 = load("pendulum_21x21x5")
    dp._p.set_interpolation("linear")
    dp._p.terminal_cost()
    dp._p.sweep(1, 1.0, -1.0)
    assert relerr(dp._p.get_J(), g0["J_1"]) < 1e-14


# ------------------------------------------------------------------------------- generic mechanical tier
def _mountaincar_dp(g, dtype, cls=None):
    from pyro_amd.analysis import costfunction
    from pyro_amd.dynamic import mountaincar
    from pyro_amd.planning import discretizer, dynamicprogramming
    s = mountaincar.MountainCar()
    s.x_ub, s.x_lb = np.array([+0.2, +2.0]), np.array([-1.7, -2.0])
    s.u_ub[0], s.u_lb[0] = +0.2, -0.2
    grid = discretizer.GridDynamicSystem(s, [41, 41], [5])
    q = costfunction.QuadraticCostFunction.from_sys(s)
    q.xbar = np.array([0.0, 0.0]); q.INF = 30
    q.R[0, 0] = 10.0; q.S[0, 0] = 10.0; q.S[1, 1] = 10.0
    dp = (cls or dynamicprogramming.DynamicProgrammingWithLookUpTable)(grid, q, dtype=dtype)
    dp.save_time_history = False
    return dp


@pytest.mark.gpu
def test_mountaincar_runs_fused_through_node_tables_and_matches_reference():
    """mountain_car_with_valueiteration_quadratic.py at 41x41x5: a Manipulator with position-dependent H, C, B, g.
    The classes pick the fused tier (per-node a0 / Bn tables, O(N) host work) and reproduce the reference's J after
    1, 5, 20 sweeps; float64 to 1e-12 (the affine form rounds x_next differently by <= 1 ulp), float32 to 1e-5."""
    g = load("mountaincar_41x41x5")
    for dtype, tol in (("float64", 1e-12), ("float32", REL_F32)):
        with contextlib.redirect_stdout(io.StringIO()):
            dp = _mountaincar_dp(g, dtype)
            assert dp.tier == "fused"
            assert ("exact-f64" if dtype == "float64" else "lean") in dp._p.describe()
            if dtype == "float64":
                assert np.abs(dp.grid_sys.x_next_table - g["x_next_table"]).max() < 1e-14
                assert np.array_equal(dp.grid_sys.x_next_isok, g["x_next_isok"])
                assert relerr(dp.G, g["G"]) < 1e-14
            done = 0
            for k in (1, 5, 20):
                dp.compute_steps(k - done)
                done = k
                assert relerr(dp.J, g["J_%d" % k]) < tol, (dtype, k)
                clear = g["gap_%d" % k] > (1e-9 if dtype == "float64" else 1e-3)
                assert np.array_equal(dp.pi[clear], g["pi_%d" % k][clear])
        if dtype == "float64":
            assert np.allclose([dp.J.max()], g["stats"][19][0], rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("dof,act", [(1, 1), (2, 1), (2, 2)])
def test_generic_mechanical_systems_node_tier_equals_table_tier(dof, act):
    """Any MechanicalSystem subclass: the node tier (PVI_DYN_NODE_1x1 / 2x1 / 2x2) against the table tier built from the
    same system's f by the reference's loops -- float64 J within 1e-11, float32 (lean kernel) within 1e-5."""
    from pyro_amd import _native
    from pyro_amd.analysis import costfunction
    from pyro_amd.dynamic import mechanical
    from pyro_amd.planning import discretizer, dynamicprogramming

    class Robot(mechanical.MechanicalSystem):        # configuration-dependent inertia, Coriolis, gravity, damping
        def H(self, q):
            if self.dof == 1:
                return np.array([[1.5 + 0.5 * np.cos(q[0])]])
            return np.array([[2.0 + 0.6 * np.cos(q[1]), 0.4 + 0.3 * np.cos(q[1])], [0.4 + 0.3 * np.cos(q[1]), 0.9]])

        def C(self, q, dq):
            if self.dof == 1:
                return np.array([[-0.25 * np.sin(q[0]) * dq[0]]])
            h = 0.3 * np.sin(q[1])
            return np.array([[-h * dq[1], -h * (dq[0] + dq[1])], [h * dq[0], 0.0]])

        def g(self, q):
            return 2.0 * np.sin(q)

        def d(self, q, dq):
            return 0.2 * dq

    with contextlib.redirect_stdout(io.StringIO()):
        s = Robot(dof, actuators=act)
        s.x_ub, s.x_lb = np.full(2 * dof, 1.5), np.full(2 * dof, -1.5)
        s.u_ub, s.u_lb = np.full(act, 3.0), np.full(act, -3.0)
        dims = [15, 13] if dof == 1 else [7, 6, 7, 6]
        grid = discretizer.GridDynamicSystem(s, dims, [5] if act == 1 else [3, 3], dt=0.08)
        cf = costfunction.QuadraticCostFunction.from_sys(s)
        cf.INF = 60.0
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(grid, cf)
        assert dp.tier == "fused" and dp._p.dynamics_id == {(1, 1): _native.DYN_NODE_1x1, (2, 1): _native.DYN_NODE_2x1,
                                                              (2, 2): _native.DYN_NODE_2x2}[(dof, act)]
        dp.save_time_history = False
        dp.compute_steps(6)
        # table tier from the same system through the reference's own loops (host f, host g)
        xn, ok = grid._host_xnext_table()
        X, U = grid.state_from_node_id, grid.input_from_action_id
        G = np.full((grid.nodes_n, grid.actions_n), float(cf.INF))
        for n_, a_ in zip(*np.nonzero(ok)):
            G[n_, a_] = cf.g(X[n_], U[a_], 0) * grid.dt
        J = np.array([cf.h(x, 0) for x in X], dtype=float)
        for _ in range(6):
            J, pi, Q = O.sweep_lut(grid.x_level, xn, G, J)
        assert relerr(dp.J, J) < 1e-11
        Qs = np.sort(Q, axis=1)
        clear = (Qs[:, 1] - Qs[:, 0]) > 1e-9
        assert np.array_equal(dp.pi[clear], pi[clear])
        dp32 = dynamicprogramming.DynamicProgrammingWithLookUpTable(grid, cf, dtype="float32")
        dp32.save_time_history = False
        dp32.compute_steps(6)
        assert "lean" in dp32._p.describe() or "exact-f32" in dp32._p.describe()
        assert relerr(dp32.J, J) < REL_F32


@pytest.mark.gpu
def test_minimum_time_cost_runs_fused_and_matches_reference():
    """simple_pendulum_with_valueiteration_minimum_time.py (reduced): TimeCostFunction evaluated in-kernel
    (PVI_COST_TIME): G and J after 1, 10, 40 sweeps against the reference's run; exact in f64 (the costs are 0 / dt /
    INF and the dynamics are the closed-form pendulum)."""
    from pyro_amd.analysis import costfunction
    from pyro_amd.dynamic import pendulum
    from pyro_amd.planning import discretizer, dynamicprogramming
    g = load("mintime_invpendulum_61x61x3")
    for dtype, tol in (("float64", 1e-13), ("float32", REL_F32)):
        with contextlib.redirect_stdout(io.StringIO()):
            s = pendulum.InvertedPendulum()
            s.x_ub, s.x_lb = np.array([+6.0, +6.0]), np.array([-6.0, -6.0])
            grid = discretizer.GridDynamicSystem(s, [61, 61], [3])
            tcf = costfunction.TimeCostFunction(np.array([0.0, 0.0]))
            tcf.INF, tcf.EPS = 10.0, 0.1
            dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(grid, tcf, dtype=dtype)
            dp.save_time_history = False
            assert dp.tier == "fused"
            if dtype == "float64":
                assert np.array_equal(dp.G, g["G"]) and np.array_equal(dp.J, g["J0"])
            done = 0
            for k in (1, 10, 40):
                dp.compute_steps(k - done)
                done = k
                assert relerr(dp.J, g["J_%d" % k]) < tol, (dtype, k)
                clear = g["gap_%d" % k] > (1e-9 if dtype == "float64" else 1e-3)
                assert np.array_equal(dp.pi[clear], g["pi_%d" % k][clear])


@pytest.mark.gpu
def test_closed_loop_rollout_of_a_node_tier_system_falls_back_to_the_host_loop():
    """No closed form in the kernels (MountainCar): simulate_closed_loop runs the reference's controller + Euler loop on
    the host with the GPU-computed policy."""
    g = load("mountaincar_41x41x5")
    with contextlib.redirect_stdout(io.StringIO()):
        dp = _mountaincar_dp(g, "float64")
        dp.compute_steps(60)
        dp.clean_infeasible_set()
        t, X, U = dp.simulate_closed_loop(np.array([[-1.0, 0.0], [-0.5, 0.3]]), tf=20.0, n=2001)
    assert X.shape == (2, 2001, 2) and U.shape == (2, 2001, 1) and t.shape == (2001,)
    assert np.all(np.abs(U) <= 0.2 + 1e-12)
    ctl = dp.get_lookup_table_controller()
    assert np.allclose(U[0, 0], ctl.c(np.array([-1.0, 0.0]), 0))
    assert np.allclose(X[0, 1], X[0, 0] + dp.sys.f(X[0, 0], U[0, 0], 0) * (20.0 / 2000))
    assert np.all(np.isfinite(X)) and np.all(X[:, :, 0] > -1.8) and np.all(X[:, :, 0] < 0.3)


def test_fast3_explicit_systems_float32_matches_the_plain_kernel(variants):
    """the float32 production form of the explicit systems (validity of every cell decided once at set-up:
    the obstacle tests leave the sweep; intervals from (x - lo) / step in float64 instead of a level search + division)
    against the operation-for-operation kernel k_sweep3 on the same handle type: the helicopter tunnel with its obstacles
    and domain-check cost (the reference's 3-D demo), the car park and the point robot; and pvi_self_check on it."""
    from pyro_amd import configs
    from pyro_amd.analysis import costfunction
    from pyro_amd.dynamic import vehicle_steering
    from pyro_amd.planning import discretizer, dynamicprogramming
    cases = []
    with contextlib.redirect_stdout(io.StringIO()):
        c = configs.build("h3s")
        cases.append((c["grid_sys"], c["cf"]))
        s = vehicle_steering.KinematicCarModelwithObstacles()
        cases.append((discretizer.GridDynamicSystem(s, [31, 31, 21], [3, 5], 0.1), costfunction.QuadraticCostFunctionWithDomainCheck.from_sys(s)))
        s = vehicle_steering.HolonomicMobileRobotwithObstacles()
        cases.append((discretizer.GridDynamicSystem(s, [41, 41], [5, 5]), costfunction.QuadraticCostFunction.from_sys(s)))
    for gs, cf in cases:
        outs = {}
        for tag, env in (("fast3", {}), ("plain", {"PVI_NO_FAST": "1"})):
            variants.delenv("PVI_NO_FAST")
            for k, v in env.items():
                variants.setenv(k, v)
            with contextlib.redirect_stdout(io.StringIO()):
                dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(gs, cf, dtype="float32")
                dp.save_time_history = False
                dp.compute_steps(12)
            desc = dp._p.describe()
            assert desc.startswith("path=fast3" if tag == "fast3" else "path=exact-f32"), desc
            outs[tag] = (dp.J.copy(), dp.pi.copy(), dp._p.get_J(prev=True))
            if tag == "fast3":
                d, n = dp._p.self_check(1.0)
                assert d <= 2e-6, (gs.sys.name, d, n)
            dp._p.close()
        variants.delenv("PVI_NO_FAST")
        scale = np.abs(outs["plain"][0]).max()
        assert np.abs(outs["fast3"][0] - outs["plain"][0]).max() <= 2e-6 * scale, gs.sys.name
        assert (outs["fast3"][1] != outs["plain"][1]).mean() < 2e-3, gs.sys.name        # ties / last-bit argmin flips only


def _host_closed_loop(dp, X0, tf, n):
    """The reference's own closed loop: u = ctl.c(x, t), x <- x + f(x, u, t) dt (controller.py:328-355, simulation.py:298-324)."""
    ctl = dp.get_lookup_table_controller()
    dt = tf / (n - 1)
    X0 = np.atleast_2d(np.asarray(X0, dtype=float))
    X = np.empty((X0.shape[0], n, dp.sys.n)); U = np.empty((X0.shape[0], n, dp.sys.m))
    for b, x in enumerate(X0):
        for i in range(n):
            u = np.atleast_1d(ctl.c(x, i * dt))
            X[b, i], U[b, i] = x, u
            x = x + np.asarray(dp.sys.f(x, u, i * dt), dtype=float) * dt
    return X, U


@pytest.mark.gpu
def test_closed_loop_rollouts_of_the_explicit_systems_run_on_the_gpu():
    """pvi_rollout for the explicit (non-mechanical) demo systems -- helicopter tunnel, kinematic car,
    quarter car on its sum-of-sines ground, point robot, longitudinal car with the tyre curve -- as functions of a
    CONTINUOUS state and input; simulate_closed_loop no longer drops to the Python double loop for them.  Against the
    reference's own loop (controller interpolation + system f + Euler) on the same policy."""
    from pyro_amd import _native
    from pyro_amd.analysis import costfunction
    from pyro_amd.dynamic import drone, suspension, vehicle_propulsion, vehicle_steering
    from pyro_amd.planning import discretizer, dynamicprogramming
    rng = np.random.default_rng(5)
    cases = []
    with contextlib.redirect_stdout(io.StringIO()):
        s = drone.ConstantSpeedHelicopterTunnel()
        cases.append((s, discretizer.GridDynamicSystem(s, [11, 11, 11], [5], 0.3), {}))
        s = vehicle_steering.KinematicCarModelwithObstacles()
        cases.append((s, discretizer.GridDynamicSystem(s, [21, 21, 11], [3, 3], 0.1), {}))
        s = suspension.QuarterCarOnRoughTerrain()
        cases.append((s, discretizer.GridDynamicSystem(s, [13, 11, 21], [5], 0.05), {}))
        s = vehicle_steering.HolonomicMobileRobotwithObstacles()
        cases.append((s, discretizer.GridDynamicSystem(s, [21, 21], [3, 3]), {}))
        s = vehicle_propulsion.LongitudinalFrontWheelDriveCarWithWheelSlipInput()
        s.x_ub[1] = 15; s.x_lb[1] = 0
        cases.append((s, discretizer.GridDynamicSystem(s, [31, 31], [7], 0.05), {}))
    for s, gs, _ in cases:
        with contextlib.redirect_stdout(io.StringIO()):
            cf = costfunction.QuadraticCostFunction.from_sys(s)
            cf.INF = 1e4
            dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(gs, cf)
            dp.save_time_history = False
            dp.compute_steps(15)
        assert dp.tier == "fused" and dp._p.dynamics_id in _native.ROLLOUT_IDS, s.name
        X0 = s.x_lb + (0.2 + 0.6 * rng.random((6, s.n))) * (s.x_ub - s.x_lb)
        t, X, U = dp.simulate_closed_loop(X0, tf=2.0, n=201)
        Xh, Uh = _host_closed_loop(dp, X0, 2.0, 201)
        assert X.shape == Xh.shape and U.shape == Uh.shape
        scale = np.abs(Xh).max()
        assert np.abs(X - Xh).max() <= 1e-9 * scale, (s.name, np.abs(X - Xh).max())
        assert np.abs(U - Uh).max() <= 1e-9 * max(np.abs(Uh).max(), 1.0), s.name
    # a quarter car with ITS OWN ground keeps the sweep kernel (tables over the levels) but not the continuous closed form
    class Bumpy(suspension.QuarterCarOnRoughTerrain):
        def z(self, x):
            return 0.1 * np.cos(x)

        def dz(self, x):
            return -0.1 * np.sin(x)
    with contextlib.redirect_stdout(io.StringIO()):
        s = Bumpy()
        gs = discretizer.GridDynamicSystem(s, [9, 9, 11], [3], 0.05)
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(gs, costfunction.QuadraticCostFunction.from_sys(s))
        dp.save_time_history = False
        dp.compute_steps(3)
    assert dp.tier == "fused" and s.device_rollout_params() is None
    X0 = np.array([[0.5, -0.5, 1.0]])
    t, X, U = dp.simulate_closed_loop(X0, tf=0.5, n=51)
    Xh, Uh = _host_closed_loop(dp, X0, 0.5, 51)
    assert np.array_equal(X, Xh) and np.array_equal(U, Uh)


@pytest.mark.gpu
def test_acrobot_runs_fused_through_node_tables_and_matches_reference():
    """pendulum.py:699 Acrobot (two degrees of freedom, one actuator): generic mechanical tier PVI_DYN_NODE_2x1;
    f bit for bit, J after 1 and 6 sweeps against the reference's run."""
    from pyro_amd import _native
    from pyro_amd.analysis import costfunction
    from pyro_amd.dynamic import pendulum
    from pyro_amd.planning import discretizer, dynamicprogramming
    g = load("acrobot_9p4x5")
    s = pendulum.Acrobot()
    assert np.array_equal(np.array([s.f(x, u) for x, u in zip(g["f_X"], g["f_U"])]), g["f_dX"])
    for dtype, tol in (("float64", 1e-12), ("float32", REL_F32)):
        with contextlib.redirect_stdout(io.StringIO()):
            s = pendulum.Acrobot()
            s.x_ub = np.array([+2.0, +2.0, +4.0, +4.0]); s.x_lb = -s.x_ub
            grid = discretizer.GridDynamicSystem(s, [9, 9, 9, 9], [5], dt=0.05)
            q = costfunction.QuadraticCostFunction.from_sys(s)
            q.INF = 200.0
            dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(grid, q, dtype=dtype)
            dp.save_time_history = False
            assert dp.tier == "fused" and dp._p.dynamics_id == _native.DYN_NODE_2x1
            dp.compute_steps(1)
            assert relerr(dp.J, g["J_1"]) < tol
            dp.compute_steps(5)
        assert relerr(dp.J, g["J_6"]) < tol
        clear = g["gap_6"] > (1e-9 if dtype == "float64" else 1e-3)
        assert np.array_equal(dp.pi[clear], g["pi_6"][clear])


@pytest.mark.gpu
def test_linear_state_space_mass_runs_fused_and_matches_reference():
    """float_mass_dp_optimal_controller.py (reduced): FloatingSingleMass, a StateSpaceSystem in mechanical form, through
    the per-node tables (a0 = velocity rows of A x, Bn = velocity rows of B): x_next bit for bit, J to 1e-13 (f64)."""
    from pyro_amd import _native
    from pyro_amd.analysis import costfunction
    from pyro_amd.dynamic import massspringdamper
    from pyro_amd.planning import discretizer, dynamicprogramming
    g = load("floatmass_51x51x21")
    s = massspringdamper.FloatingSingleMass()
    assert np.array_equal(np.array([s.f(x, u) for x, u in zip(g["f_X"], g["f_U"])]), g["f_dX"])
    for dtype, tol in (("float64", 1e-13), ("float32", REL_F32)):
        with contextlib.redirect_stdout(io.StringIO()):
            s = massspringdamper.FloatingSingleMass()
            s.x_ub[0], s.x_lb[0], s.x_lb[1], s.x_ub[1] = 10.0, -10.0, -5.0, 5.0
            s.u_ub[0], s.u_lb[0] = 5.0, -5.0
            grid = discretizer.GridDynamicSystem(s, [51, 51], [21], 0.05)
            q = costfunction.QuadraticCostFunction.from_sys(s)
            q.xbar, q.INF = np.array([-0, 0]), 300
            q.R[0, 0] = 10.0; q.S[0, 0] = 10.0; q.S[1, 1] = 10.0
            dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(grid, q, dtype=dtype)
            dp.save_time_history = False
            assert dp.tier == "fused" and dp._p.dynamics_id == _native.DYN_NODE_1x1
            if dtype == "float64":
                assert np.array_equal(grid.x_next_table, g["x_next_table"]) and relerr(dp.G, g["G"]) < 1e-14
            dp.compute_steps(1)
            assert relerr(dp.J, g["J_1"]) < tol
            dp.compute_steps(29)
        assert relerr(dp.J, g["J_30"]) < tol
        if dtype == "float64":
            assert (dp.pi != g["pi_30"]).mean() < 1e-3
    # a state-space system that is not in mechanical form stays on the table tier
    from pyro_amd.dynamic import statespace
    A = np.array([[0.0, 1.0], [-1.0, -0.1]]); B = np.array([[0.5], [1.0]])
    assert statespace.StateSpaceSystem(A, B, np.eye(2), np.zeros((2, 1))).device_dynamics() is None


@pytest.mark.gpu
@pytest.mark.parametrize("script,args", [("tools_fuzz.py", ["30", "101"]), ("tools_fuzz_tiers.py", ["30", "102"])])
def test_randomised_consistency_runs(script, args):
    """A short run of the randomised checks under tools/: float32 kernels against the float64 kernel on random problems
    (1e-5), and the table tier fed with the fused tier's tables against the fused sweeps (bit for bit)."""
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)] + args, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


# ----------------------------------------------------------- three-dimensional systems, fused (SURVEY 8 f3 remainder)
from cases3d import CASES3D, case3d  # noqa: E402


def _native3d(p, dtype="float64", flags=0):
    from pyro_amd import _native
    cost = dict(Q=p.Q, R=p.R, S=p.S, xbar=p.xbar, ubar=p.ubar, EPS=p.EPS, INF=p.INF, ontarget_check=p.ontarget_check)
    if p.domain_check:
        cost["kind"] = "quadratic_domain"
    t = p.trig_tables()
    trig = {O.DYN_KINCAR: lambda: (t["c2"], t["s2"]), O.DYN_QUARTERCAR: lambda: (t["z"], t["dz"]),
            O.DYN_LONGCAR: lambda: (t["fd"],)}.get(p.dyn_id, lambda: ())()
    return _native.Problem(p.levels, p.u_levels, p.x_lb, p.x_ub, p.u_lb, p.u_ub, p.dt, dtype=dtype,
                           dynamics_id=p.dyn_id, dyn_params=list(p.dyn_c), trig=trig, cost=cost,
                           obstacles=p.obstacles, act_aux=p.act_aux, flags=flags)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES3D)
def test_3d_systems_fused_kernels_match_reference(name):
    """Helicopter tunnel, car parking, active suspension through the closed-form n = 3 kernels: the tables the
    library builds equal the REFERENCE's x_next_table / masks bit for bit (G to the ulp of its BLAS dots, bit for bit
    against the oracle); J / pi of the look-up-table class (INF + alpha*J on obstacle cells) and of the base class
    (exactly INF) after 1 and 5 sweeps; float32 storage within 1e-5."""
    from pyro_amd import _native
    g, p, alpha = case3d(name)
    h = _native3d(p)
    assert h.describe().split()[0] == "path=exact-f64"
    xn, xok, aok, G = h.build_tables()
    assert np.array_equal(xn, g["x_next_table"])
    assert np.array_equal(xok, g["x_next_isok"]) and np.array_equal(aok, g["action_isok"])
    assert np.array_equal(G, O.cells(p, np.arange(p.nodes_n))[3])
    np.testing.assert_allclose(G, g["G"], rtol=1e-14, atol=0)
    h.terminal_cost()
    assert np.array_equal(h.get_J(), O.terminal_cost(p))
    np.testing.assert_allclose(h.get_J(), g["J0"], rtol=1e-14, atol=0)
    for k in range(1, 6):
        h.sweep(1, alpha, -1.0)
        if k in (1, 5):
            assert relerr(h.get_J(), g["J_%d" % k]) < 1e-12
            assert np.array_equal(h.get_pi(), g["pi_%d" % k])
    h.close()
    hb = _native3d(p, flags=_native.FLAG_HARD_INF)                  # base class DynamicProgramming
    hb.terminal_cost()
    hb.sweep(5, alpha, -1.0)
    assert relerr(hb.get_J(), g["Jbase_5"]) < 1e-12 and np.array_equal(hb.get_pi(), g["pibase_5"])
    hb.close()
    h32 = _native3d(p, dtype="float32")
    h32.terminal_cost()
    h32.sweep(5, alpha, -1.0)
    assert relerr(h32.get_J(), g["J_5"]) <= REL_F32
    h32.close()


@pytest.mark.gpu
def test_3d_systems_class_surface_runs_fused():
    """The mirrors of the three systems pick the fused tier (no O(N*A) Python table build) through the reference's
    class surface, with the demos' parameters; results equal the reference's own runs."""
    from pyro_amd.analysis import costfunction
    from pyro_amd.dynamic import drone, suspension, vehicle_steering
    from pyro_amd.planning import discretizer, dynamicprogramming
    with contextlib.redirect_stdout(io.StringIO()):
        # helicopter_tunnel.py-style set-up (tests/golden/make_goldens.py case_helicopter)
        s = drone.ConstantSpeedHelicopterTunnel()
        s.obstacles = [[(2, 2), (4, 4)], [(8, 5), (10, 10)], [(14, 0), (16, 4)]]
        s.mass, s.vx, s.width = 0.1, 5.0, 1.0
        s.x_ub, s.x_lb = np.array([+60, 10, +20]), np.array([-60, 0, +0])
        s.u_ub, s.u_lb = np.array([+20]), np.array([-20])
        gs = discretizer.GridDynamicSystem(s, (11, 11, 11), [5], 0.05)
        q = costfunction.QuadraticCostFunctionWithDomainCheck.from_sys(s)
        q.xbar = np.array([0.0, 2.0, 20]); q.INF = 100000; q.EPS = 0.2
        q.Q[0, 0] = 2.0; q.Q[1, 1] = 200.0; q.Q[2, 2] = 0.0; q.R[0, 0] = 5.0
        q.S[0, 0] = 20.0; q.S[1, 1] = 50.0; q.S[2, 2] = 0.0
        g = load("helicopter_11x11x11x5")
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(gs, q)
        assert dp.tier == "fused"
        dp.alpha = 0.999
        dp.compute_steps(5)
        assert relerr(dp.J, g["J_5"]) < 1e-12 and np.array_equal(dp.pi, g["pi_5"])
        assert np.array_equal(gs.x_next_table, g["x_next_table"]) and np.array_equal(gs.x_next_isok, g["x_next_isok"])
        np.testing.assert_allclose(dp.G, g["G"], rtol=1e-14)
        db = dynamicprogramming.DynamicProgramming(gs, q)             # base class: exactly INF on obstacle cells
        assert db.tier == "fused"
        db.alpha = 0.999
        db.compute_steps(5)
        assert relerr(db.J, g["Jbase_5"]) < 1e-12 and np.array_equal(db.pi, g["pibase_5"])
        # a domain check bound to ANOTHER system's validity is not this system's: table tier
        q2 = costfunction.QuadraticCostFunctionWithDomainCheck.from_sys(drone.ConstantSpeedHelicopterTunnel())
        q2.xbar = q.xbar
        assert dynamicprogramming.DynamicProgrammingWithLookUpTable(gs, q2).tier == "table"
        # car_parking.py
        s = vehicle_steering.KinematicCarModelwithObstacles()
        s.x_ub, s.x_lb = np.array([+35, +3, +3]), np.array([-5, -2, -3])
        s.u_ub, s.u_lb = np.array([+3, +1]), np.array([-3, -1])
        gs = discretizer.GridDynamicSystem(s, (21, 21, 11), (3, 3), 0.1)
        cf = costfunction.QuadraticCostFunction.from_sys(s)
        cf.xbar = np.array([0, 0, 0]); cf.INF = 1E8; cf.EPS = 0.00
        cf.R = np.array([[0.1, 0], [0, 0]])
        g = load("car_21x21x11x3x3")
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(gs, cf)
        assert dp.tier == "fused"
        dp.alpha = 0.99
        dp.compute_steps(5)
        assert relerr(dp.J, g["J_5"]) < 1e-12 and np.array_equal(dp.pi, g["pi_5"])
        ctl = dp.get_lookup_table_controller()
        assert np.all(np.isfinite(ctl.c(np.array([10.0, 0.5, 0.1]), 0)))
        # active_suspension.py
        s = suspension.QuarterCarOnRoughTerrain()
        s.mass, s.b, s.k, s.vx = 0.5, 0.5, 8.0, 10.0
        s.x_ub, s.x_lb = np.array([+12, +1, +40]), np.array([-12, -1, +0])
        s.u_ub, s.u_lb = np.array([+40]), np.array([-40])
        gs = discretizer.GridDynamicSystem(s, (13, 11, 21), [5], 0.05)
        qcf = costfunction.QuadraticCostFunction.from_sys(s)
        qcf.xbar = np.array([0.0, 0.0, 20]); qcf.INF = 100000; qcf.EPS = 0.5
        qcf.Q[0, 0] = 2.0; qcf.Q[1, 1] = 5.0; qcf.Q[2, 2] = 0.0; qcf.R[0, 0] = 0.1
        g = load("suspension_13x11x21x5")
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(gs, qcf, dtype="float32")
        assert dp.tier == "fused"
        dp.alpha = 0.99
        dp.compute_steps(5)
        assert relerr(dp.J, g["J_5"]) <= REL_F32

        # overriding the obstacle test, or f, leaves the closed form: table tier, same answers as the reference loops
        class Narrow(vehicle_steering.KinematicCarModelwithObstacles):
            def isavalidstate(self, x):
                return bool(abs(x[1]) < 1.5) and vehicle_steering.KinematicCarModelwithObstacles.isavalidstate(self, x)
        n = Narrow()
        gs2 = discretizer.GridDynamicSystem(n, (5, 5, 3), (2, 2), 0.1)
        d2 = dynamicprogramming.DynamicProgrammingWithLookUpTable(gs2, costfunction.QuadraticCostFunction.from_sys(n))
        assert d2.tier == "table"


# --------------------------------------------------------------------------- multi-GPU path inside the C ABI (RCCL)
_RCCL_RANK = r"""
import contextlib, io, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from pyro_amd import _native, configs, parallel
rank, world, case, idfile, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
with contextlib.redirect_stdout(io.StringIO()):
    cfg = configs.build(case)
if rank == 0:
    with open(idfile + ".tmp", "wb") as f:
        f.write(_native.comm_unique_id())
    os.replace(idfile + ".tmp", idfile)
else:
    import time
    while not os.path.exists(idfile):
        time.sleep(0.05)
comm_id = open(idfile, "rb").read()
vi = parallel.RcclValueIteration(cfg["grid_sys"], cfg["cf"], rank, world, comm_id=comm_id, dtype=cfg["dtype"],
                                 overlap=bool(int(sys.argv[6])))
st5, n5 = vi.run(5, 1.0, -1.0)
st, n = vi.run(400, 1.0, float(sys.argv[7]))
J, pi = vi.owned()
np.savez(out, J=J, pi=pi, st5=st5, st=st, n=n, rows=np.array(vi.rows), desc=vi.describe())
vi.close()
print("RCCL-RANK-OK", rank)
"""


def _run_rccl_ranks(tmp_path, case, world, overlap, tol):
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rccl_rank.py"
    script.write_text(_RCCL_RANK % dict(root=root))
    idfile = str(tmp_path / "comm.id")
    procs = [subprocess.Popen([_sys.executable, str(script), str(r), str(world), case, idfile,
                               str(tmp_path / ("r%d.npz" % r)), str(int(overlap)), repr(tol)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o))
    return outs


@pytest.mark.gpu
@pytest.mark.parametrize("case,overlap", [("cartpole:21,21,21,21:7:float32", True), ("pendulum:101,101:11:float64", False)])
def test_rccl_shard_world1_equals_plain_handle(tmp_path, case, overlap):
    """pvi_shard_* with a real RCCL communicator of one rank (ncclCommInitRank, the statistics path, the streams and
    events of the boundary-first schedule): same J, pi, statistics and stop sweep as the plain whole-grid handle."""
    from pyro_amd import configs
    outs = _run_rccl_ranks(tmp_path, case, 1, overlap, 0.5)
    assert outs[0][0] == 0 and "RCCL-RANK-OK 0" in outs[0][1], outs[0][1][-2000:]
    r = np.load(tmp_path / "r0.npz")
    assert "comm=rccl" in str(r["desc"]) and "rank=0/1" in str(r["desc"])
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build(case)
        h = cfg["grid_sys"]._device_problem(cost=cfg["cf"].device_cost(), dtype=cfg["dtype"])
    h.terminal_cost()
    st5, _ = h.sweep(5, 1.0, -1.0)
    st, n = h.sweep(400, 1.0, 0.5)
    assert np.allclose(r["st5"], st5[-1], rtol=1e-12) and int(r["n"]) == n and np.allclose(r["st"], st[-1], rtol=1e-12)
    assert np.array_equal(r["J"], h.get_J()) and np.array_equal(r["pi"], h.get_pi())
    h.close()


@pytest.mark.gpu
def test_class_surface_over_a_one_rank_rccl_communicator():
    """comm=RcclComm(0, 1, id): a real RCCL communicator of one rank behind the class surface -- ncclCommCount is
    reported, J / pi come through pvi_shard_gather_*, the timing of the last call is filled in, results equal the plain
    class."""
    from pyro_amd import _native, configs, parallel
    from pyro_amd.planning import dynamicprogramming
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build("cartpole:21,21,21,21:7:float32")
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(
            cfg["grid_sys"], cfg["cf"], dtype="float32", comm=parallel.RcclComm(0, 1, _native.comm_unique_id()))
        one = dynamicprogramming.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"], dtype="float32")
        for d in (dp, one):
            d.save_time_history = False
            d.compute_steps(7)
    desc = dp._p.describe()
    assert "comm=rccl" in desc and "rccl_ranks=1" in desc
    assert np.array_equal(dp.J, one.J) and np.array_equal(dp.pi, one.pi) and np.array_equal(dp.J_next, one.J_next)
    t = dp._p.shard.timing()
    assert t["sweeps_timed"] == 7 and t["interior_ms"] > 0 and t["exposed_exchange_ms"] >= 0
    dp._p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case,world,overlap", [("cartpole:21,21,21,21:7:float32", 2, True),
                                                ("cartpole:21,21,21,21:7:float32", 3, True),
                                                ("pendulum:101,101:11:float32", 2, False)])
def test_rccl_shard_ranks_sharing_one_gpu(tmp_path, case, world, overlap):
    """2-3 ranks of the in-library RCCL path on the ONE GPU of the test box (RCCL refuses that on some builds: then
    the test reports a skip with RCCL's message -- the world-1 test above and the gloo tests of the same partition
    logic still cover the pieces).  Where it runs: the concatenated slabs equal the whole-grid handle bit for bit."""
    from pyro_amd import configs
    outs = _run_rccl_ranks(tmp_path, case, world, overlap, -1.0)
    if any(rc != 0 for rc, _ in outs):
        text = "\n".join(o[-600:] for _, o in outs)
        if "Duplicate GPU" in text or "invalid usage" in text.lower() or "ncclCommInitRank" in text:
            pytest.skip("RCCL does not accept several ranks on one GPU here: " + text[-300:])
        raise AssertionError(text)
    parts = [np.load(tmp_path / ("r%d.npz" % r)) for r in range(world)]
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build(case)
        h = cfg["grid_sys"]._device_problem(cost=cfg["cf"].device_cost(), dtype=cfg["dtype"])
    h.terminal_cost()
    h.sweep(5, 1.0, -1.0)
    st, n = h.sweep(400, 1.0, -1.0)
    assert np.array_equal(np.concatenate([p["J"] for p in parts]), h.get_J())
    assert np.array_equal(np.concatenate([p["pi"] for p in parts]), h.get_pi())
    for p in parts:
        assert np.allclose(p["st"], st[-1], rtol=1e-12)
    h.close()


_TRANSPORT_RANK = r"""
import contextlib, ctypes as C, io, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
from pyro_amd import configs, parallel
rank, world, case, port, out, overlap = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], int(sys.argv[6])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + port, rank=rank, world_size=world)
tc = parallel.staged_transport(dist, rank, world)       # host-staged halo exchange over gloo
sendrecv, max3 = tc.sendrecv, tc.max3


with contextlib.redirect_stdout(io.StringIO()):
    if case == "python-system":
        sys.path.insert(0, os.path.join(%(root)r, "tests"))
        from table_case import table_case
        cfg = table_case()
    else:
        cfg = configs.build(case.split(":", 1)[1] if case.startswith(("halo-mismatch:", "fb:")) else case)
if case.startswith("halo-mismatch:"):
    # one width for the whole grid: ranks that pass different widths must ALL be refused (no hang, no wrong rows)
    try:
        parallel.RcclValueIteration(cfg["grid_sys"], cfg["cf"], rank, world, dtype=cfg["dtype"], halo=2 + rank,
                                    transport=(sendrecv, max3))
    except Exception as e:
        assert "halo_rows differs" in str(e), e
        np.savez(out, refused=True)
        dist.destroy_process_group()
        print("TRANSPORT-RANK-OK", rank)
        sys.exit(0)
    raise SystemExit("a per-rank halo width was accepted")
vi = parallel.RcclValueIteration(cfg["grid_sys"], cfg["cf"], rank, world, dtype=cfg["dtype"], overlap=bool(overlap),
                                 transport=(sendrecv, max3), f32_feedback=case.startswith("fb:"))
st5, n5 = vi.run(5, 1.0, -1.0)
st, n = vi.run(400, 1.0, float(sys.argv[7]))
J, pi = vi.owned()
np.savez(out, J=J, pi=pi, st5=st5, st=st, n=n, rows=np.array(vi.rows), desc=vi.describe())
vi.close()
dist.destroy_process_group()
print("TRANSPORT-RANK-OK", rank)
"""


@pytest.mark.gpu
@pytest.mark.parametrize("case,world,overlap,tol", [("python-system", 3, True, 0.5),
                                                    ("cartpole:21,21,21,21:7:float32", 2, True, -1.0),
                                                    ("cartpole:21,21,21,21:7:float32", 3, True, 0.5),
                                                    # error-feedback storage: every piece of a slab keeps the residuals of its rows
                                                    ("fb:cartpole:21,21,21,21:7:float32", 3, True, -1.0),
                                                    ("pendulum:101,101:11:float64", 2, False, 0.5),
                                                    ("pendulum:101,101:11:float32", 4, True, -1.0),
                                                    ("halo-mismatch:pendulum:41,41:5:float32", 2, True, -1.0)])
def test_c_shard_schedule_with_caller_transport_on_one_gpu(tmp_path, case, world, overlap, tol):
    """The slab schedule of pvi_shard_sweep itself -- partition, boundary / interior handles over shared buffers, the
    two streams and events, which rows go to which neighbour, the stop test on all-reduced statistics -- with 2-4 ranks
    sharing the one GPU of the test box.  RCCL refuses several ranks per GPU, so the two inter-rank steps go through
    pvi_shard_create_with_transport (host-staged over gloo); everything else is the code the RCCL path runs.  The
    concatenated slabs must equal the whole-grid handle bit for bit, and every rank must stop at the same sweep."""
    import subprocess
    import sys as _sys
    from pyro_amd import configs
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "transport_rank.py"
    script.write_text(_TRANSPORT_RANK % dict(root=root))
    port = str(29600 + (os.getpid() + world * 7 + int(overlap)) % 300)
    procs = [subprocess.Popen([_sys.executable, str(script), str(r), str(world), case, port, str(tmp_path / ("t%d.npz" % r)),
                               str(int(overlap)), repr(tol)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    for r, p in enumerate(procs):
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0 and ("TRANSPORT-RANK-OK %d" % r) in o, o[-3000:]
    parts = [np.load(tmp_path / ("t%d.npz" % r)) for r in range(world)]
    if case.startswith("halo-mismatch:"):
        assert all(bool(p["refused"]) for p in parts)
        return
    assert "comm=caller" in str(parts[0]["desc"]) and ("+overlap" in str(parts[0]["desc"])) == bool(overlap)
    if case == "python-system":
        # an arbitrary Python system (table tier): every rank built the look-up tables of its own rows; the sharded
        # sweeps must equal the single-GPU table tier
        from pyro_amd.planning import dynamicprogramming
        from table_case import table_case
        with contextlib.redirect_stdout(io.StringIO()):
            cfg = table_case()
            dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"], dtype=cfg["dtype"])
        assert dp.tier == "table" and "table" in str(parts[0]["desc"])
        h = dp._p
    else:
        from pyro_amd import _native
        with contextlib.redirect_stdout(io.StringIO()):
            cfg = configs.build(case[3:] if case.startswith("fb:") else case)
            h = cfg["grid_sys"]._device_problem(cost=cfg["cf"].device_cost(), dtype=cfg["dtype"],
                                                flags=_native.FLAG_F32_FEEDBACK if case.startswith("fb:") else 0)
        h.terminal_cost()
        if case.startswith("fb:"):
            assert "feedback=1" in h.describe()
    st5, _ = h.sweep(5, 1.0, -1.0)
    st, n = h.sweep(400, 1.0, tol)
    assert np.array_equal(np.concatenate([p["J"] for p in parts]), h.get_J())
    assert np.array_equal(np.concatenate([p["pi"] for p in parts]), h.get_pi())
    for p in parts:
        assert int(p["n"]) == n and np.allclose(p["st"], st[-1], rtol=1e-12) and np.allclose(p["st5"], st5[-1], rtol=1e-12)
    h.close()


_SURFACE_RANK = r"""
import contextlib, io, os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, %(root)r)
from pyro_amd import configs, parallel
from pyro_amd.planning import dynamicprogramming
rank, world, port, out, dtype = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + port, rank=rank, world_size=world)
with contextlib.redirect_stdout(io.StringIO()) as log:
    cfg = configs.build("c1")
    dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"], dtype=dtype,
                                                              comm=parallel.staged_transport(dist, rank, world))
    dp.save_time_history = False
    assert dp.sharded and dp.tier == "fused" and "comm=caller" in dp._p.describe()
    dp.compute_steps(3)
    dp.solve_bellman_equation(tol=0.1)
    J, pi, Jn, k = dp.J.copy(), dp.pi.copy(), dp.J_next.copy(), dp.k
    u = dp.get_lookup_table_controller().c(np.array([-1.0, 0.5]), None)
    dp.clean_infeasible_set()
    Jc, pic = dp.J.copy(), dp.pi.copy()
    dp.verbose = False
    dp.stats_every_sweep = False
    dp.compute_steps(2)                     # the cleaned J went back to every slab; quiet batch of two
    J2, k2 = dp.J.copy(), dp.k
    dp.save_latest(out + "_latest")
np.savez(out, J=J, pi=pi, Jn=Jn, k=k, u=u, Jc=Jc, pic=pic, J2=J2, k2=k2, lines=log.getvalue().count(" max: "))
dp._p.close()
dist.destroy_process_group()
print("SURFACE-RANK-OK", rank)
"""


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_sharded_class_surface_solves_config1(tmp_path, dtype):
    """VERDICT r2 row b': DynamicProgrammingWithLookUpTable(grid_sys, cf, comm=...) over two ranks (the library's slab
    schedule, host-staged transport because both ranks share the one GPU): compute_steps, solve_bellman_equation, J, pi,
    J_next, clean_infeasible_set, get_lookup_table_controller and save_latest as on one GPU -- config 1 stops after the
    reference's 618 sweeps with the reference's J* and pi* (float64; float32 within 1e-5 and the same sweep count as the
    one-GPU float32 solve)."""
    import subprocess
    import sys as _sys
    from pyro_amd import configs
    from pyro_amd.planning import dynamicprogramming
    g = load("config1_pendulum_101x101x11")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "surface_rank.py"
    script.write_text(_SURFACE_RANK % dict(root=root))
    port = str(29300 + (os.getpid() + (7 if dtype == "float32" else 0)) % 300)
    procs = [subprocess.Popen([_sys.executable, str(script), str(r), "2", port, str(tmp_path / ("s%d.npz" % r)), dtype],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    for r, p in enumerate(procs):
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0 and ("SURFACE-RANK-OK %d" % r) in o, o[-3000:]
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build("c1")
        one = dynamicprogramming.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"], dtype=dtype)
        one.save_time_history = False
        one.solve_bellman_equation(tol=0.1)
        J1, pi1, Jn1, k1 = one.J.copy(), one.pi.copy(), one.J_next.copy(), one.k
        u1 = one.get_lookup_table_controller().c(np.array([-1.0, 0.5]), None)
        one.clean_infeasible_set()
        Jc1, pic1 = one.J.copy(), one.pi.copy()
        one.compute_steps(2)
    for r in range(2):
        p = np.load(tmp_path / ("s%d.npz" % r))
        # sharded == one GPU, bit for bit, in either dtype
        assert int(p["k"]) == k1 and int(p["k2"]) == k1 + 2 and int(p["lines"]) == k1
        assert np.array_equal(p["J"], J1) and np.array_equal(p["pi"], pi1) and np.array_equal(p["Jn"], Jn1)
        assert np.array_equal(p["Jc"], Jc1) and np.array_equal(p["pic"], pic1) and np.array_equal(p["u"], u1)
        assert np.array_equal(p["J2"], one.J)
        assert np.array_equal(np.load(str(tmp_path / ("s%d.npz" % r)) + "_latest_J_inf.npy"), one.J_next)
        # ... and the reference's result
        if dtype == "float64":
            assert int(p["k"]) == int(g["sweeps"]) == 618
            assert relerr(p["J"], g["J"]) < 1e-12 and np.array_equal(p["pi"], g["pi"].astype(np.int64))
        else:
            assert relerr(p["J"], g["J"]) <= REL_F32


@pytest.mark.gpu
def test_remaining_dp_demos_run_fused():
    """2D_navigation.py (point robot with obstacles), car_braking.py (longitudinal car: its isavalidinput rejects inputs
    that lift an axle, tested per cell in-kernel) and pendulum_reachability.py (Reachability cost) through the class
    surface: fused tier, the reference's own tables and results."""
    from pyro_amd.analysis import costfunction
    from pyro_amd.dynamic import pendulum, vehicle_propulsion, vehicle_steering
    from pyro_amd.planning import discretizer, dynamicprogramming
    with contextlib.redirect_stdout(io.StringIO()):
        s = vehicle_steering.HolonomicMobileRobotwithObstacles()
        gs = discretizer.GridDynamicSystem(s, [21, 21], [3, 3])
        q = costfunction.QuadraticCostFunction.from_sys(s)
        q.xbar = np.array([10.0, 0.0]); q.R[0, 0] = 0.0; q.R[1, 1] = 0.0; q.S[0, 0] = 10.0; q.S[1, 1] = 10.0; q.INF = 8000
        g = load("obstacles_21x21x3x3")
        for cls, key in ((dynamicprogramming.DynamicProgrammingWithLookUpTable, "J_5"), (dynamicprogramming.DynamicProgramming, "Jbase_5")):
            dp = cls(gs, q)
            assert dp.tier == "fused"
            dp.compute_steps(5)
            assert relerr(dp.J, g[key]) < 1e-12 and np.array_equal(dp.pi, g[key.replace("J", "pi")])
        assert int(g["differs"]) > 0                       # the two classes really differ on this case
        assert np.array_equal(gs.x_next_table, g["x_next_table"]) and np.array_equal(gs.x_next_isok, g["x_next_isok"])

        s = vehicle_propulsion.LongitudinalFrontWheelDriveCarWithWheelSlipInput()
        s.x_ub[1] = 15; s.x_lb[1] = 0; s.yc = 1.5
        gs = discretizer.GridDynamicSystem(s, [31, 31], [7], 0.05)
        q = costfunction.QuadraticCostFunction.from_sys(s)
        q.xbar = np.array([45, 0]); q.Q[0, 0] = 0.1; q.Q[1, 1] = 0.1; q.INF = 1000000
        g = load("longcar_31x31x7")
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(gs, q)
        assert dp.tier == "fused"
        dp.compute_steps(5)
        assert relerr(dp.J, g["J_5"]) < 1e-12 and np.array_equal(dp.pi, g["pi_5"])
        assert np.array_equal(gs.x_next_table, g["x_next_table"])
        xn, xok, aok, G = dp._p.build_tables()
        assert np.array_equal(aok, g["action_isok"]) and not aok.all() and np.array_equal(xok, g["x_next_isok"])
        np.testing.assert_allclose(G, g["G"], rtol=1e-14)
        db = dynamicprogramming.DynamicProgramming(gs, q, dtype="float32")
        db.compute_steps(5)
        assert relerr(db.J, g["Jbase_5"]) <= REL_F32

        s = pendulum.SinglePendulum()
        s.xbar = np.array([-3.14, 0])
        gs = discretizer.GridDynamicSystem(s, [41, 41], [3])
        cf = costfunction.Reachability(s.isavalidstate, s.xbar)
        g = load("reachability_41x41x3")
        for dtype, tol in (("float64", 1e-12), ("float32", REL_F32)):
            dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(gs, cf, dtype=dtype)
            assert dp.tier == "fused"
            assert np.array_equal(dp.J, g["J0"])
            dp.compute_steps(20)
            assert relerr(dp.J, g["J_20"]) <= tol
        # a custom target test is arbitrary Python: table tier
        cf2 = costfunction.Reachability(s.isavalidstate, s.xbar, isontarget=lambda x, t=0: bool(abs(x[0] + 3.14) < 0.3))
        assert dynamicprogramming.DynamicProgrammingWithLookUpTable(gs, cf2).tier == "table"


@pytest.mark.gpu
def test_sanitized_host_build_runs_clean():
    """the host side of the C ABI under a sanitizer.  tools/abi_tour.py (every handle kind, sweeps with a stop,
    up/downloads, tables, packed tables, spline mode, rollouts, obstacle systems, self check, one-rank shards with and
    without RCCL, error paths) runs against pyro_amd/libpyrovi_ubsan.so (UBSan + bounds, every report fatal) preloaded
    under python: it must finish and report nothing."""
    import subprocess
    import sys as _sys
    from pyro_amd import _build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = _build.build_sanitized(verbose=False)        # no-op when newer than every source file, rebuilt when stale
    env = dict(os.environ, LD_PRELOAD=_build.sanitizer_runtime(), PYROVI_LIB=lib,
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([_sys.executable, os.path.join(root, "tools", "abi_tour.py")], env=env, capture_output=True, text=True,
                       timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "ABI-TOUR-OK" in out and "runtime error" not in out, out[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["pendulum:201,201:21:float32", "cartpole:21,21,21,21:7:float32", "twolink:9,9,9,9:3,3:float64",
                                  "pendulum:101,101:11:float64"])
def test_self_check_production_path_against_plain_gather(case):
    """pvi_self_check: one backup by the handle's production path (LDS windows, set-up tables, float32 displacement / the
    float64 second form) and by the plain-gather kernel, compared on the device.  float64: identical; float32: within the
    stated tolerance, few policy differences (near-ties)."""
    from pyro_amd import configs
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build(case)
        h = cfg["grid_sys"]._device_problem(cost=cfg["cf"].device_cost(), dtype=cfg["dtype"])
    h.terminal_cost()
    h.sweep(6, 1.0, -1.0)
    J = h.get_J()
    d, n = h.self_check(1.0)
    assert np.array_equal(h.get_J(), J)                     # the current cost-to-go is left alone
    if cfg["dtype"] == "float64":
        assert d == 0.0 and n == 0
    else:
        assert d <= 2e-6 and n <= 0.02 * J.size, (d, n)
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("script,args,expect", [
    ("pendulum_optimal_swingup.py", [], "u(x0)"),
    ("mountain_car.py", [], None),
    ("helicopter_tunnel.py", ["31", "float32"], "path=fast3"),
    ("helicopter_tunnel.py", ["21"], "u(x0)"),
    ("policy_evaluation_computed_torque.py", ["101"], "gpu: controller + dynamics + cost"),
])
def test_example_scripts_run(script, args, expect):
    """examples/ are the reference's demo scripts with the imports switched: each one runs to the end on the GPU."""
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MPLBACKEND="Agg")
    r = subprocess.run([_sys.executable, os.path.join(root, "examples", script)] + args, capture_output=True, text=True,
                       timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    if expect:
        assert expect in r.stdout, r.stdout[-2000:]


@pytest.mark.gpu
def test_example_sharded_script_runs_on_one_rank():
    """examples/cartpole_sharded.py under torch.distributed.run with one rank: the class surface over an in-library RCCL
    communicator, gathers, controller and save_latest."""
    import subprocess
    import sys as _sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, GRID="15", MASTER_ADDR="127.0.0.1")
        port = str(29700 + os.getpid() % 200)
        r = subprocess.run([_sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                            "127.0.0.1", "--master-port", port, os.path.join(root, "examples", "cartpole_sharded.py")],
                           capture_output=True, text=True, timeout=900, env=env, cwd=tmp)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert "u(hanging, at rest)" in r.stdout and "comm=rccl" in r.stdout, r.stdout[-2000:]


@pytest.mark.gpu
def test_h3_full_size_fast_sweep_against_the_plain_kernel(variants):
    """The n = 3 bench workload (helicopter tunnel 201 x 201 x 401 x 9, float32) at FULL size: the mask-walking production
    sweep k_sweep3_fast against k_sweep3 -- the operation-for-operation kernel that the reference's goldens pin at small sizes
    (test_3d_systems_fused_kernels_match_reference) -- on the whole grid after 1, 5 and 30 sweeps; policies differ on ties
    and last-bit flips only; obstacle nodes cost the same INF in both."""
    from pyro_amd import configs
    from pyro_amd.planning import dynamicprogramming
    with contextlib.redirect_stdout(io.StringIO()):
        c = configs.build("h3")
    outs = {}
    # "fast3": every XCD sweeps its eighth of axis 1 (201 = 8 x 25 + 1: ragged eighths); "fast3_plain_order": blocks in node order
    for tag, env in (("fast3", {}), ("fast3_plain_order", {"PVI_XCD3": "0"}), ("plain", {"PVI_NO_FAST": "1"})):
        variants.delenv("PVI_NO_FAST")
        variants.delenv("PVI_XCD3")
        for k, v in env.items():
            variants.setenv(k, v)
        with contextlib.redirect_stdout(io.StringIO()):
            dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(c["grid_sys"], c["cf"], dtype="float32")
        dp.save_time_history = False
        desc = dp._p.describe()
        assert desc.startswith("path=fast3" if tag.startswith("fast3") else "path=exact-f32"), desc
        res, done = [], 0
        for k in (1, 5, 30):
            dp._p.sweep(k - done, 1.0, -1.0)
            done = k
            res.append((dp._p.get_J(), dp._p.get_pi()))
        outs[tag] = res
        dp._p.close()
    variants.delenv("PVI_NO_FAST")
    variants.delenv("PVI_XCD3")
    for (Ja, pa), (Jb, pb) in zip(outs["fast3"], outs["fast3_plain_order"]):       # the launch order changes no bit
        assert np.array_equal(Ja, Jb) and np.array_equal(pa, pb)
    for (Jf, pf), (Jp, pp_) in zip(outs["fast3"], outs["plain"]):
        scale = np.abs(Jp).max()
        assert np.abs(Jf - Jp).max() <= 2e-6 * scale
        assert (pf != pp_).mean() < 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("name,path", [
    ("cartpole:13,11,15,14:300:float32", "lean"),      # 4-D LDS-window sweep, two-byte policy
    ("cartpole:9,9,9,9:300:float64", "exact-f64v2"),
    ("pendulum:61,57:300:float32", "lean"),            # 2-D
    ("pendulum:61,57:300:float64", "exact-f64v2"),
    ("twolink:7,8,9,10:17,16:float32", None),          # 272 torque pairs
])
def test_action_sets_beyond_one_byte(name, path):
    """A > 255 actions: pi is stored in two bytes on the device (pvi_pi_itemsize; the sweeps' `unsigned short`
    instantiations) -- J and the first-argmin policy against the oracle's C twin on the whole grid after 1, 2 and 6 sweeps."""
    from oracle import c_oracle as CO
    cfg, p = _full_problem(name)
    g = cfg["grid_sys"]
    assert g.actions_n > 255
    with contextlib.redirect_stdout(io.StringIO()):
        h = g._device_problem(cost=cfg["cf"].device_cost(), dtype=cfg["dtype"])
    desc = h.describe()
    if path:
        assert desc.startswith("path=" + path), desc
    f32 = cfg["dtype"] == "float32"
    c = CO.CProblem(p)
    h.terminal_cost()
    N = p.nodes_n
    done = 0
    for k in (1, 2, 6):
        if k - 1 - done > 0:
            h.sweep(k - 1 - done, 1.0, -1.0)
        Jk = h.get_J()
        h.sweep(1, 1.0, -1.0)
        done = k
        Jk1, pik1 = h.get_J(), h.get_pi()
        assert pik1.max() > 255 or k == 1, (name, k, pik1.max())       # the high actions do get chosen
        Jo, pio = c.sweep(Jk, 1.0, 0, N, f32=f32)
        scale = max(np.abs(Jo).max(), 1e-300)
        assert np.abs(Jk1 - Jo).max() <= (REL_F32 if f32 else 1e-12) * scale, (name, k, desc)
        if not f32:
            assert np.array_equal(pik1, pio), (name, k)
        else:
            nodes = np.arange(N)
            q, qmin = c.q_at(Jk, nodes, pik1)
            assert (q - qmin).max() <= 1e-5 * scale, (name, k)
    h.close()
