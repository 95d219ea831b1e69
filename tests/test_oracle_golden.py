"""
Pins the CPU oracle (oracle/vi_oracle.py) to golden vectors captured from the reference itself
(tests/golden/make_goldens.py).  CPU only.

Tolerances: the oracle reproduces the reference's float64 arithmetic except for (i) BLAS dot
products that use FMA inside OpenBLAS and (ii) LAPACK's 2x2 inverse -- both differ in the last
ulp, so J is compared at 1e-12 relative; classification masks and pi must be identical.
"""
import os

import numpy as np
import pytest
from scipy.interpolate import RegularGridInterpolator

from conftest import GOLDEN
from oracle import vi_oracle as O


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def problem_from(g, dyn_id, consts):
    lv = O.make_levels(g["x_lb"], g["x_ub"], g["dims"])
    ul = O.make_levels(g["u_lb"], g["u_ub"], g["udims"])
    return O.Problem(lv, ul, float(g["dt"]), dyn_id, consts, g["Q"], g["R"], g["S"], g["xbar"], g["ubar"],
                     float(g["INF"]), float(g["EPS"]))


TWOLINK = dict(l1=0.5, lc1=0.2, lc2=0.1, m1=1, I1=0, m2=1, I2=0, gravity=9.81, d1=0.5, d2=0.5)
DOUBLEP = dict(l1=1, lc1=1, lc2=1, m1=1, I1=0, m2=1, I2=0, gravity=9.81, d1=0, d2=0)


# --------------------------------------------------------------------------------------------- f
@pytest.mark.parametrize("key,dyn,consts,tol", [
    ("pendulum", O.DYN_PENDULUM, O.pendulum_consts(), 0.0),
    ("inverted", O.DYN_PENDULUM, O.pendulum_consts(inverted=True), 0.0),
    ("cartpole", O.DYN_CARTPOLE, O.cartpole_consts(), 1e-13),
    ("twolink", O.DYN_TWOLINK, O.twolink_consts(**TWOLINK), 1e-12),
    ("doublependulum", O.DYN_TWOLINK, O.twolink_consts(**DOUBLEP), 1e-12),
])
def test_f_kat(key, dyn, consts, tol):
    g = load("f_kat")
    dX = O.f_batch(dyn, consts, g[key + "_X"], g[key + "_U"])
    ref = g[key + "_dX"]
    err = np.abs(dX - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() <= tol


def test_f_spot_values():
    g = load("f_kat")
    assert np.array_equal(O.f_batch(O.DYN_PENDULUM, O.pendulum_consts(), [[0.3, 1.2]], [[0.7]])[0], g["spot_pendulum"])
    x = np.linspace(0.3, 1.2, 4)[None]
    np.testing.assert_allclose(O.f_batch(O.DYN_CARTPOLE, O.cartpole_consts(), x, [[0.7]])[0], g["spot_cartpole"], rtol=1e-13)
    np.testing.assert_allclose(O.f_batch(O.DYN_TWOLINK, O.twolink_consts(**TWOLINK), x, [[0.7, -0.4]])[0],
                               g["spot_twolink"], rtol=1e-12)
    # the literal values quoted in SURVEY A.1
    assert g["spot_pendulum"][1] == -1.0995266136738704


# ------------------------------------------------------------------------------------------ cost
def test_cost_kat():
    g = load("cost_kat")
    X, U = g["X"], g["U"]
    dx = tuple(X[:, d] - g["xbar"][d] for d in range(4))
    du = tuple(U[:, k] - g["ubar"][k] for k in range(2))
    on = O.l2norm(dx) < float(g["EPS"])
    gg = np.where(on, 0.0, O.quad_form(g["Q"], dx) + O.quad_form(g["R"], du))
    hh = np.where(on, 0.0, O.quad_form(g["S"], dx))
    np.testing.assert_allclose(gg, g["g"], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(hh, g["h"], rtol=1e-13, atol=1e-13)
    assert on[:9].all() and (g["g"][:9] == 0).all()
    assert np.array_equal(np.where(on, 0.0, 1.0), g["time_g"])


# ------------------------------------------------------------------------------------------ grid
@pytest.mark.parametrize("n,m", [(2, 1), (2, 2), (3, 1), (3, 2), (4, 1), (4, 2)])
def test_grid_kat(n, m):
    g = load("grid_kat")
    k = "n%dm%d_" % (n, m)
    b = g[k + "bounds"]
    dims, udims = g[k + "dims"], g[k + "udims"]
    lv = O.make_levels(b[:n], b[n:2 * n], dims)
    ul = O.make_levels(b[2 * n:2 * n + m], b[2 * n + m:], udims)
    for d in range(n):
        assert np.array_equal(lv[d], g[k + "x_level%d" % d])
    N = int(np.prod(dims))
    assert np.array_equal(O.coords_from_ids(np.arange(N), lv), g[k + "state_from_node_id"])
    assert np.array_equal(O.multi_index(np.arange(N), dims), g[k + "index_from_node_id"])
    assert np.array_equal(np.arange(N).reshape(dims), g[k + "node_id_from_index"])
    A = int(np.prod(udims))
    assert np.array_equal(O.coords_from_ids(np.arange(A), ul), g[k + "input_from_action_id"])
    assert np.array_equal(np.arange(A).reshape(udims), g[k + "action_id_from_index"])


# --------------------------------------------------------------------------------- interpolation
@pytest.mark.parametrize("dims", [(7, 9), (5, 6, 7), (5, 6, 7, 8)])
def test_interp_matches_scipy_bitwise(dims):
    rng = np.random.default_rng(0)
    lv = [np.linspace(-rng.uniform(1, 2), rng.uniform(1, 2), d) for d in dims]
    V = rng.normal(size=dims)
    P = np.stack([rng.uniform(l[0] * 1.1, l[-1] * 1.1, 5000) for l in lv], -1)
    for d in range(len(dims)):
        P[:200, d] = lv[d][rng.integers(0, dims[d], 200)]          # exactly on nodes
    P[200:220] = [l[-1] for l in lv]
    P[220:240] = [l[0] for l in lv]
    ref = RegularGridInterpolator(tuple(lv), V, "linear", False, 0)(P)
    assert np.array_equal(ref, O.interp_nlinear(lv, V, P))


# ---------------------------------------------------------------------------------- pendulum VI
def test_pendulum_small_tables_and_sweeps():
    g = load("pendulum_21x21x5")
    p = problem_from(g, O.DYN_PENDULUM, O.pendulum_consts())
    xn, xok, aok, G = O.cells(p, np.arange(p.nodes_n))
    assert np.array_equal(xn, g["x_next_table"])                  # bitwise
    assert np.array_equal(xok, g["x_next_isok"]) and np.array_equal(aok, g["action_isok"])
    np.testing.assert_allclose(G, g["G"], rtol=1e-14, atol=1e-14)
    J = O.terminal_cost(p)
    assert np.array_equal(J, g["J0"])
    for k in range(1, 11):
        Jn, pi = O.sweep(p, J)
        st, delta = O.sweep_stats(Jn, J)
        np.testing.assert_allclose(list(st) + [delta], g["stats"][k - 1], rtol=1e-13, atol=1e-13)
        J = Jn
        if k in (1, 2, 10):
            np.testing.assert_allclose(J, g["J_%d" % k], rtol=1e-13, atol=1e-13)
            assert np.array_equal(pi, g["pi_%d" % k])
            # base-class (cell-by-cell) sweep is identical for box validity
            np.testing.assert_allclose(J, g["Jbase_%d" % k], rtol=1e-13, atol=1e-13)
            assert np.array_equal(pi, g["pibase_%d" % k])
    # table-driven form on the reference's own tables
    J2, pi2, _ = O.sweep_lut(p.levels, g["x_next_table"], g["G"], g["J0"])
    assert np.array_equal(J2, g["J_1"]) and np.array_equal(pi2, g["pi_1"])
    # clean_infeasible_set + controller
    Jc, pic = O.clean_infeasible_set(p, J, pi, np.zeros(1))
    np.testing.assert_allclose(Jc, g["J_clean"], rtol=1e-13)
    assert np.array_equal(pic, g["pi_clean"])
    assert np.array_equal(O.policy_inputs(p, pic, 0), g["u0_from_policy"])
    np.testing.assert_allclose(O.controller_c(p, pic, g["ctl_x"]), g["ctl_u"], rtol=1e-13, atol=1e-13)


def test_config1_solve():
    """Config 1 (BASELINE.json configs[0]): 101x101x11 pendulum to tol 0.1 -> 618 sweeps."""
    g = load("config1_pendulum_101x101x11")
    p = problem_from(g, O.DYN_PENDULUM, O.pendulum_consts())
    J, pi, k = O.solve(p, tol=0.1)
    assert k == int(g["sweeps"]) == 618
    np.testing.assert_allclose(J, g["J"], rtol=1e-12, atol=1e-12)
    bad = pi != g["pi"]
    # exact ties / sub-ulp gaps may flip argmin: judge those nodes by Q-regret
    if bad.any():
        reg = O.q_regret(p, g["J_prev"], pi)
        assert reg[bad].max() < 1e-10
    assert bad.mean() < 1e-3
    # the survey's spot values
    assert J[0] == 300.0 and abs(J[5100] - 90.92910217938157) < 1e-10
    assert np.array_equal(np.bincount(g["pi"], minlength=11),
                          [2219, 362, 450, 604, 1044, 1900, 716, 615, 503, 669, 1119])


def test_lowdef_large_eps():
    g = load("pendulum_lowdef_41x21x3")
    p = problem_from(g, O.DYN_PENDULUM, O.pendulum_consts())
    _, _, _, G = O.cells(p, np.arange(p.nodes_n))
    np.testing.assert_allclose(G, g["G"], rtol=1e-14, atol=1e-14)
    assert (G == 0).sum() > 1                                      # several nodes inside the EPS ball
    J, pi, k = O.solve(p, tol=1.0)
    assert k == int(g["sweeps"])
    np.testing.assert_allclose(J, g["J"], rtol=1e-12, atol=1e-12)
    assert np.array_equal(pi, g["pi"])


def test_pendulum_demo_nonzero_terminal_cost():
    g = load("pendulum_demo_51x51x9")
    p = problem_from(g, O.DYN_PENDULUM, O.pendulum_consts())
    J = O.terminal_cost(p)
    np.testing.assert_allclose(J, g["J0"], rtol=1e-14)
    for _ in range(int(g["sweeps"])):
        J, pi = O.sweep(p, J)
    np.testing.assert_allclose(J, g["J"], rtol=1e-12, atol=1e-12)
    assert (pi != g["pi"]).mean() < 2e-3


# ------------------------------------------------------------------------------------- 4-D cases
def _check_4d(name, dyn, consts, sample=True, rtol=1e-11):
    g = load(name)
    p = problem_from(g, dyn, consts)
    if sample:
        ids = g["sample_ids"]
        xn, xok, _, G = O.cells(p, ids)
        ref = g["x_next_sample"]
        assert (np.abs(xn - ref) / np.maximum(1, np.abs(ref))).max() < 1e-12
        assert np.array_equal(xok, g["x_next_isok_sample"])
        np.testing.assert_allclose(G, g["G_sample"], rtol=1e-13, atol=1e-13)
    J = O.terminal_cost(p)
    np.testing.assert_allclose(J, g["J0"], rtol=1e-13, atol=1e-13)
    alpha = float(g["alpha"])
    Jprev = J
    for _ in range(int(g["sweeps"])):
        Jprev = J
        J, pi = O.sweep(p, J, alpha)
    ref = g["J"].astype(np.float64)
    np.testing.assert_allclose(J, ref, rtol=rtol, atol=rtol)
    bad = pi != g["pi"]
    if bad.any():
        assert O.q_regret(p, Jprev, g["pi"].astype(np.int64), alpha)[bad].max() < 1e-9
    return g, p, J, pi


def test_cartpole_small():
    _check_4d("cartpole_11p4x5", O.DYN_CARTPOLE, O.cartpole_consts())


def test_twolink_small():
    g, *_ = _check_4d("twolink_11p4x3x3", O.DYN_TWOLINK, O.twolink_consts(**TWOLINK))
    assert abs(float(g["isok_frac"]) - 0.073) < 0.01                # survey: 7.3 % of cells in bounds


def test_twolink_dt01():
    """dt = 0.01 two-link: 8 of 131 769 cells land within an ulp of a velocity bound (rational
    multiples of pi with sin(q) ~ 1e-16), where the last bit of LAPACK's 2x2 inverse decides the
    reference's classification.  Such cells are 'ulp-ambiguous' (the reference is not reproducible
    there across BLAS builds); everything else must agree, and with the reference's own mask
    substituted on those cells J must match to 1e-11."""
    g = load("twolink_11p4x3x3_dt01")
    p = problem_from(g, O.DYN_TWOLINK, O.twolink_consts(**TWOLINK))
    ids = np.arange(p.nodes_n)
    xn, xok, aok, G = O.cells(p, ids)
    ref_ok = np.unpackbits(g["x_next_isok_packed"])[:xok.size].astype(bool).reshape(xok.shape)
    mism = np.argwhere(xok != ref_ok)
    assert len(mism) <= 16
    for s, a in mism:
        gap = np.minimum(np.abs(xn[s, a] - p.x_lb), np.abs(xn[s, a] - p.x_ub)).min()
        assert gap <= 4 * np.spacing(2 * np.pi)
    G = np.where(ref_ok, np.where(xok, G, O.cells(p, ids)[3]), p.INF)
    # cells the reference calls valid but the oracle calls invalid need their g*dt back
    flip = ref_ok & ~xok
    if flip.any():
        dx = O.coords_from_ids(ids, p.levels) - p.xbar
        gx = np.einsum("si,ij,sj->s", dx, p.Q, dx)
        du = p.u_table - p.ubar
        gu = np.einsum("ai,ij,aj->a", du, p.R, du)
        G = np.where(flip, (gx[:, None] + gu[None, :]) * p.dt, G)
    J = O.terminal_cost(p)
    for _ in range(int(g["sweeps"])):
        J, pi, _ = O.sweep_lut(p.levels, xn, G, J, float(g["alpha"]))
    np.testing.assert_allclose(J, g["J"], rtol=1e-11, atol=1e-11)
    assert (pi != g["pi"]).mean() < 1e-3


def test_doublependulum_demo_weights():
    _check_4d("doublependulum_13x11x13x11x3x3", O.DYN_TWOLINK, O.twolink_consts(**DOUBLEP), sample=False)


def test_cartpole_mid_f32_golden():
    """21^4 x 7, 20 sweeps; the golden J is stored rounded to f32."""
    path = os.path.join(GOLDEN, "cartpole_21p4x7.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    _check_4d("cartpole_21p4x7", O.DYN_CARTPOLE, O.cartpole_consts(), sample=False, rtol=1e-6)


# ------------------------------------------------------------------------- table-driven goldens (tier B)
@pytest.mark.parametrize("name,sweeps", [("obstacles_21x21x3x3", 5), ("helicopter_11x11x11x5", 5), ("reachability_41x41x3", 20)])
def test_table_goldens_lut_and_base(name, sweeps):
    """Obstacle / domain-check / reachability cases: arbitrary sys.isavalidstate and cost -> the recursion on
    the reference's own tables, LUT semantics (INF + alpha*J) and base-class semantics (exactly INF)."""
    g = load(name)
    lv = O.make_levels(g["x_lb"], g["x_ub"], g["dims"])
    alpha = 0.999 if "helicopter" in name else 1.0
    ok = g["x_next_isok"] & g["action_isok"]
    J = Jb = g["J0"]
    for k in range(1, sweeps + 1):
        J, pi, _ = O.sweep_lut(lv, g["x_next_table"], g["G"], J, alpha)
        Jb, pib, _ = O.sweep_base(lv, g["x_next_table"], g["G"], ok, Jb, float(g["INF"]), alpha)
        if k in (1, sweeps):
            np.testing.assert_allclose(J, g["J_%d" % k], rtol=1e-13, atol=1e-13)
            assert np.array_equal(pi, g["pi_%d" % k])
            np.testing.assert_allclose(Jb, g["Jbase_%d" % k], rtol=1e-13, atol=1e-13)
            assert np.array_equal(pib, g["pibase_%d" % k])
    if "obstacles" in name:
        assert int(g["differs"]) > 0 and (J != Jb).sum() == int(g["differs"])   # the two semantics really differ


def test_policy_evaluator_golden():
    g = load("policy_eval_41x41")
    lv = O.make_levels(g["x_lb"], g["x_ub"], g["dims"])
    J = g["J0"]
    for k in range(1, 11):
        J, _, _ = O.sweep_lut(lv, g["x_next_table"][:, None, :], g["G"][:, None], J)
        if k in (1, 10):
            np.testing.assert_allclose(J, g["J_%d" % k], rtol=1e-13, atol=1e-13)


def test_rollout_golden():
    """Closed-loop Euler rollouts of the look-up-table policy against the reference's ctl + sys loop."""
    g = load("rollout_pendulum_21x21x5")
    p = problem_from(g, O.DYN_PENDULUM, O.pendulum_consts())
    X, U = O.rollout(p, g["pi"].astype(np.int64), g["X0"], int(g["npts"]), float(g["tf"]))
    np.testing.assert_allclose(U, g["U"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(X, g["X"], rtol=1e-9, atol=1e-9)
    assert (g["U"][2, 5:] == 0).all()          # the trajectory that leaves the grid gets u = 0 (fill value)


# ------------------------------------------------------------------------------- bicubic-spline class
def test_spline_restatement_matches_scipy():
    """oracle spline_* (FITPACK regrid/bispev restated) against scipy.interpolate.RectBivariateSpline itself."""
    from scipy.interpolate import RectBivariateSpline
    rng = np.random.default_rng(3)
    for n0, n1 in [(4, 4), (5, 7), (21, 17), (64, 33)]:
        x = np.linspace(-2.0, 3.0, n0)
        y = np.linspace(0.5, 9.0, n1)
        Z = rng.normal(size=(n0, n1))
        S = RectBivariateSpline(x, y, Z, bbox=[None, None, None, None], kx=3, ky=3)
        tx, ty, C = O.spline_fit([x, y], Z)
        assert np.array_equal(tx, S.get_knots()[0]) and np.array_equal(ty, S.get_knots()[1])
        assert np.abs(C.ravel() - S.get_coeffs()).max() < 1e-12
        P = rng.uniform([-4, -1], [5, 11], size=(500, 2))            # inside and outside the box (clamped)
        P[:4] = [[x[0], y[0]], [x[-1], y[-1]], [x[1], y[-2]], [x[-1], y[2]]]
        assert np.abs(O.spline_eval(tx, ty, C, P) - S(P[:, 0], P[:, 1], grid=False)).max() < 1e-12
    with pytest.raises(ValueError):
        O.spline_knots(np.linspace(0, 1, 3))


def test_spline_value_iteration_golden():
    """DynamicProgramming2DRectBivariateSpline run by the reference (dynamicprogramming.py:578-614)."""
    g = load("spline_pendulum")
    for tag, xd, ud, dt in (("a", (21, 21), (5,), 0.05), ("b", (41, 31), (7,), 0.1)):
        lv = O.make_levels(g["x_lb"], g["x_ub"], xd)
        ul = O.make_levels(g["u_lb"], g["u_ub"], ud)
        p = O.Problem(lv, ul, dt, O.DYN_PENDULUM, O.pendulum_consts(), g["Q"], g["R"], g["S"], g["xbar"], g["ubar"],
                      float(g["INF"]), float(g["EPS"]))
        xn, _, _, G = O.cells(p, np.arange(p.nodes_n))
        J = O.terminal_cost(p)
        assert np.array_equal(O.spline_knots(lv[0]), g[tag + "_knots_x"])
        for k in range(1, 9):
            if k in (1, 2, 8):
                C = O.spline_fit(lv, J.reshape(xd))[2]
                assert np.abs(C.ravel() - g["%s_coef_%d" % (tag, k)]).max() <= 1e-12 * max(1.0, np.abs(C).max())
            Jn, pi, _ = O.sweep_spline(lv, xn, G, J)
            st, delta = O.sweep_stats(Jn, J)
            assert np.allclose([*st, delta], g[tag + "_stats"][k - 1], rtol=1e-11)
            if k in (1, 2, 8):
                assert np.abs(Jn - g["%s_J_%d" % (tag, k)]).max() <= 1e-11 * np.abs(Jn).max()
                clear = g["%s_gap_%d" % (tag, k)] > 1e-8          # argmin decided by more than rounding
                assert np.array_equal(pi[clear], g["%s_pi_%d" % (tag, k)][clear])
                assert (pi != g["%s_pi_%d" % (tag, k)]).mean() < 0.02
            J = Jn


def test_cubic_interpolation_golden():
    """dp.interpol_method = 'cubic' / 'cubic_legacy' on 2-D grids, run by the reference (dynamicprogramming.py:186-189 ->
    discretizer.py:570-587 -> RegularGridInterpolator(method, bounds_error=False, fill_value=0)): the interpolating tensor spline of
    RectBivariateSpline(kx=ky=3) inside the grid box and ZERO outside.  'cubic_legacy' is SciPy's exact fit: 1e-11 after sweeps 1,
    2, 8 of the LUT class on two grids and 2 sweeps of the cell-by-cell base class.  'cubic' is the same spline fitted by an
    iterative solver at its default tolerance (SciPy >= 1.13): the reference's own two solves differ by more than 1e-7 of max J,
    and the exact spline is as close to 'cubic' as 'cubic_legacy' is."""
    g = load("cubic_pendulum")
    for tag in ("a", "b"):
        xd, ud, dt = tuple(int(d) for d in g[tag + "_dims"]), tuple(int(d) for d in g[tag + "_udims"]), float(g[tag + "_dt"])
        lv = O.make_levels(g[tag + "_x_lb"], g[tag + "_x_ub"], xd)
        ul = O.make_levels(g[tag + "_u_lb"], g[tag + "_u_ub"], ud)
        p = O.Problem(lv, ul, dt, O.DYN_PENDULUM, O.pendulum_consts(), g[tag + "_Q"], g[tag + "_R"], g[tag + "_S"], g[tag + "_xbar"],
                      g[tag + "_ubar"], float(g[tag + "_INF"]), float(g[tag + "_EPS"]))
        xn, x_ok, a_ok, G = O.cells(p, np.arange(p.nodes_n))
        assert np.array_equal(O.box_mask(lv, xn), x_ok)       # (a stock system: the grid box IS the valid set)
        J = O.terminal_cost(p)
        Jb = J.copy()
        for k in range(1, 9):
            Jn, pi, _ = O.sweep_lut(lv, xn, G, J, method="cubic")
            if k in (1, 2, 8):
                Jg = g["%s_J_%d" % (tag, k)]
                assert np.abs(Jn - Jg).max() <= 1e-11 * np.abs(Jg).max(), (tag, k)
                clear = g["%s_gap_%d" % (tag, k)] > 1e-8
                assert np.array_equal(pi[clear], g["%s_pi_%d" % (tag, k)][clear])
                assert (pi != g["%s_pi_%d" % (tag, k)]).mean() < 0.02
            if k in (2, 8):     # SciPy's iterative fit: a tolerance of the reference's environment, not of the restatement
                Ji, Jg = g["%s_J_%d_iter" % (tag, k)], g["%s_J_%d" % (tag, k)]
                tol_scipy = np.abs(Ji - Jg).max()
                assert 1e-7 * np.abs(Jg).max() < tol_scipy < 1e-3 * np.abs(Jg).max()
                assert np.abs(Jn - Ji).max() <= 1.0001 * tol_scipy + 1e-11 * np.abs(Jg).max()
            J = Jn
            if tag == "a" and k <= 2:
                Jb, pib, _ = O.sweep_base_cubic(lv, xn, G, x_ok & a_ok, Jb, float(g["a_INF"]))
        if tag == "a":
            assert np.abs(Jb - g["a_base_J_2"]).max() <= 1e-11 * np.abs(Jb).max()
            assert (pib != g["a_base_pi_2"]).mean() < 0.02
    # it is NOT the clamped spline of the RectBivariateSpline class, nor the linear interpolant
    Q2 = g["b_J_2"]
    Qc, Ql, Qq = O.sweep_spline(lv, xn, G, Q2)[2], O.sweep_lut(lv, xn, G, Q2)[2], O.sweep_lut(lv, xn, G, Q2, method="cubic")[2]
    assert np.abs(Qq - Qc).max() > 1e-3 and np.abs(Qq - Ql).max() > 1e-3


# --------------------------------------------------------------- three-dimensional systems (SURVEY 8 f3 remainder)
from cases3d import CASES3D, case3d  # noqa: E402


@pytest.mark.parametrize("name", CASES3D)
def test_3d_systems_tables_and_sweeps_match_reference(name):
    """Helicopter tunnel (obstacles + domain-check cost), car parking (obstacles, two inputs, tan of the steering
    angle), active suspension (ground-profile tables): x_next_table and the validity masks bit for bit, G to the last
    ulp of the BLAS dot products, J / pi of the look-up-table class after 1 and 5 sweeps."""
    g, p, alpha = case3d(name)
    ids = np.arange(p.nodes_n)
    xn, x_ok, a_ok, G = O.cells(p, ids)
    assert np.array_equal(xn, g["x_next_table"])
    assert np.array_equal(x_ok, g["x_next_isok"]) and np.array_equal(a_ok, g["action_isok"])
    np.testing.assert_allclose(G, g["G"], rtol=1e-14, atol=0)
    J = O.terminal_cost(p)
    np.testing.assert_allclose(J, g["J0"], rtol=1e-14, atol=0)
    for k in range(1, 6):
        J, pi = O.sweep(p, J, alpha)
        if k in (1, 5):
            np.testing.assert_allclose(J, g["J_%d" % k], rtol=1e-12, atol=1e-12)
            assert np.array_equal(pi, g["pi_%d" % k])
    if "kat_valid" in g.files:                      # isavalidstate at random points around the box
        X = g["kat_X"]
        valid = O.state_valid(p, tuple(X[:, d] for d in range(p.n)))
        assert np.array_equal(valid, g["kat_valid"])


def test_reachability_cost_restatement_matches_reference():
    """costfunction.Reachability on the pendulum (pendulum_reachability.py): G = 0 on valid cells | INF, J0 = 0 inside the
    target ball | INF, J after 1 and 20 sweeps of the look-up-table class."""
    g = load("reachability_41x41x3")
    lv = O.make_levels(g["x_lb"], g["x_ub"], g["dims"])
    ul = O.make_levels(g["u_lb"], g["u_ub"], g["udims"])
    p = O.Problem(lv, ul, float(g["dt"]), O.DYN_PENDULUM, O.pendulum_consts(), np.zeros((2, 2)), np.zeros((1, 1)),
                  np.zeros((2, 2)), g["xbar"], np.zeros(1), float(g["INF"]), float(g["EPS"]), reachability=True)
    xn, x_ok, a_ok, G = O.cells(p, np.arange(p.nodes_n))
    assert np.array_equal(xn, g["x_next_table"]) and np.array_equal(G, g["G"])
    J = O.terminal_cost(p)
    assert np.array_equal(J, g["J0"])
    for k in range(1, 21):
        J, pi = O.sweep(p, J)
        if k in (1, 20):
            np.testing.assert_allclose(J, g["J_%d" % k], rtol=1e-12, atol=1e-12)


def test_nearest_interpolation_matches_reference_golden():
    """dp.interpol_method = 'nearest' (RegularGridInterpolator method='nearest', discretizer.py:570-587): the oracle's
    restatement against the reference's own runs -- a 2-D pendulum for 1, 3 and 8 sweeps, a switch from linear to nearest
    between sweeps, and a 4-D cart-pole (scipy's generic path)."""
    g = load("nearest_pendulum_31x21x5")
    lv = O.make_levels(g["x_lb"], g["x_ub"], g["dims"])
    J = g["J0"].copy()
    for k in range(1, 9):
        J, pi, _ = O.sweep_lut(lv, g["x_next_table"], g["G"], J, method="nearest")
        if k in (1, 3, 8):
            assert np.array_equal(J, g["J_%d" % k]) and np.array_equal(pi, g["pi_%d" % k]), k
    J = g["J0"].copy()
    for k in range(5):
        J, pi, _ = O.sweep_lut(lv, g["x_next_table"], g["G"], J, method="linear" if k < 3 else "nearest")
    assert np.abs(J - g["Jmix_5"]).max() <= 1e-13 * np.abs(g["Jmix_5"]).max() and np.array_equal(pi, g["pimix_5"])


def test_slinear_is_the_linear_interpolant_in_the_reference_too():
    """dp.interpol_method = 'slinear' (VERDICT r5 missing #5: the reference forwards any method of RegularGridInterpolator,
    discretizer.py:570-587).  Order-1 spline = linear interpolation: the reference's own 12-sweep pendulum and 4-sweep cart-pole
    solves with the two methods agree to 2e-15 of max J with identical policies, and the oracle's closed-form sweep reproduces
    both at 1e-12 -- which is what lets the GPU build serve 'slinear' with its linear sweeps."""
    for name, key, n, dyn, consts in (("slinear_pendulum_31x21x5", "12", 12, O.DYN_PENDULUM, O.pendulum_consts()),
                                      ("slinear_cartpole_7x9x7x9x3", "4", 4, O.DYN_CARTPOLE, O.cartpole_consts())):
        g = load(name)
        Js, Jl = g["J%s_slinear" % key], g["J%s_linear" % key]
        assert np.abs(Js - Jl).max() <= 1e-13 * np.abs(Jl).max() and np.array_equal(g["pi%s_slinear" % key], g["pi%s_linear" % key])
        lv = O.make_levels(g["x_lb"], g["x_ub"], g["dims"])
        ul = O.make_levels(g["u_lb"], g["u_ub"], g["udims"])
        p = O.Problem(lv, ul, float(g["dt"]), dyn, consts, g["Q"], g["R"], g["S"], g["xbar"], g["ubar"], float(g["INF"]), float(g["EPS"]))
        J = O.terminal_cost(p)
        for _ in range(n):
            J, pi = O.sweep(p, J)
        assert np.abs(J - Js).max() <= 1e-12 * np.abs(Js).max()
        assert (pi != g["pi%s_slinear" % key]).mean() < 1e-3
