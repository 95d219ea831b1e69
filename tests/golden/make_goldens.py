#!/usr/bin/env python3
"""
Golden-vector generator (runs ONLY in the build container, where /root/reference exists).

Imports the reference implementation (SherbyRobotics/pyro, pure Python) and records inputs and
expected outputs of its grid value-iteration path as small .npz fixtures next to this file.
The fixtures are DATA (arrays + scalars); no reference source text is stored.

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py [case ...]

Cases follow SURVEY.md section 8(c).  Environment used for the committed fixtures:
Python 3.10.12, NumPy 2.2.6, SciPy 1.15.3.
"""
import contextlib
import io
import os
import sys
import time

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
REF = os.environ.get("PYRO_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


with quiet():
    from pyro.dynamic import pendulum, cartpole, manipulator, system  # noqa: E402
    from pyro.analysis import costfunction  # noqa: E402
    from pyro.planning import discretizer, dynamicprogramming  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %-40s %8.1f KiB" % (name + ".npz", os.path.getsize(path) / 1024.0))


# ---------------------------------------------------------------------------------------------
def case_f_kat():
    """64 uniform-random (x,u) in bounds per system -> dx  (SURVEY 8c.1)."""
    out = {}
    for key, cls in [("pendulum", pendulum.SinglePendulum), ("inverted", pendulum.InvertedPendulum),
                     ("cartpole", cartpole.CartPole), ("twolink", manipulator.TwoLinkManipulator),
                     ("doublependulum", pendulum.DoublePendulum)]:
        with quiet():
            s = cls()
        rng = np.random.default_rng(0)
        X = rng.uniform(s.x_lb, s.x_ub, size=(64, s.n))
        U = rng.uniform(s.u_lb, s.u_ub, size=(64, s.m))
        dX = np.array([s.f(X[i], U[i]) for i in range(64)])
        out[key + "_X"], out[key + "_U"], out[key + "_dX"] = X, U, dX
        out[key + "_bounds"] = np.concatenate([s.x_lb, s.x_ub, s.u_lb, s.u_ub])
    # the spot values quoted in SURVEY A.1
    with quiet():
        out["spot_pendulum"] = pendulum.SinglePendulum().f(np.array([0.3, 1.2]), np.array([0.7]))
        out["spot_cartpole"] = cartpole.CartPole().f(np.linspace(0.3, 1.2, 4), np.array([0.7]))
        out["spot_twolink"] = manipulator.TwoLinkManipulator().f(np.linspace(0.3, 1.2, 4), np.array([0.7, -0.4]))
    save("f_kat", **out)


def case_cost_kat():
    """Quadratic g/h incl. on-target zeroing and non-diagonal weights; Time; Reachability."""
    rng = np.random.default_rng(1)
    n, m = 4, 2
    q = costfunction.QuadraticCostFunction(n, m)
    A = rng.normal(size=(n, n)); q.Q = A @ A.T
    B = rng.normal(size=(m, m)); q.R = B @ B.T
    C = rng.normal(size=(n, n)); q.S = C @ C.T
    q.xbar = rng.normal(size=n); q.ubar = rng.normal(size=m)
    q.EPS = 0.25
    X = rng.normal(size=(48, n)); U = rng.normal(size=(48, m))
    X[:8] = q.xbar + 0.05 * rng.normal(size=(8, n))          # inside the on-target ball
    X[8] = q.xbar
    g = np.array([q.g(X[i], U[i], 0.0) for i in range(48)], dtype=float)
    h = np.array([q.h(X[i], 0.0) for i in range(48)], dtype=float)
    t = costfunction.TimeCostFunction(q.xbar.copy()); t.EPS = 0.25
    tg = np.array([t.g(X[i], U[i], 0.0) for i in range(48)], dtype=float)
    r = costfunction.Reachability(lambda x: bool(np.all(np.abs(x) < 1.5)), xbar=q.xbar.copy())
    rg = np.array([r.g(X[i], U[i], 0.0) for i in range(48)], dtype=float)
    rh = np.array([r.h(X[i], 0.0) for i in range(48)], dtype=float)
    save("cost_kat", Q=q.Q, R=q.R, S=q.S, xbar=q.xbar, ubar=q.ubar, EPS=q.EPS, INF=q.INF, X=X, U=U,
         g=g, h=h, time_g=tg, reach_g=rg, reach_h=rh, reach_INF=r.INF, reach_EPS=r.EPS)


def case_grid_kat():
    """x_level, node/action enumeration maps for n=2,3,4 and m=1,2."""
    out = {}
    for dims in [(5, 4), (3, 4, 5), (3, 4, 5, 6)]:
        for udims in [(3,), (2, 3)]:
            n, m = len(dims), len(udims)
            with quiet():
                s = system.ContinuousDynamicSystem(n, m, n)
            rng = np.random.default_rng(n * 10 + m)
            s.x_lb = -rng.uniform(1, 3, n); s.x_ub = rng.uniform(1, 3, n)
            s.u_lb = -rng.uniform(1, 3, m); s.u_ub = rng.uniform(1, 3, m)
            with quiet():
                g = discretizer.GridDynamicSystem(s, list(dims), list(udims), dt=0.1, lookup=False)
            k = "n%dm%d_" % (n, m)
            out[k + "bounds"] = np.concatenate([s.x_lb, s.x_ub, s.u_lb, s.u_ub])
            out[k + "dims"] = np.array(dims); out[k + "udims"] = np.array(udims)
            for d in range(n):
                out[k + "x_level%d" % d] = g.x_level[d]
            for d in range(m):
                out[k + "u_level%d" % d] = g.u_level[d]
            out[k + "state_from_node_id"] = g.state_from_node_id
            out[k + "index_from_node_id"] = g.index_from_node_id
            out[k + "node_id_from_index"] = g.node_id_from_index
            out[k + "input_from_action_id"] = g.input_from_action_id
            out[k + "index_from_action_id"] = g.index_from_action_id
            out[k + "action_id_from_index"] = g.action_id_from_index
            out[k + "x_step_size"] = g.x_step_size
            out[k + "u_step_size"] = g.u_step_size
            # index helpers on a few probe points (discretizer.py:453-537)
            P = rng.uniform(s.x_lb * 1.2, s.x_ub * 1.2, size=(16, n))
            out[k + "probe_x"] = P
            out[k + "probe_index"] = np.array([g.get_index_from_state(x) for x in P])
            out[k + "probe_nearest"] = np.array([g.get_nearest_index_from_state(x) for x in P])
            out[k + "probe_node"] = np.array([g.get_nearest_node_id_from_state(x) for x in P])
            PU = rng.uniform(s.u_lb * 1.2, s.u_ub * 1.2, size=(16, m))
            out[k + "probe_u"] = PU
            out[k + "probe_action"] = np.array([g.get_nearest_action_id_from_input(u) for u in PU])
    save("grid_kat", **out)


def _pendulum_problem(xdims, udims, dt=0.05, INF=300.0, xbar=(-3.14, 0.0), bounds=None, EPS=None, S=None, R=None):
    with quiet():
        s = pendulum.SinglePendulum()
        if bounds is not None:
            s.x_lb = np.array(bounds[0], dtype=float); s.x_ub = np.array(bounds[1], dtype=float)
        g = discretizer.GridDynamicSystem(s, list(xdims), list(udims), dt=dt)
        q = costfunction.QuadraticCostFunction.from_sys(s)
        q.xbar = np.array(xbar, dtype=float); q.INF = INF
        if EPS is not None:
            q.EPS = EPS
        if S is not None:
            q.S = np.array(S, dtype=float)
        if R is not None:
            q.R = np.array(R, dtype=float)
    return s, g, q


def _meta(s, g, q):
    return dict(x_lb=s.x_lb, x_ub=s.x_ub, u_lb=s.u_lb, u_ub=s.u_ub, dims=g.x_grid_dim, udims=g.u_grid_dim,
                dt=g.dt, Q=q.Q, R=q.R, S=q.S, xbar=q.xbar, ubar=q.ubar, INF=q.INF, EPS=q.EPS)


def case_pendulum_small():
    """Pendulum 21x21x5: tables, G, J0 and J/pi after 1, 2, 10 sweeps, LUT and base class (8c.4)."""
    s, g, q = _pendulum_problem((21, 21), (5,))
    out = _meta(s, g, q)
    with quiet():
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, q)
        dp.save_time_history = False
        out.update(x_next_table=g.x_next_table, x_next_isok=g.x_next_isok, action_isok=g.action_isok,
                   G=dp.G, J0=dp.J.copy())
        stats = []
        for k in range(1, 11):
            dp.initialize_backward_step(); dp.compute_backward_step()
            delta = dp.finalize_backward_step()
            d = dp.J - dp.J_next
            stats.append([dp.J.max(), d.max(), d.min(), delta])
            if k in (1, 2, 10):
                out["J_%d" % k] = dp.J.copy(); out["pi_%d" % k] = dp.pi.copy()
        out["stats"] = np.array(stats)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            db = dynamicprogramming.DynamicProgramming(g, q)
            db.save_time_history = False
            for k in range(1, 11):
                db.initialize_backward_step(); db.compute_backward_step(); db.finalize_backward_step()
                if k in (1, 2, 10):
                    out["Jbase_%d" % k] = db.J.copy(); out["pibase_%d" % k] = db.pi.copy()
        # clean_infeasible_set + controller samples (8c.9)
        dp.clean_infeasible_set()
        out["J_clean"] = dp.J.copy(); out["pi_clean"] = dp.pi.copy()
        ctl = dp.get_lookup_table_controller()
        rng = np.random.default_rng(9)
        P = rng.uniform(s.x_lb * 1.15, s.x_ub * 1.15, size=(32, 2))
        P[0] = s.x_lb; P[1] = s.x_ub; P[2] = [g.x_level[0][3], g.x_level[1][7]]
        out["ctl_x"] = P
        out["ctl_u"] = np.array([ctl.c(x, 0) for x in P])
        out["u0_from_policy"] = g.get_input_from_policy(dp.pi, 0)
    save("pendulum_21x21x5", **out)


def case_config1():
    """Config 1: pendulum 101x101x11 solved to tol 0.1 (8c.5)."""
    with quiet():
        s = pendulum.SinglePendulum()
        s.xbar = np.array([-3.14, 0.0])
        q = costfunction.QuadraticCostFunction.from_sys(s); q.INF = 300
        g = discretizer.GridDynamicSystem(s, [101, 101], [11])
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, q)
        dp.save_time_history = False
        t0 = time.time()
        dp.solve_bellman_equation()
        el = time.time() - t0
    out = _meta(s, g, q)
    out.update(J=dp.J, pi=dp.pi.astype(np.int16), sweeps=dp.k, J_prev=dp.J_next, ref_solve_seconds=el)
    save("config1_pendulum_101x101x11", **out)


def case_lowdef():
    """Pendulum 41x21x3, dt 0.2, EPS 1.0 to tol 1.0 (8c.6)."""
    s, g, q = _pendulum_problem((41, 21), (3,), dt=0.2, bounds=([-4.0, -5.0], [4.0, 5.0]), EPS=1.0)
    with quiet():
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, q)
        dp.save_time_history = False
        dp.solve_bellman_equation(tol=1.0)
    out = _meta(s, g, q)
    out.update(J=dp.J, pi=dp.pi.astype(np.int16), sweeps=dp.k, G=dp.G)
    save("pendulum_lowdef_41x21x3", **out)


def case_pendulum_demo():
    """Pendulum 51x51x9 with the swing-up demo's bounds/weights (S=10 I, INF 500): non-zero J0."""
    s, g, q = _pendulum_problem((51, 51), (9,), INF=500.0, bounds=([-10, -10], [10, 10]),
                                S=[[10.0, 0], [0, 10.0]])
    with quiet():
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, q)
        dp.save_time_history = False
        J0 = dp.J.copy()
        dp.compute_steps(30)
    out = _meta(s, g, q)
    out.update(J0=J0, J=dp.J, pi=dp.pi.astype(np.int16), sweeps=dp.k)
    save("pendulum_demo_51x51x9", **out)


def _run_4d(s, xdims, udims, dt, q, sweeps, keep=(), f32=False, alpha=1.0):
    with quiet():
        g = discretizer.GridDynamicSystem(s, list(xdims), list(udims), dt=dt)
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, q)
        dp.save_time_history = False
        dp.alpha = alpha
        out = _meta(s, g, q)
        out["alpha"] = alpha
        out["isok_frac"] = g.x_next_isok.mean()
        out["J0"] = dp.J.copy()
        for k in range(1, sweeps + 1):
            dp.initialize_backward_step(); dp.compute_backward_step(); dp.finalize_backward_step()
            if k in keep:
                out["J_%d" % k] = dp.J.copy(); out["pi_%d" % k] = dp.pi.astype(np.int16)
        out["J"] = dp.J.astype(np.float32) if f32 else dp.J
        out["pi"] = dp.pi.astype(np.int16)
        out["J_prev"] = dp.J_next.astype(np.float32) if f32 else dp.J_next
        out["sweeps"] = dp.k
    return g, dp, out


def case_cartpole_small():
    """Cart-pole 11^4 x 5, 5 sweeps, with a slice of the x_next table (8c.7)."""
    with quiet():
        s = cartpole.CartPole(); s.xbar = np.array([0, np.pi, 0, 0.0])
        q = costfunction.QuadraticCostFunction.from_sys(s)
    g, dp, out = _run_4d(s, (11,) * 4, (5,), 0.05, q, 5, keep=(1,))
    ids = np.arange(0, g.nodes_n, 37)
    out.update(sample_ids=ids, x_next_sample=g.x_next_table[ids], x_next_isok_sample=g.x_next_isok[ids],
               G_sample=dp.G[ids])
    save("cartpole_11p4x5", **out)


def case_cartpole_mid():
    """Cart-pole 21^4 x 7, 20 sweeps; J stored rounded to f32, full pi (8c.7)."""
    with quiet():
        s = cartpole.CartPole(); s.xbar = np.array([0, np.pi, 0, 0.0])
        q = costfunction.QuadraticCostFunction.from_sys(s)
    _, _, out = _run_4d(s, (21,) * 4, (7,), 0.05, q, 20, f32=True)
    out["J0"] = out["J0"].astype(np.float32)
    save("cartpole_21p4x7", **out)


def case_twolink_small():
    """Two-link manipulator 11^4 x 3x3, 5 sweeps (8c.7)."""
    with quiet():
        s = manipulator.TwoLinkManipulator()
        q = costfunction.QuadraticCostFunction.from_sys(s)
    g, dp, out = _run_4d(s, (11,) * 4, (3, 3), 0.05, q, 5, keep=(1,))
    ids = np.arange(0, g.nodes_n, 37)
    out.update(sample_ids=ids, x_next_sample=g.x_next_table[ids], x_next_isok_sample=g.x_next_isok[ids],
               G_sample=dp.G[ids])
    save("twolink_11p4x3x3", **out)


def case_twolink_dt():
    """Two-link 11^4 x 3x3 with dt = 0.01 (better-conditioned variant suggested by SURVEY 8d)."""
    with quiet():
        s = manipulator.TwoLinkManipulator()
        q = costfunction.QuadraticCostFunction.from_sys(s)
    g, _, out = _run_4d(s, (11,) * 4, (3, 3), 0.01, q, 5)
    # full validity mask (bit-packed): this case has cells whose x_next lands within an ulp of a
    # bound, where LAPACK's 2x2 inverse decides the classification (see test_twolink_dt01)
    out["x_next_isok_packed"] = np.packbits(g.x_next_isok.ravel())
    save("twolink_11p4x3x3_dt01", **out)


def case_doublependulum():
    """DoublePendulum 13x11x13x11 x 3x3 with the demo's bounds/weights, dt 0.1, EPS 1, alpha<1."""
    with quiet():
        s = pendulum.DoublePendulum()
        s.x_ub[0] = +0.5; s.x_lb[0] = -5.0; s.x_ub[1] = +4.0; s.x_lb[1] = -1.5
        s.x_ub[2] = +5.5; s.x_lb[2] = -4.0; s.x_ub[3] = +7.0; s.x_lb[3] = -4.0
        s.u_ub[0] = +12.0; s.u_lb[0] = -12.0; s.u_ub[1] = +12.0; s.u_lb[1] = -12.0
        q = costfunction.QuadraticCostFunction.from_sys(s)
        q.xbar = np.array([0, 0, 0, 0.0])
        q.Q[0, 0] = 1.0; q.Q[1, 1] = 0.5; q.Q[2, 2] = 0.1; q.Q[3, 3] = 0.05
        q.R[0, 0] = 0.05; q.R[1, 1] = 0.05
        q.INF = 1000; q.EPS = 1.0
    _, _, out = _run_4d(s, (13, 11, 13, 11), (3, 3), 0.1, q, 8, alpha=0.98)
    save("doublependulum_13x11x13x11x3x3", **out)


def _lut_and_base(g, q, sweeps, keep, alpha=1.0):
    """Tables + LUT-class and base-class J/pi after the sweeps in `keep`."""
    import warnings
    out = {}
    with quiet():
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, q)
        dp.save_time_history = False
        dp.alpha = alpha
        out.update(x_next_table=g.x_next_table, x_next_isok=g.x_next_isok, action_isok=g.action_isok, G=dp.G,
                   J0=dp.J.copy())
        for k in range(1, sweeps + 1):
            dp.initialize_backward_step(); dp.compute_backward_step(); dp.finalize_backward_step()
            if k in keep:
                out["J_%d" % k] = dp.J.copy(); out["pi_%d" % k] = dp.pi.astype(np.int16)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            db = dynamicprogramming.DynamicProgramming(g, q)
            db.save_time_history = False
            db.alpha = alpha
            for k in range(1, sweeps + 1):
                db.initialize_backward_step(); db.compute_backward_step(); db.finalize_backward_step()
                if k in keep:
                    out["Jbase_%d" % k] = db.J.copy(); out["pibase_%d" % k] = db.pi.astype(np.int16)
    return out


def case_obstacles():
    """HolonomicMobileRobotwithObstacles 21x21 x 3x3 (2D_navigation.py weights): isavalidstate with obstacles,
    where LUT (INF + alpha*J) and base class (INF) semantics differ."""
    from pyro.dynamic import vehicle_steering
    with quiet():
        s = vehicle_steering.HolonomicMobileRobotwithObstacles()
        g = discretizer.GridDynamicSystem(s, [21, 21], [3, 3])
        q = costfunction.QuadraticCostFunction.from_sys(s)
        q.xbar = np.array([10.0, 0.0])
        q.R[0, 0] = 0.0; q.R[1, 1] = 0.0; q.S[0, 0] = 10.0; q.S[1, 1] = 10.0
        q.INF = 8000
    out = _meta(s, g, q)
    out.update(_lut_and_base(g, q, 5, (1, 5)))
    out["differs"] = int((out["J_5"] != out["Jbase_5"]).sum())
    save("obstacles_21x21x3x3", **out)


def case_helicopter():
    """ConstantSpeedHelicopterTunnel 3-D 11x11x11 x 5 with QuadraticCostFunctionWithDomainCheck, alpha 0.999."""
    from pyro.dynamic import drone
    with quiet():
        s = drone.ConstantSpeedHelicopterTunnel()
        s.obstacles = [[(2, 2), (4, 4)], [(8, 5), (10, 10)], [(14, 0), (16, 4)]]
        s.mass = 0.1; s.vx = 5.0; s.width = 1.0
        s.x_ub = np.array([+60, 10, +20]); s.x_lb = np.array([-60, 0, +0])
        s.u_ub = np.array([+20]); s.u_lb = np.array([-20])
        g = discretizer.GridDynamicSystem(s, (11, 11, 11), [5], 0.05)
        q = costfunction.QuadraticCostFunctionWithDomainCheck.from_sys(s)
        q.xbar = np.array([0.0, 2.0, 20]); q.INF = 100000; q.EPS = 0.2
        q.Q[0, 0] = 2.0; q.Q[1, 1] = 200.0; q.Q[2, 2] = 0.0; q.R[0, 0] = 5.0
        q.S[0, 0] = 20.0; q.S[1, 1] = 50.0; q.S[2, 2] = 0.0
    out = _meta(s, g, q)
    out.update(_lut_and_base(g, q, 5, (1, 5), alpha=0.999))
    save("helicopter_11x11x11x5", **out)


def case_car():
    """KinematicCarModelwithObstacles 3-D 21x21x11 x 3x3 with the car_parking.py bounds / weights (plain quadratic
    cost, alpha 0.99): tables, LUT and base class; f / isavalidstate known answers."""
    from pyro.dynamic import vehicle_steering
    with quiet():
        s = vehicle_steering.KinematicCarModelwithObstacles()
        s.x_ub = np.array([+35, +3, +3]); s.x_lb = np.array([-5, -2, -3])
        s.u_ub = np.array([+3, +1]); s.u_lb = np.array([-3, -1])
        g = discretizer.GridDynamicSystem(s, (21, 21, 11), (3, 3), 0.1)
        q = costfunction.QuadraticCostFunction.from_sys(s)
        q.xbar = np.array([0, 0, 0]); q.INF = 1E8; q.EPS = 0.00
        q.R = np.array([[0.1, 0], [0, 0]])
    out = _meta(s, g, q)
    out.update(_lut_and_base(g, q, 5, (1, 5), alpha=0.99))
    out.update(_kat3(s))
    out.update(lenght=s.lenght, width=s.width, obstacles=np.array(s.obstacles, dtype=float).reshape(-1, 4))
    save("car_21x21x11x3x3", **out)


def case_suspension():
    """QuarterCarOnRoughTerrain 3-D 13x11x21 x 5 with the active_suspension.py parameters (alpha 0.99)."""
    from pyro.dynamic import suspension
    with quiet():
        s = suspension.QuarterCarOnRoughTerrain()
        s.mass = 0.5; s.b = 0.5; s.k = 8.0; s.vx = 10.0
        s.x_ub = np.array([+12, +1, +40]); s.x_lb = np.array([-12, -1, +0])
        s.u_ub = np.array([+40]); s.u_lb = np.array([-40])
        g = discretizer.GridDynamicSystem(s, (13, 11, 21), [5], 0.05)
        q = costfunction.QuadraticCostFunction.from_sys(s)
        q.xbar = np.array([0.0, 0.0, 20]); q.INF = 100000; q.EPS = 0.5
        q.Q[0, 0] = 2.0; q.Q[1, 1] = 5.0; q.Q[2, 2] = 0.0; q.R[0, 0] = 0.1
        q.S[0, 0] = 0.0; q.S[1, 1] = 0.0; q.S[2, 2] = 0.0
    out = _meta(s, g, q)
    out.update(_lut_and_base(g, q, 5, (1, 5), alpha=0.99))
    out.update(_kat3(s))
    out.update(params=np.array([s.mass, s.k, s.b, s.vx], dtype=float), a=s.a, w=s.w, phi=s.phi)
    save("suspension_13x11x21x5", **out)


def case_longcar():
    """LongitudinalFrontWheelDriveCarWithWheelSlipInput 31x31 x 7 with the car_braking.py bounds / weights: a
    state-dependent isavalidinput (negative wheel loads), tables, LUT and base class."""
    from pyro.dynamic import vehicle_propulsion
    with quiet():
        s = vehicle_propulsion.LongitudinalFrontWheelDriveCarWithWheelSlipInput()
        s.x_ub[1] = 15; s.x_lb[1] = 0
        s.yc = 1.5                  # a high centre of gravity: hard braking would lift the rear axle -> isavalidinput bites
        g = discretizer.GridDynamicSystem(s, [31, 31], [7], 0.05)
        q = costfunction.QuadraticCostFunction.from_sys(s)
        q.xbar = np.array([45, 0]); q.Q[0, 0] = 0.1; q.Q[1, 1] = 0.1; q.INF = 1000000
    out = _meta(s, g, q)
    out.update(_lut_and_base(g, q, 5, (1, 5)))
    rng = np.random.default_rng(4)
    X = rng.uniform(s.x_lb - 1.0, s.x_ub + 1.0, size=(64, 2)); U = rng.uniform(s.u_lb * 1.3, s.u_ub * 1.3, size=(64, 1))
    out.update(kat_X=X, kat_U=U, kat_dX=np.array([s.f(X[i], U[i]) for i in range(64)]),
               kat_uvalid=np.array([bool(s.isavalidinput(X[i], U[i])) for i in range(64)]),
               params=np.array([s.lenght, s.xc, s.yc, s.mass, s.gravity, s.rho, s.cdA, s.mu_max, s.mu_slope], dtype=float))
    save("longcar_31x31x7", **out)


def _kat3(s):
    """64 random (x, u) around the state box -> f, isavalidstate of a three-dimensional system."""
    rng = np.random.default_rng(3)
    X = rng.uniform(s.x_lb - 1.0, s.x_ub + 1.0, size=(64, s.n))
    U = rng.uniform(s.u_lb, s.u_ub, size=(64, s.m))
    return dict(kat_X=X, kat_U=U, kat_dX=np.array([s.f(X[i], U[i]) for i in range(64)]),
                kat_valid=np.array([bool(s.isavalidstate(X[i])) for i in range(64)]))


def case_reachability():
    """Reachability cost on the pendulum 41x41 x 3 (pendulum_reachability.py), 20 sweeps."""
    with quiet():
        s = pendulum.SinglePendulum()
        s.xbar = np.array([-3.14, 0])
        g = discretizer.GridDynamicSystem(s, [41, 41], [3])
        cf = costfunction.Reachability(s.isavalidstate, s.xbar)
    out = dict(x_lb=s.x_lb, x_ub=s.x_ub, u_lb=s.u_lb, u_ub=s.u_ub, dims=g.x_grid_dim, udims=g.u_grid_dim, dt=g.dt,
               xbar=s.xbar, INF=cf.INF, EPS=cf.EPS)
    out.update(_lut_and_base(g, cf, 20, (1, 20)))
    save("reachability_41x41x3", **out)


def case_policy_eval():
    """PolicyEvaluatorWithLookUpTable with a computed-torque controller on the pendulum 41x41 (A = 1)."""
    from pyro.control import nonlinear
    with quiet():
        s = pendulum.SinglePendulum()
        s.x_ub = np.array([+6.0, +6.0]); s.x_lb = np.array([-9.0, -6.0])
        ctl = nonlinear.ComputedTorqueController(s)
        ctl.rbar = np.array([-3.14])
        q = costfunction.QuadraticCostFunction.from_sys(s)
        q.xbar = np.array([ctl.rbar[0], 0]); q.INF = 300
        q.S[0, 0] = 10.0; q.S[1, 1] = 10.0
        s.u_ub[0] = +200; s.u_lb[0] = -200
        g = discretizer.GridDynamicSystem(s, [41, 41], [11], 0.05, False)
        ev = dynamicprogramming.PolicyEvaluatorWithLookUpTable(ctl, g, q)
        ev.save_time_history = False
        out = _meta(s, g, q)
        out.update(x_next_table=ev.x_next_table, G=ev.G, J0=ev.J.copy())
        U = np.array([ctl.c(g.state_from_node_id[i], ctl.rbar, 0) for i in range(g.nodes_n)])
        out["U"] = U
        for k in range(1, 11):
            ev.initialize_backward_step(); ev.compute_backward_step(); ev.finalize_backward_step()
            if k in (1, 10):
                out["J_%d" % k] = ev.J.copy()
    save("policy_eval_41x41", **out)


def case_rollout():
    """Closed-loop Euler rollouts of a LookUpTableController (pendulum 21x21x5, 40 sweeps, cleaned policy):
    x_{i+1} = cl_sys.f(x_i, r, t_i)*dt + x_i exactly as Simulator('euler') (simulation.py:298-324) with
    cl_sys = ctl + sys (controller.py:328-355)."""
    s, g, q = _pendulum_problem((21, 21), (5,))
    with quiet():
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, q)
        dp.save_time_history = False
        dp.compute_steps(40)
        dp.clean_infeasible_set()
        ctl = dp.get_lookup_table_controller()
        cl = ctl + s
    rng = np.random.default_rng(5)
    X0 = rng.uniform(s.x_lb * 0.9, s.x_ub * 0.9, size=(12, 2))
    X0[0] = [0.0, 0.0]; X0[1] = [-3.0, 0.5]; X0[2] = [6.2, 6.2]        # the last one leaves the grid quickly
    tf, npts = 3.0, 121
    dt = (tf + 0.0) / (npts - 1)
    t = np.linspace(0, tf, npts)
    X = np.zeros((12, npts, 2)); U = np.zeros((12, npts, 1))
    for b in range(12):
        x = X0[b].copy()
        for i in range(npts):
            X[b, i] = x
            U[b, i] = ctl.c(x, ctl.rbar, t[i])
            if i + 1 < npts:
                x = cl.f(x, ctl.rbar, t[i]) * dt + x
    out = _meta(s, g, q)
    out.update(J=dp.J, pi=dp.pi.astype(np.int16), X0=X0, tf=tf, npts=npts, X=X, U=U)
    save("rollout_pendulum_21x21x5", **out)


def case_spline():
    """DynamicProgramming2DRectBivariateSpline (dynamicprogramming.py:578-614): bicubic FITPACK spline refit of J
    every sweep, evaluated at x_next CLAMPED to the grid box (no zero fill).  Pendulum 21x21x5 and 41x31x7 (dt 0.1)."""
    out = {}
    for tag, (xd, ud, dt) in dict(a=((21, 21), (5,), 0.05), b=((41, 31), (7,), 0.1)).items():
        s, g, q = _pendulum_problem(xd, ud, dt=dt)
        if tag == "a":
            out.update(_meta(s, g, q))
        with quiet():
            dp = dynamicprogramming.DynamicProgramming2DRectBivariateSpline(g, q)
            dp.save_time_history = False
            stats = []
            for k in range(1, 9):
                dp.initialize_backward_step()
                if k in (1, 2, 8):
                    out["%s_coef_%d" % (tag, k)] = np.asarray(dp.J_interpol.get_coeffs()).copy()
                dp.compute_backward_step()
                delta = dp.finalize_backward_step()
                d = dp.J - dp.J_next
                stats.append([dp.J.max(), d.max(), d.min(), delta])
                if k in (1, 2, 8):
                    out["%s_J_%d" % (tag, k)] = dp.J.copy(); out["%s_pi_%d" % (tag, k)] = dp.pi.copy()
                    Q = np.sort(dp.Q, axis=1)
                    out["%s_gap_%d" % (tag, k)] = Q[:, 1] - Q[:, 0]      # margin of the argmin (tie detector)
            out["%s_stats" % tag] = np.array(stats)
            out["%s_knots_x" % tag], out["%s_knots_y" % tag] = dp.J_interpol.get_knots()
    save("spline_pendulum", **out)


def case_mountaincar():
    """MountainCar (mountaincar.py:22) as in mountain_car_with_valueiteration_quadratic.py, reduced to 41x41x5:
    a Manipulator with position-dependent H, C, B, g -- the generic mechanical tier (per-node tables)."""
    from pyro.dynamic import mountaincar
    with quiet():
        s = mountaincar.MountainCar()
        s.x_ub = np.array([+0.2, +2.0]); s.x_lb = np.array([-1.7, -2.0])
        s.u_ub[0] = +0.2; s.u_lb[0] = -0.2
        g = discretizer.GridDynamicSystem(s, [41, 41], [5])
        q = costfunction.QuadraticCostFunction.from_sys(s)
        q.xbar = np.array([0.0, 0.0]); q.INF = 30
        q.R[0, 0] = 10.0; q.S[0, 0] = 10.0; q.S[1, 1] = 10.0
        out = _meta(s, g, q)
        rng = np.random.default_rng(4)
        X = rng.uniform(s.x_lb, s.x_ub, size=(64, 2)); U = rng.uniform(s.u_lb, s.u_ub, size=(64, 1))
        out.update(f_X=X, f_U=U, f_dX=np.array([s.f(X[i], U[i]) for i in range(64)]))
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, q)
        dp.save_time_history = False
        out.update(x_next_table=g.x_next_table, x_next_isok=g.x_next_isok, G=dp.G, J0=dp.J.copy())
        stats = []
        for k in range(1, 21):
            dp.initialize_backward_step(); dp.compute_backward_step()
            delta = dp.finalize_backward_step()
            d = dp.J - dp.J_next
            stats.append([dp.J.max(), d.max(), d.min(), delta])
            if k in (1, 5, 20):
                out["J_%d" % k] = dp.J.copy(); out["pi_%d" % k] = dp.pi.copy()
                Q = np.sort(dp.Q, axis=1)
                out["gap_%d" % k] = Q[:, 1] - Q[:, 0]
        out["stats"] = np.array(stats)
    save("mountaincar_41x41x5", **out)


def case_mintime():
    """simple_pendulum_with_valueiteration_minimum_time.py reduced to 61x61x3: InvertedPendulum, bounds +-6,
    TimeCostFunction (g = 1 outside the EPS ball around xbar, h = 0), INF 10, EPS 0.1."""
    with quiet():
        s = pendulum.InvertedPendulum()
        s.x_ub = np.array([+6.0, +6.0]); s.x_lb = np.array([-6.0, -6.0])
        g = discretizer.GridDynamicSystem(s, [61, 61], [3])
        tcf = costfunction.TimeCostFunction(np.array([0.0, 0.0]))
        tcf.INF = 10.0; tcf.EPS = 0.1
        out = dict(x_lb=s.x_lb, x_ub=s.x_ub, u_lb=s.u_lb, u_ub=s.u_ub, dims=g.x_grid_dim, udims=g.u_grid_dim, dt=g.dt,
                   xbar=tcf.xbar, INF=tcf.INF, EPS=tcf.EPS)
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, tcf)
        dp.save_time_history = False
        out.update(G=dp.G, J0=dp.J.copy())
        stats = []
        for k in range(1, 41):
            dp.initialize_backward_step(); dp.compute_backward_step()
            delta = dp.finalize_backward_step()
            d = dp.J - dp.J_next
            stats.append([dp.J.max(), d.max(), d.min(), delta])
            if k in (1, 10, 40):
                out["J_%d" % k] = dp.J.copy(); out["pi_%d" % k] = dp.pi.copy()
                Q = np.sort(dp.Q, axis=1)
                out["gap_%d" % k] = Q[:, 1] - Q[:, 0]
        out["stats"] = np.array(stats)
    save("mintime_invpendulum_61x61x3", **out)


def case_acrobot():
    """Acrobot (pendulum.py:699): under-actuated double pendulum, 9x9x9x9 grid, 5 torques, a few sweeps."""
    with quiet():
        s = pendulum.Acrobot()
        s.x_ub = np.array([+2.0, +2.0, +4.0, +4.0]); s.x_lb = -s.x_ub
        g = discretizer.GridDynamicSystem(s, [9, 9, 9, 9], [5], dt=0.05)
        q = costfunction.QuadraticCostFunction.from_sys(s)
        q.INF = 200.0
        out = _meta(s, g, q)
        rng = np.random.default_rng(6)
        X = rng.uniform(s.x_lb, s.x_ub, size=(64, 4)); U = rng.uniform(s.u_lb, s.u_ub, size=(64, 1))
        out.update(f_X=X, f_U=U, f_dX=np.array([s.f(X[i], U[i]) for i in range(64)]))
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, q)
        dp.save_time_history = False
        for k in range(1, 7):
            dp.initialize_backward_step(); dp.compute_backward_step(); dp.finalize_backward_step()
            if k in (1, 6):
                out["J_%d" % k] = dp.J.copy(); out["pi_%d" % k] = dp.pi.copy()
                Q = np.sort(dp.Q, axis=1)
                out["gap_%d" % k] = Q[:, 1] - Q[:, 0]
    save("acrobot_9p4x5", **out)


def case_floatmass():
    """float_mass_dp_optimal_controller.py reduced to 51x51x21: FloatingSingleMass (linear state space in mechanical
    form), quadratic cost with R = 10, S = 10 I, INF 300."""
    from pyro.dynamic import massspringdamper
    with quiet():
        s = massspringdamper.FloatingSingleMass()
        s.x_ub[0] = 10.0; s.x_lb[0] = -10.0; s.x_lb[1] = -5.0; s.x_ub[1] = 5.0
        s.u_ub[0] = 5.0; s.u_lb[0] = -5.0
        g = discretizer.GridDynamicSystem(s, [51, 51], [21], 0.05)
        q = costfunction.QuadraticCostFunction.from_sys(s)
        q.xbar = np.array([-0, 0]); q.INF = 300
        q.R[0, 0] = 10.0; q.S[0, 0] = 10.0; q.S[1, 1] = 10.0
        out = _meta(s, g, q)
        rng = np.random.default_rng(8)
        X = rng.uniform(s.x_lb, s.x_ub, size=(64, 2)); U = rng.uniform(s.u_lb, s.u_ub, size=(64, 1))
        out.update(f_X=X, f_U=U, f_dX=np.array([s.f(X[i], U[i]) for i in range(64)]))
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, q)
        dp.save_time_history = False
        out.update(x_next_table=g.x_next_table, G=dp.G)
        for k in range(1, 31):
            dp.initialize_backward_step(); dp.compute_backward_step(); dp.finalize_backward_step()
            if k in (1, 30):
                out["J_%d" % k] = dp.J.copy(); out["pi_%d" % k] = dp.pi.copy()
    save("floatmass_51x51x21", **out)


def case_nearest():
    """interpol_method = 'nearest' (discretizer.py:570-587 hands it to RegularGridInterpolator every sweep): the LUT class on
    a 2-D pendulum, and on the 4-D cart-pole for a few sweeps; plus a switch linear -> nearest between sweeps."""
    s, g, q = _pendulum_problem((31, 21), (5,))
    out = _meta(s, g, q)
    with quiet():
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, q)
        dp.save_time_history = False
        dp.interpol_method = "nearest"
        out.update(x_next_table=g.x_next_table, G=dp.G, J0=dp.J.copy())
        for k in range(1, 9):
            dp.initialize_backward_step(); dp.compute_backward_step(); dp.finalize_backward_step()
            if k in (1, 3, 8):
                out["J_%d" % k] = dp.J.copy(); out["pi_%d" % k] = dp.pi.copy()
        # three linear sweeps, then nearest from the fourth on
        dm = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, q)
        dm.save_time_history = False
        dm.compute_steps(3)
        dm.interpol_method = "nearest"
        dm.compute_steps(2)
        out["Jmix_5"] = dm.J.copy(); out["pimix_5"] = dm.pi.copy()
    save("nearest_pendulum_31x21x5", **out)
    with quiet():
        c = cartpole.CartPole()
        g4 = discretizer.GridDynamicSystem(c, [7, 9, 7, 9], [3])
        q4 = costfunction.QuadraticCostFunction.from_sys(c)
        q4.INF = 1000.0
        d4 = dynamicprogramming.DynamicProgrammingWithLookUpTable(g4, q4)
        d4.save_time_history = False
        d4.interpol_method = "nearest"
        d4.compute_steps(4)
    out4 = _meta(c, g4, q4)
    out4.update(J_4=d4.J.copy(), pi_4=d4.pi.copy())
    save("nearest_cartpole_7x9x7x9x3", **out4)


def case_slinear():
    """interpol_method = 'slinear' (the order-1 spline of RegularGridInterpolator: the same interpolant as 'linear', evaluated by
    another code path of scipy): the LUT class on a 2-D pendulum and a 4-D cart-pole, and the same solves with 'linear' beside
    them -- the two differ by rounding only, which is what lets the GPU build serve 'slinear' with its linear sweeps."""
    s, g, q = _pendulum_problem((31, 21), (5,))
    out = _meta(s, g, q)
    with quiet():
        for tag in ("slinear", "linear"):
            dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, q)
            dp.save_time_history = False
            dp.interpol_method = tag
            dp.compute_steps(12)
            out["J12_" + tag] = dp.J.copy(); out["pi12_" + tag] = dp.pi.copy()
    save("slinear_pendulum_31x21x5", **out)
    with quiet():
        c = cartpole.CartPole()
        g4 = discretizer.GridDynamicSystem(c, [7, 9, 7, 9], [3])
        q4 = costfunction.QuadraticCostFunction.from_sys(c)
        q4.INF = 1000.0
        out4 = _meta(c, g4, q4)
        for tag in ("slinear", "linear"):
            d4 = dynamicprogramming.DynamicProgrammingWithLookUpTable(g4, q4)
            d4.save_time_history = False
            d4.interpol_method = tag
            d4.compute_steps(4)
            out4["J4_" + tag] = d4.J.copy(); out4["pi4_" + tag] = d4.pi.copy()
    save("slinear_cartpole_7x9x7x9x3", **out4)


def case_cubic():
    """interpol_method = 'cubic' / 'cubic_legacy' on 2-D grids (RegularGridInterpolator's order-3 spline: the not-a-knot tensor
    spline through J_k, ZERO outside the grid box -- bounds_error=False, fill_value=0 -- where
    DynamicProgramming2DRectBivariateSpline clamps).  SciPy >= 1.13 fits 'cubic' with an ITERATIVE sparse solver (gcrotmk at its
    default tolerance: the interpolant it returns is some 1e-5 of max|J| away from the interpolating spline) and keeps the exact
    fit (successive 1-D make_interp_spline) as 'cubic_legacy'; the reference forwards either string.  Recorded: the LUT class with
    'cubic_legacy' on two pendulum grids (J / pi / margin of the argmin after sweeps 1, 2, 8), the same solves with 'cubic' (J after
    2 and 8: the solver's tolerance, measured), and the cell-by-cell base class with 'cubic_legacy' for 2 sweeps on the small grid."""
    out = {}
    for tag, (xd, ud, dt) in dict(a=((31, 21), (5,), 0.05), b=((41, 31), (7,), 0.1)).items():
        s, g, q = _pendulum_problem(xd, ud, dt=dt)
        for k_, v_ in _meta(s, g, q).items():
            out[tag + "_" + k_] = v_
        with quiet():
            for method, sfx in (("cubic_legacy", ""), ("cubic", "_iter")):
                dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, q)
                dp.save_time_history = False
                dp.interpol_method = method
                for k in range(1, 9):
                    dp.initialize_backward_step()
                    dp.compute_backward_step()
                    dp.finalize_backward_step()
                    if k in (1, 2, 8) and not sfx:
                        out["%s_J_%d" % (tag, k)] = dp.J.copy(); out["%s_pi_%d" % (tag, k)] = dp.pi.copy()
                        Q = np.sort(dp.Q, axis=1)
                        out["%s_gap_%d" % (tag, k)] = Q[:, 1] - Q[:, 0]
                    elif k in (2, 8):
                        out["%s_J_%d%s" % (tag, k, sfx)] = dp.J.copy()
            if tag == "a":
                db = dynamicprogramming.DynamicProgramming(g, q)
                db.save_time_history = False
                db.interpol_method = "cubic_legacy"
                db.compute_steps(2)
                out["a_base_J_2"] = db.J.copy(); out["a_base_pi_2"] = db.pi.copy()
    save("cubic_pendulum", **out)


def case_trajectory():
    """What the reference's scripts do with a solve's policy: `cl = ctl + sys; cl.x0 = ...; cl.compute_trajectory(tf, n, 'euler')`
    (controller.py:517-530 -> simulation.CLosedLoopSimulator, :391-447) -- every field of the Trajectory (x, u, t, dx, y, r, J, dJ)
    for two initial states on the pendulum 21x21x5 policy of case_rollout; and an OPEN-loop pendulum under a constant torque
    with the default solver (`sys.compute_trajectory(2.0, 41)`: solve_ivp, simulation.py:259-292)."""
    s, g, q = _pendulum_problem((21, 21), (5,))
    out = _meta(s, g, q)
    with quiet():
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, q)
        dp.save_time_history = False
        dp.compute_steps(40)
        dp.clean_infeasible_set()
        ctl = dp.get_lookup_table_controller()
        cl = ctl + s
        out["pi"] = dp.pi.astype(np.int16)
        for k, x0 in enumerate(([-3.0, 0.5], [0.0, 0.0])):
            cl.x0 = np.array(x0)
            tr = cl.compute_trajectory(3.0, 121, "euler")
            for name in ("x", "u", "t", "dx", "y", "r", "J", "dJ"):
                out["cl%d_%s" % (k, name)] = np.asarray(getattr(tr, name))
            out["cl%d_x0" % k] = np.array(x0)
        s2 = pendulum.SinglePendulum()
        s2.ubar = np.array([1.5])
        s2.x0 = np.array([0.4, -0.2])
        tr = s2.compute_trajectory(2.0, 41)
        for name in ("x", "u", "t", "dx", "y", "J", "dJ"):
            out["ol_" + name] = np.asarray(getattr(tr, name))
    save("trajectory_pendulum_21x21x5", **out)


CASES = dict(trajectory=case_trajectory, cubic=case_cubic, slinear=case_slinear, nearest=case_nearest, longcar=case_longcar, car=case_car, suspension=case_suspension, floatmass=case_floatmass, acrobot=case_acrobot, mintime=case_mintime, mountaincar=case_mountaincar, spline=case_spline, f_kat=case_f_kat, rollout=case_rollout, obstacles=case_obstacles, helicopter=case_helicopter, reachability=case_reachability,
             policy_eval=case_policy_eval, cost_kat=case_cost_kat, grid_kat=case_grid_kat,
             pendulum_small=case_pendulum_small, config1=case_config1, lowdef=case_lowdef,
             pendulum_demo=case_pendulum_demo, cartpole_small=case_cartpole_small,
             cartpole_mid=case_cartpole_mid, twolink_small=case_twolink_small,
             twolink_dt=case_twolink_dt, doublependulum=case_doublependulum)

if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    for nm in names:
        t0 = time.time()
        CASES[nm]()
        print("  %s done in %.1f s" % (nm, time.time() - t0))
