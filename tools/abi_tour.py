"""A tour of the C ABI for the sanitizer host build (run with LD_PRELOAD=<ubsan runtime> PYROVI_LIB=.../libpyrovi_ubsan.so):
create / destroy of every handle kind, sweeps with a stop, uploads and downloads, tables, packed tables, spline mode,
rollouts, the explicit systems with obstacles, self check, a one-rank shard, error paths."""
import contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pyro_amd import _native, configs
from pyro_amd.analysis import costfunction
from pyro_amd.dynamic import drone, pendulum, vehicle_propulsion
from pyro_amd.planning import discretizer, dynamicprogramming

assert os.environ.get("PYROVI_LIB") is None or os.environ["PYROVI_LIB"] == _native.LIB_PATH
with contextlib.redirect_stdout(io.StringIO()):
    for name in ("pendulum:61,61:9:float32", "pendulum:33,47:300:float64", "cartpole:13,11,12,10:5:float32",
                 "twolink:9,9,9,9:3,3:float64"):
        cfg = configs.build(name)
        g = cfg["grid_sys"]
        p = g._device_problem(cost=cfg["cf"].device_cost(), dtype=cfg["dtype"])
        p.terminal_cost()
        p.sweep(7, 0.99, -1.0)
        p.sweep(50, 1.0, 5.0)
        J, pi = p.get_J(), p.get_pi()
        p.set_J(J); p.set_pi(pi)
        d, n = p.self_check(1.0)
        assert d < 1e-4
        xn, xo, ao, G = p.build_tables()
        if g.sys.n == 2:
            X, U = p.rollout(np.zeros((5, 2)), 20, 0.05)
        p.close()
        for dtype in ("float32", "float64"):
            h = _native.Problem(g.x_level, g.u_level, g.sys.x_lb, g.sys.x_ub, g.sys.u_lb, g.sys.u_ub, g.dt, dtype=dtype,
                                dynamics_id=_native.DYN_TABLE, table_inf=float(cfg["cf"].INF))
            if g.sys.n == 2 and dtype == "float64":
                h.set_interpolation("bicubic")
            h.set_tables(xn, G, (xo & ao) if dtype == "float32" else None)
            h.set_J(J)
            h.sweep(3, 1.0, -1.0)
            h.get_J(); h.get_pi()
            try:
                h.set_tables(xn[:5], G[:5])
            except ValueError:
                pass
            h.close()
    s = drone.ConstantSpeedHelicopterTunnel()
    gs = discretizer.GridDynamicSystem(s, (9, 9, 9), [5], 0.05)
    q = costfunction.QuadraticCostFunctionWithDomainCheck.from_sys(s)
    dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(gs, q)
    dp.compute_steps(3)
    _ = gs.x_next_table, dp.G
    s = vehicle_propulsion.LongitudinalFrontWheelDriveCarWithWheelSlipInput()
    gs = discretizer.GridDynamicSystem(s, [15, 15], [5], 0.05)
    dp = dynamicprogramming.DynamicProgramming(gs, costfunction.QuadraticCostFunction.from_sys(s), dtype="float32")
    dp.compute_steps(3)
    s = pendulum.SinglePendulum()
    gs = discretizer.GridDynamicSystem(s, [41, 41], [3])
    dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(gs, costfunction.Reachability(s.isavalidstate, np.array([-3.14, 0.0])))
    dp.solve_bellman_equation(tol=1.0)
    # one-rank shard without and with an RCCL communicator
    cfg = configs.build("cartpole:9,9,9,9:5:float32")
    sh = cfg["grid_sys"]._shard_problem(0, 1, 3, cost=cfg["cf"].device_cost(), dtype="float32")
    sh.terminal_cost(); sh.sweep(4, 1.0, -1.0); sh.get_J(); sh.get_pi(); sh.describe(); sh.close()
    sh = cfg["grid_sys"]._shard_problem(0, 1, 3, comm_id=_native.comm_unique_id(), cost=cfg["cf"].device_cost(), dtype="float32")
    sh.terminal_cost(); sh.sweep(4, 1.0, 0.5); sh.close()
    # error paths
    try:
        _native.Problem([np.linspace(0, 1, 1)] * 2, [np.linspace(0, 1, 2)], [0, 0], [1, 1], [0], [1], 0.1)
    except RuntimeError:
        pass
print("ABI-TOUR-OK")
