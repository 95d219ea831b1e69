"""
The BASELINE.json configurations, built through the class surface.
Each entry returns (sys, grid_sys, cost_function, dtype).  Used by bench.py and the full-size tests.
"""
import contextlib
import io

import numpy as np

from pyro_amd.analysis import costfunction
from pyro_amd.dynamic import cartpole, drone, manipulator, pendulum
from pyro_amd.planning import discretizer


def _quiet(fn, *a, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **kw)


def _pendulum(xdims, udims, dtype, demo=False):
    s = pendulum.SinglePendulum()
    if demo:      # examples/demos_by_tool/dynamicprogramming/pendulum_optimal_swingup_demo.py:18-33
        s.x_ub, s.x_lb = np.array([10.0, 10.0]), np.array([-10.0, -10.0])
    g = _quiet(discretizer.GridDynamicSystem, s, list(xdims), list(udims))
    cf = costfunction.QuadraticCostFunction.from_sys(s)
    cf.xbar = np.array([-3.14, 0.0])
    cf.INF = 500 if demo else 300
    if demo:
        cf.S = np.diag([10.0, 10.0])
    return s, g, cf, dtype


def _cartpole(xdims, udims, dtype, rail=1):
    s = cartpole.CartPole()
    s.xbar = np.array([0.0, np.pi, 0.0, 0.0])          # upright (examples/.../cartpole_with_lqr.py)
    if rail != 1:                                      # a rail `rail` times as long (the dynamics do not depend on x0)
        s.x_lb[0], s.x_ub[0] = s.x_lb[0] * rail, s.x_ub[0] * rail
    g = _quiet(discretizer.GridDynamicSystem, s, list(xdims), list(udims))
    cf = costfunction.QuadraticCostFunction.from_sys(s)
    cf.INF = 1000
    return s, g, cf, dtype


def _twolink(xdims, udims, dtype, dt=0.05):
    s = manipulator.TwoLinkManipulator()
    g = _quiet(discretizer.GridDynamicSystem, s, list(xdims), list(udims), dt)
    cf = costfunction.QuadraticCostFunction.from_sys(s)
    cf.INF = 1000
    return s, g, cf, dtype


def _helicopter(xdims, udims, dtype):
    """examples/demos_by_tool/dynamicprogramming/helicopter_tunnel.py:18-58 (the reference's 3-D demo: obstacle boxes in
    isavalidstate, QuadraticCostFunctionWithDomainCheck) on a finer grid."""
    s = drone.ConstantSpeedHelicopterTunnel()
    s.obstacles = [[(2, 2), (4, 4)], [(8, 5), (10, 10)], [(14, 0), (16, 4)]]
    s.mass, s.vx, s.width = 0.1, 5.0, 1.0
    s.x_ub, s.x_lb = np.array([+60.0, 10.0, +20.0]), np.array([-60.0, 0.0, 0.0])
    s.u_ub, s.u_lb = np.array([+20.0]), np.array([-20.0])
    g = _quiet(discretizer.GridDynamicSystem, s, list(xdims), list(udims), 0.05)
    cf = costfunction.QuadraticCostFunctionWithDomainCheck.from_sys(s)
    cf.xbar = np.array([0.0, 2.0, 20.0])
    cf.INF, cf.EPS = 100000, 0.2
    cf.Q[0, 0], cf.Q[1, 1], cf.Q[2, 2] = 2.0, 200.0, 0.0
    cf.R[0, 0] = 5.0
    cf.S[0, 0], cf.S[1, 1], cf.S[2, 2] = 20.0, 50.0, 0.0
    return s, g, cf, dtype


CONFIGS = {
    # name: (description, builder)
    "c1": ("pendulum 101x101 x 11 actions, f64 (BASELINE configs[0])", lambda: _pendulum((101, 101), (11,), "float64")),
    "c2": ("pendulum 1001x1001 x 51 actions, f32 (BASELINE configs[1])", lambda: _pendulum((1001, 1001), (51,), "float32")),
    "c2p": ("pendulum 201x201 x 201 actions, swing-up demo bounds, f32 (north-star grid)",
            lambda: _pendulum((201, 201), (201,), "float32", demo=True)),
    "c3": ("cart-pole 101^4 x 21 actions, f32 (BASELINE configs[2])", lambda: _cartpole((101,) * 4, (21,), "float32")),
    "c4": ("cart-pole 151^4 x 31 actions, f32 (BASELINE configs[3])", lambda: _cartpole((151,) * 4, (31,), "float32")),
    "c5": ("two-link 101^4 x 11x11 torques, f64 (BASELINE configs[4])", lambda: _twolink((101,) * 4, (11, 11), "float64")),
    # with the default dt = 0.05 only 12 % of C5's cells land inside the grid box, so most of the in-kernel
    # H(q)^-1 dynamics that configs[4] names is skipped; dt = 0.01 keeps most of them in (the dense variant)
    "c5d": ("two-link 101^4 x 11x11 torques, dt = 0.01, f64 (BASELINE configs[4], dense variant of SURVEY 8d)",
            lambda: _twolink((101,) * 4, (11, 11), "float64", dt=0.01)),
    # the explicit (non-mechanical) systems: the reference's 3-D demo on a grid 64 times the demo's 51^3
    "h3": ("helicopter tunnel 201x201x401 x 11 actions, obstacles + domain-check cost, f32 (helicopter_tunnel.py at 64x the nodes)",
           lambda: _helicopter((201, 201, 401), (11,), "float32")),
    "h3s": ("helicopter tunnel 51^3 x 11 actions, f32 (the demo's own grid)", lambda: _helicopter((51, 51, 51), (11,), "float32")),
    # reduced twins for quick checks
    "c3s": ("cart-pole 41^4 x 21 actions, f32", lambda: _cartpole((41,) * 4, (21,), "float32")),
    "c5s": ("two-link 41^4 x 11x11 torques, f64", lambda: _twolink((41,) * 4, (11, 11), "float64")),
    "c5ds": ("two-link 41^4 x 11x11 torques, dt = 0.01, f64", lambda: _twolink((41,) * 4, (11, 11), "float64", dt=0.01)),
}


def _custom(spec):
    """'pendulum:1001,1001:51:float32' | 'cartpole:41,41,41,41:21:float32' | 'twolink:21,21,21,21:5,5:float64'"""
    kind, xd, ud, dtype = spec.split(":")
    xd = tuple(int(v) for v in xd.split(","))
    ud = tuple(int(v) for v in ud.split(","))
    fn = {"pendulum": _pendulum, "cartpole": _cartpole, "twolink": _twolink}[kind]
    return "custom " + spec, (lambda: fn(xd, ud, dtype))


def weak_c3(world):
    """The weak-scaling family of bench.py --gpus N: C3's spacing on every axis, 100 N + 1 rows of axis 0 on a rail N
    times as long, so that every rank owns 100-101 rows of 101^3 nodes.  world = 1 is C3 itself."""
    world = int(world)
    return ("cart-pole (100*%d+1)x101^3 x 21 actions, f32: BASELINE configs[2] per GPU, rail x%d at the same spacing"
            % (world, world), lambda: _cartpole((100 * world + 1, 101, 101, 101), (21,), "float32", rail=world))


def build(name, world=1):
    if ":" in name:
        CONFIGS[name] = _custom(name)
    if name == "c3w":
        CONFIGS[name] = weak_c3(world)
    desc, fn = CONFIGS[name]
    s, g, cf, dtype = fn()
    return dict(name=name, description=desc, sys=s, grid_sys=g, cf=cf, dtype=dtype)
