#!/usr/bin/env python3
"""Experiment (development tool, not a product path): the cart-pole with its generalised coordinates SWAPPED, q = (theta, x).

Same physics, same grid, same cost -- only the order of the state axes differs: (theta, x, dtheta, dx) instead of
(x, theta, dx, dtheta).  Why: the displacement of a node depends on (theta, dtheta) only (cartpole.py:369-437).  In the reference
order the LAST axis -- along which the lanes of k_sweep_lean4 run -- is dtheta, so every lane of a wave gathers with its own
shift and a 32-lane group spans 33 window slots as soon as one floor steps (42 % of the LDS cycles are bank conflicts,
DESIGN.md 4.2b).  With the coordinates swapped the last axis is dx: the lanes of a tile row all shift by the same amount, and the
row pitch of the window is already congruent to the tile width modulo 32.  The system goes through the generic tier (per-node
tables a0, Bn: PVI_DYN_NODE_2x1, filled here by array arithmetic), the sweep is the same k_sweep_lean4 loop.  No kernel changes:
if this sweeps faster than C3, an internal axis permutation of the cart-pole's closed form is worth building.

    python tools/tools_swapped_cartpole.py 101 21 200      # grid points per axis, actions, timed sweeps
    python tools/tools_swapped_cartpole.py 31 21 20 check  # + J after the sweeps against the reference order, transposed

CPU part (`python tools/tools_swapped_cartpole.py 9 5 0 tables`): the vectorised tables against the per-node loop of the model terms.
"""
import contextlib, io, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from pyro_amd.analysis import costfunction
from pyro_amd.dynamic import cartpole, mechanical
from pyro_amd.planning import discretizer


class CartPoleSwapped(mechanical.MechanicalSystem):
    """cartpole.py:322-437 with q = (theta, x)"""

    def __init__(self):
        super().__init__(dof=2, actuators=1)
        self.name = "Cart Pole, q = (theta, x)"
        self.u_lb[0], self.u_ub[0] = -10, +10
        self.l, self.lcg = 3, 0.5
        self.m1, self.m2, self.gravity = 1, 0.1, 9.81

    def H(self, q):
        off = self.m2 * self.lcg * np.cos(q[0])
        return np.array([[self.m2 * self.lcg ** 2, off], [off, self.m1 + self.m2]], dtype=float)

    def C(self, q, dq):
        C = np.zeros((2, 2))
        C[1, 0] = -self.m2 * self.lcg * np.sin(q[0]) * dq[0]
        return C

    def B(self, q):
        return np.array([[0.0], [1.0]])

    def g(self, q):
        return np.array([self.m2 * self.gravity * self.lcg * np.sin(q[0]), 0.0])

    def d(self, q, dq):
        return np.zeros(2)

    def _trig_vectorized(self, x_level):
        """(a0 [N, 2], Bn [Nq, 2, 1]) by array arithmetic: ddq = inv(H) (B u - C dq - g), the inverse of the 2 x 2 matrix written out"""
        th, xs, dth, dxs = [np.asarray(l, dtype=float) for l in x_level]
        s, c = np.sin(th), np.cos(th)
        h00, h11, off = self.m2 * self.lcg ** 2, self.m1 + self.m2, self.m2 * self.lcg * c
        det = h00 * h11 - off * off
        # rhs at u = 0: row 0: -g0 = -m2 g lcg sin(theta); row 1: -C[1,0] dtheta = m2 lcg sin(theta) dtheta^2
        r0 = (-self.m2 * self.gravity * self.lcg * s)[:, None]                   # [theta, dtheta]
        r1 = (self.m2 * self.lcg * s)[:, None] * (dth ** 2)[None, :]
        a_th = (h11 * r0 - off[:, None] * r1) / det[:, None]
        a_x = (-off[:, None] * r0 + h00 * r1) / det[:, None]
        n = [len(th), len(xs), len(dth), len(dxs)]
        a0 = np.empty(n + [2])
        a0[..., 0] = a_th[:, None, :, None]
        a0[..., 1] = a_x[:, None, :, None]
        Bn = np.empty((n[0], n[1], 2, 1))
        Bn[..., 0, 0] = (-off / det)[:, None]
        Bn[..., 1, 0] = (h00 / det)[:, None]
        return a0.reshape(-1, 2), Bn.reshape(-1, 2, 1)


def problem(npts, nact, swapped, dtype="float32"):
    s = CartPoleSwapped() if swapped else cartpole.CartPole()
    s.xbar = np.array([np.pi, 0.0, 0.0, 0.0]) if swapped else np.array([0.0, np.pi, 0.0, 0.0])
    with contextlib.redirect_stdout(io.StringIO()):
        g = discretizer.GridDynamicSystem(s, [npts] * 4, [nact])
    cf = costfunction.QuadraticCostFunction.from_sys(s)
    cf.INF = 1000
    return s, g, cf


def main():
    npts, nact, nsweeps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    mode = sys.argv[4] if len(sys.argv) > 4 else ""
    if mode == "tables":
        s, g, cf = problem(npts, nact, True)
        a0, Bn = s._trig_vectorized(g.x_level)
        s._trig_vectorized = None
        a1, B1 = mechanical.MechanicalSystem.device_trig(s, g.x_level)
        print("a0 max |vectorised - loop| %.3e (max |a0| %.3e), Bn %.3e" % (np.abs(a0 - a1).max(), np.abs(a1).max(), np.abs(Bn - B1).max()))
        # the same physics: f of the swapped system is the reference order's f with the axes permuted
        r = cartpole.CartPole()
        rng = np.random.default_rng(1)
        worst = 0.0
        for _ in range(200):
            x = rng.uniform(-5, 5, 4)
            u = rng.uniform(-10, 10, 1)
            fr = r.f(x, u)
            fs = CartPoleSwapped().f(x[[1, 0, 3, 2]], u)
            worst = max(worst, np.abs(fs - fr[[1, 0, 3, 2]]).max())
        print("f(swapped) against f(reference order) permuted: %.3e" % worst)
        return
    from pyro_amd import _native
    from pyro_amd.planning import dynamicprogramming as DP
    ov = dict(a.split("=", 1) for a in sys.argv[4:] if "=" in a)
    out = {}
    for swapped in (False, True):
        s, g, cf = problem(npts, nact, swapped)
        t0 = time.time()
        with _native.overrides(**ov), contextlib.redirect_stdout(io.StringIO()):
            dp = DP.DynamicProgrammingWithLookUpTable(g, cf, dtype="float32")
        p = dp._p
        p.synchronize()
        t1 = time.time()
        if nsweeps:
            p.sweep(3, 1.0, -1.0)
            p.sweep(nsweeps, 1.0, -1.0)
        print("nodes %d %s" % (g.nodes_n, p.describe()))
        print("TIME %s %d^4 x %d setup %.2f s  %.4f ms/sweep" % ("swapped" if swapped else "reference-order", npts, nact, t1 - t0,
                                                                  p.last_sweep_ms() / max(nsweeps, 1)), flush=True)
        if mode == "check":
            out[swapped] = p.get_J().reshape([npts] * 4)
        p.close()
    if mode == "check":
        Jr, Js = out[False], out[True].transpose(1, 0, 3, 2)
        print("J after %d sweeps: max |swapped^T - reference order| / max J = %.3e" % (nsweeps + 3, np.abs(Js - Jr).max() / np.abs(Jr).max()))


if __name__ == "__main__":
    main()
