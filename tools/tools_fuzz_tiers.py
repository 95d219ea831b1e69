"""Randomised cross-tier consistency on the GPU (float64): the table tier fed with the fused tier's own tables
(pvi_build_tables -> pvi_set_tables) must reproduce the fused sweeps bit for bit; the bicubic-spline mode must agree
between its fused and table variants.   usage: tools_fuzz_tiers.py [n_cases] [seed]"""
import sys, contextlib, io
sys.path.insert(0, "/root/repo")
import numpy as np
from pyro_amd import _native
from pyro_amd.analysis import costfunction
from pyro_amd.dynamic import pendulum, cartpole, manipulator
from pyro_amd.planning import discretizer

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fails = 0
for case in range(n_cases):
    kind = rng.choice(["pendulum", "inverted", "cartpole", "doublependulum", "twolink"])
    with contextlib.redirect_stdout(io.StringIO()):
        if kind in ("pendulum", "inverted"):
            s = pendulum.SinglePendulum() if kind == "pendulum" else pendulum.InvertedPendulum()
            dims = [int(rng.integers(4, 90)), int(rng.integers(4, 90))]
            udims = [int(rng.integers(1, 40))]
        else:
            s = {"cartpole": cartpole.CartPole, "doublependulum": pendulum.DoublePendulum,
                 "twolink": manipulator.TwoLinkManipulator}[kind]()
            dims = [int(rng.integers(3, 11)) for _ in range(4)]
            udims = [int(rng.integers(1, 12))] if s.m == 1 else [int(rng.integers(1, 5)), int(rng.integers(1, 5))]
        s.x_ub, s.x_lb = s.x_ub * rng.uniform(0.3, 1.5, size=s.n), s.x_lb * rng.uniform(0.3, 1.5, size=s.n)
        dt = float(rng.choice([0.01, 0.05, 0.1]))
        grid = discretizer.GridDynamicSystem(s, dims, udims, dt=dt, lookup=False)
        cf = costfunction.QuadraticCostFunction.from_sys(s)
        cf.xbar = rng.uniform(s.x_lb, s.x_ub) * 0.5
        cf.INF = float(rng.choice([50.0, 300.0]))
        alpha, nsw = float(rng.choice([1.0, 0.97])), int(rng.integers(1, 8))
        spline = s.n == 2 and min(dims) >= 4 and rng.random() < 0.5
        p = grid._device_problem(cost=cf.device_cost(), dtype="float64")
        xn, xo, ao, G = p.build_tables()
        if spline:
            p.set_interpolation("bicubic")
        p.terminal_cost(); J0 = p.get_J()
        p.sweep(nsw, alpha, -1.0)
        Jf, pif = p.get_J(), p.get_pi()
        p.close()
        h = _native.Problem(grid.x_level, grid.u_level, s.x_lb, s.x_ub, s.u_lb, s.u_ub, dt, dynamics_id=_native.DYN_TABLE,
                            table_inf=float(cf.INF))
        base_sem = rng.random() < 0.5 and not spline          # base-class semantics: exact INF on invalid cells
        okm = (xo & ao) if base_sem else None
        if spline:
            h.set_interpolation("bicubic")
        h.set_tables(xn, G, okm)
        h.set_J(J0)
        h.sweep(nsw, alpha, -1.0)
        Jt, pit = h.get_J(), h.get_pi()
        h.close()
    same = np.array_equal(Jf, Jt) and np.array_equal(pif, pit)
    fails += not same
    print("%3d %-14s dims %-16s A %-7s dt %.2f a %.2f sw %d %s  %s" %
          (case, kind, dims, udims, dt, alpha, nsw, "spline" if spline else ("linear/base" if base_sem else "linear"),
           "ok" if same else "MISMATCH max|dJ| %.3e" % np.abs(Jf - Jt).max()), flush=True)
print("mismatches %d / %d" % (fails, n_cases))
sys.exit(1 if fails else 0)
