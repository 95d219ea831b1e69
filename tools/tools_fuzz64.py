"""Randomised bit-identity run on the GPU, float64: the operation-for-operation kernel (k_sweep, override NO_SWEEP64=1) against
every form of k_sweep64 -- dense / sparse walk, line / patch mapping, whatever set-up timing picks -- on random problems
(system, dims, action counts, bounds, dt, alpha, cost weights, sweep counts).  Any difference in J, pi or the statistics
is a failure.   usage: tools_fuzz64.py [n_cases] [seed]"""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pyro_amd.analysis import costfunction
from pyro_amd.dynamic import cartpole, manipulator, pendulum
from pyro_amd import _native
from pyro_amd.planning import discretizer, dynamicprogramming

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
VARIANTS = [("ref", {"NO_SWEEP64": 1}), ("dense-line", {"SPARSE": 0, "PATCH": 0}),
            ("dense-patch", {"SPARSE": 0, "PATCH": 1}), ("sparse-line", {"SPARSE": 1, "PATCH": 0}),
            ("sparse-patch", {"SPARSE": 1, "PATCH": 1}), ("auto", {}),          # pvi_override pins
            # small grids: one launch per sweep / the fenced multi-sweep launch / the register table without the LDS window
            ("single", {"MULTI": 0}), ("multi-fenced", {"REGTAB": 0}), ("regtab-nowin", {"JWIN": 0})]
fails = 0
for case in range(n_cases):
    kind = rng.choice(["pendulum", "inverted", "cartpole", "doublependulum", "twolink", "twolink"])
    with contextlib.redirect_stdout(io.StringIO()):
        if kind in ("pendulum", "inverted"):
            s = pendulum.SinglePendulum() if kind == "pendulum" else pendulum.InvertedPendulum()
            dims = [int(rng.integers(4, 200)), int(rng.integers(4, 200))]
            udims = [int(rng.integers(1, 70)) if rng.random() < 0.5 else int(rng.integers(1, 13))]
        else:
            s = {"cartpole": cartpole.CartPole, "doublependulum": pendulum.DoublePendulum,
                 "twolink": manipulator.TwoLinkManipulator}[kind]()
            dims = [int(rng.integers(3, 20)) for _ in range(4)]
            udims = [int(rng.integers(1, 30))] if s.m == 1 else [int(rng.integers(1, 12)), int(rng.integers(1, 12))]
        scale = rng.uniform(0.3, 1.5, size=s.n)
        s.x_ub, s.x_lb = s.x_ub * scale, s.x_lb * scale * rng.uniform(0.5, 1.0, size=s.n)
        s.u_ub, s.u_lb = s.u_ub * rng.uniform(0.05, 2.0), s.u_lb * rng.uniform(0.05, 2.0)
        dt = float(rng.choice([0.01, 0.05, 0.1, 0.2]))
        grid = discretizer.GridDynamicSystem(s, dims, udims, dt=dt)
        cf = costfunction.QuadraticCostFunction.from_sys(s)
        cf.xbar = rng.uniform(s.x_lb, s.x_ub) * 0.5
        cf.INF = float(rng.choice([5.0, 50.0, 300.0, 1000.0]))      # small INF: in-box cells can cost MORE than INF
        cf.EPS = float(rng.choice([1e-3, 0.3]))
        cf.R = cf.R * rng.uniform(0.1, 5.0)
        cf.S = cf.S + np.eye(s.n) * rng.uniform(0.0, 5.0)
        alpha = float(rng.choice([1.0, 0.97]))
        nsw = int(rng.integers(1, 10))
        res = {}
        for tag, env in VARIANTS:
            with _native.overrides(**env):
                dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(grid, cf, dtype="float64")
            dp.save_time_history = False
            dp.verbose = False
            dp.alpha = alpha
            stats, n = dp._p.sweep(nsw, alpha, -1.0)
            res[tag] = (dp._p.get_J(), dp._p.get_pi(), np.array(stats), dp._p.describe())
            dp._p.close()
    ref = res["ref"]
    bad = [t for t, _ in VARIANTS[1:] if not (np.array_equal(res[t][0], ref[0]) and np.array_equal(res[t][1], ref[1]) and
                                              np.array_equal(res[t][2], ref[2]))]
    fails += bool(bad)
    frac = [w for w in res["sparse-patch"][3].split() if w.startswith("inbox")]
    print("%3d %-14s dims %-20s A %-8s dt %.2f a %.2f INF %6.0f sw %d  auto: %s %s %s" % (
        case, kind, dims, udims, dt, alpha, cf.INF, nsw, " ".join(w for w in res["auto"][3].split() if w.startswith(("mapping", "sparse", "multi", "regtab"))),
        frac[0] if frac else "", "FAIL " + ",".join(bad) if bad else ""), flush=True)
print("cases %d  failures %d" % (n_cases, fails))
sys.exit(1 if fails else 0)
