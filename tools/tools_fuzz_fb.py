"""Randomised run on the GPU of the float32 error-feedback storage (PVI_FLAG_F32_FEEDBACK, k_sweep_lean4fb): random 4-D problems
(three closed-form systems, dims, action counts, bounds, dt, alpha, cost weights, 10-300 sweeps), float32 with feedback and plain
float32 against float64.  Failure: the feedback run further than 1e-6 from float64 (relative to max |J|), or further than the
plain run by more than 2e-7.  Handles the library refuses (the 4-D window sweep does not apply) are counted, not failed.
usage: tools_fuzz_fb.py [n_cases] [seed]"""
import sys, contextlib, io
sys.path.insert(0, "/root/repo")
import numpy as np
from pyro_amd import _native
from pyro_amd.analysis import costfunction
from pyro_amd.dynamic import pendulum, cartpole, manipulator
from pyro_amd.planning import discretizer, dynamicprogramming

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst, fails, refused = 0.0, 0, 0
for case in range(n_cases):
    kind = rng.choice(["cartpole", "doublependulum", "twolink"])
    with contextlib.redirect_stdout(io.StringIO()):
        s = {"cartpole": cartpole.CartPole, "doublependulum": pendulum.DoublePendulum, "twolink": manipulator.TwoLinkManipulator}[kind]()
        dims = [int(rng.integers(4, 17)) for _ in range(4)]
        udims = [int(rng.integers(1, 30))] if s.m == 1 else [int(rng.integers(1, 7)), int(rng.integers(1, 7))]
        scale = rng.uniform(0.3, 1.5, size=s.n)
        s.x_ub, s.x_lb = s.x_ub * scale, s.x_lb * scale * rng.uniform(0.5, 1.0, size=s.n)
        s.u_ub, s.u_lb = s.u_ub * rng.uniform(0.2, 2.0), s.u_lb * rng.uniform(0.2, 2.0)
        dt = float(rng.choice([0.01, 0.05, 0.1]))
        grid = discretizer.GridDynamicSystem(s, dims, udims, dt=dt)
        cf = costfunction.QuadraticCostFunction.from_sys(s)
        cf.xbar = rng.uniform(s.x_lb, s.x_ub) * 0.5
        cf.INF = float(rng.choice([300.0, 1000.0, 10000.0]))
        cf.EPS = float(rng.choice([1e-3, 0.3]))
        cf.R = cf.R * rng.uniform(0.1, 5.0)
        cf.S = cf.S + np.eye(s.n) * rng.uniform(0.0, 5.0)
        alpha = float(rng.choice([1.0, 1.0, 0.99]))
        nsw = int(rng.integers(10, 300))
        res = {}
        try:
            for key, dtype, fb in (("f64", "float64", False), ("f32", "float32", False), ("fb", "float32", True)):
                dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(grid, cf, dtype=dtype, f32_feedback=fb)
                dp.save_time_history = False
                dp.alpha = alpha
                dp.compute_steps(nsw)
                res[key] = (dp.J.copy(), dp.pi.copy(), " ".join(w for w in dp._p.describe().split() if w.startswith(("kernel=", "tile="))))
                dp._p.close()
        except _native.NativeError as e:
            if "PVI_FLAG_F32_FEEDBACK" not in str(e):
                raise
            refused += 1
            print("%3d %-14s dims %-18s A %-8s refused: %s" % (case, kind, dims, udims, str(e)[-70:]), flush=True)
            continue
    m = max(np.abs(res["f64"][0]).max(), 1e-300)
    e_fb, e_pl = np.abs(res["fb"][0] - res["f64"][0]).max() / m, np.abs(res["f32"][0] - res["f64"][0]).max() / m
    worst = max(worst, e_fb)
    bad = e_fb > 1e-6 or e_fb > e_pl + 2e-7 or "k_sweep_lean4fb<" not in res["fb"][2]
    fails += bad
    print("%3d %-14s dims %-18s A %-8s dt %.2f a %.2f sw %3d  %s  feedback %.2e plain %.2e %s" %
          (case, kind, dims, udims, dt, alpha, nsw, res["fb"][2], e_fb, e_pl, "FAIL" if bad else ""), flush=True)
print("feedback: worst rel err %.3e, failures %d / %d (refused %d)" % (worst, fails, n_cases, refused))
sys.exit(1 if fails else 0)
