"""Randomised run of the kernels that have not yet run on hardware (round 6; under emulation today -- PYROVI_LIB=tests/emu/_build/
libpyrovi_emu.so -- and on a GPU once one is reachable): random problems (systems, dims, action counts, bounds, dt, alpha, cost
weights, sweep counts) through

  fb2d     error-feedback storage of the 2-D window sweep (k_sweep_leanfb; pendulum family and MountainCar; UNPROVEN=1): within 1e-6
           of float64 and no further than plain storage + 2e-7;
  swapped  the cart-pole with internal_order="swapped" (PVI_DYN_CARTPOLE_SW, aligned window planes; with and without feedback, with
           and without the congruent pitch): within 1e-5 (1e-6 with feedback) of float64 in the reference's order;
  multi32  batches of the 2-D float32 sweep as one cooperative launch (k_sweep_leanm; MULTI32=1) against one launch per sweep:
           J, pi and every sweep's statistics bit for bit;
  fbcheck  the corruption detector (k_sweep_lean4fbc; FBCHECK=1) on random 4-D problems: silent, the bits of k_sweep_lean4fb.

usage: tools_fuzz_unproven.py [n_cases] [seed] [kind ...]"""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pyro_amd import _native
from pyro_amd.analysis import costfunction
from pyro_amd.dynamic import cartpole, manipulator, mountaincar, pendulum
from pyro_amd.planning import discretizer
from pyro_amd.planning import dynamicprogramming as DP

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
KINDS = sys.argv[3:] or ["fb2d", "swapped", "multi32", "fbcheck"]


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def make(grid, cf, dtype, alpha, fb=False, order="reference", **ov):
    with quiet(), _native.overrides(**ov):
        dp = DP.DynamicProgrammingWithLookUpTable(grid, cf, dtype=dtype, f32_feedback=fb, internal_order=order)
    dp.save_time_history = False
    dp.verbose = False
    dp.alpha = alpha
    return dp


def random_cost(s, cf):
    cf.xbar = rng.uniform(s.x_lb, s.x_ub) * 0.5
    cf.INF = float(rng.choice([300.0, 1000.0, 10000.0]))
    cf.EPS = float(rng.choice([1e-3, 0.3]))
    cf.R = cf.R * rng.uniform(0.1, 5.0)
    cf.S = cf.S + np.eye(s.n) * rng.uniform(0.0, 5.0)
    return float(rng.choice([1.0, 1.0, 0.99]))


fails, worst = 0, {}
for case in range(n_cases):
    kind = KINDS[case % len(KINDS)]
    with quiet():
        if kind in ("fb2d", "multi32"):
            name = str(rng.choice(["pendulum", "inverted", "mountaincar"]))
            s = {"pendulum": pendulum.SinglePendulum, "inverted": pendulum.InvertedPendulum, "mountaincar": mountaincar.MountainCar}[name]()
            dims = [int(rng.integers(8, 90)), int(rng.integers(8, 90))]
            udims = [int(rng.integers(2, 60)) if rng.random() < 0.4 else int(rng.integers(2, 13))]
        elif kind == "swapped":
            name, s = "cartpole", cartpole.CartPole()
            dims = [int(rng.integers(5, 15)) for _ in range(4)]
            udims = [int(rng.integers(1, 24))]
        else:
            name = str(rng.choice(["cartpole", "doublependulum", "twolink"]))
            s = {"cartpole": cartpole.CartPole, "doublependulum": pendulum.DoublePendulum, "twolink": manipulator.TwoLinkManipulator}[name]()
            dims = [int(rng.integers(4, 14)) for _ in range(4)]
            udims = [int(rng.integers(1, 24))] if s.m == 1 else [int(rng.integers(1, 6)), int(rng.integers(1, 6))]
        if name != "mountaincar":
            scale = rng.uniform(0.4, 1.4, size=s.n)
            s.x_ub, s.x_lb = s.x_ub * scale, s.x_lb * scale * rng.uniform(0.6, 1.0, size=s.n)
            s.u_ub, s.u_lb = s.u_ub * rng.uniform(0.2, 2.0), s.u_lb * rng.uniform(0.2, 2.0)
        dt = float(rng.choice([0.01, 0.05, 0.1]))
        grid = discretizer.GridDynamicSystem(s, dims, udims, dt=dt)
        cf = costfunction.QuadraticCostFunction.from_sys(s)
        alpha = random_cost(s, cf)
        if name == "mountaincar":
            cf.INF = 100.0
    nsw = int(rng.integers(8, 120))
    tag = "%3d %-8s %-14s dims %-16s A %-7s dt %.2f a %.2f sw %3d " % (case, kind, name, dims, udims, dt, alpha, nsw)
    try:
        if kind == "fb2d":
            d64, d32 = make(grid, cf, "float64", alpha), make(grid, cf, "float32", alpha)
            dfb = make(grid, cf, "float32", alpha, True, UNPROVEN="1")
            with quiet():
                for dp in (d64, d32, dfb):
                    dp.compute_steps(nsw)
            m = max(np.abs(d64.J).max(), 1e-300)
            e_fb, e_pl = np.abs(dfb.J - d64.J).max() / m, np.abs(d32.J - d64.J).max() / m
            desc = dfb._p.describe()
            bad = e_fb > 1e-6 or e_fb > e_pl + 2e-7 or "kernel=k_sweep_leanfb<" not in desc
            worst[kind] = max(worst.get(kind, 0.0), e_fb)
            print(tag + "feedback %.2e plain %.2e %s" % (e_fb, e_pl, "FAIL " + desc[:120] if bad else ""), flush=True)
        elif kind == "swapped":
            fb = bool(rng.random() < 0.5)
            ov = {"RS_CONG": "1"} if rng.random() < 0.5 else {}
            d64, dsw = make(grid, cf, "float64", alpha), make(grid, cf, "float32", alpha, fb, "swapped", **ov)
            with quiet():
                d64.compute_steps(nsw)
                dsw.compute_steps(nsw)
            m = max(np.abs(d64.J).max(), 1e-300)
            e = np.abs(dsw.J - d64.J).max() / m
            desc = dsw._p.describe()
            bad = e > (1e-6 if fb else 1e-5) or "order=swapped" not in desc
            worst[kind] = max(worst.get(kind, 0.0), e)
            print(tag + "fb=%d %s err %.2e %s" % (fb, ov, e, "FAIL " + desc[:160] if bad else ""), flush=True)
        elif kind == "multi32":
            with _native.overrides(MULTI32="1"):
                a = make(grid, cf, "float32", alpha)._p
                sa, _ = a.sweep(1, alpha, -1.0)
            with _native.overrides(MULTI="0"):
                b = make(grid, cf, "float32", alpha)._p
                sb, _ = b.sweep(1, alpha, -1.0)
            took = "kernel=k_sweep_leanm<" in a.describe()
            bad = not np.array_equal(np.array(sa), np.array(sb))
            for n in (nsw // 3 + 1, nsw // 2 + 1):
                sa, na = a.sweep(n, alpha, -1.0)
                sb, nb = b.sweep(n, alpha, -1.0)
                bad = bad or na != nb or not np.array_equal(np.array(sa), np.array(sb)) or not np.array_equal(a.get_J(), b.get_J()) \
                    or not np.array_equal(a.get_pi(), b.get_pi())
            print(tag + "multi launch taken: %s %s" % (took, "FAIL " + a.describe()[:160] if bad else ""), flush=True)
        else:
            try:
                a = make(grid, cf, "float32", alpha, True, FBCHECK="1")
                b = make(grid, cf, "float32", alpha, True)
            except _native.NativeError as e:
                if "PVI_FLAG_F32_FEEDBACK" not in str(e):
                    raise
                print(tag + "refused (no 4-D window sweep)", flush=True)
                continue
            with quiet():
                a.compute_steps(nsw)     # (PVI_ECORRUPT would raise here)
                b.compute_steps(nsw)
            bad = not np.array_equal(a.J, b.J) or not np.array_equal(a.pi, b.pi) or "kernel=k_sweep_lean4fbc<" not in a._p.describe()
            print(tag + "detector silent, bits equal: %s" % (not bad), flush=True)
    except Exception as e:                                   # noqa: BLE001 -- a campaign reports and goes on
        bad = True
        print(tag + "EXCEPTION %s: %s" % (type(e).__name__, str(e)[:200]), flush=True)
    fails += bool(bad)
print("unproven kernels: failures %d / %d; worst errors %s" % (fails, n_cases, {k: "%.2e" % v for k, v in worst.items()}))
sys.exit(1 if fails else 0)
