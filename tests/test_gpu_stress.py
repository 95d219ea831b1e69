"""
GPU stress tests (run with -m gpu): properties a build of the library must have whatever its parity figures say.

Round 4 ended with a GPU memory access fault that one build of the library (commit 4e5b14a) produced in `bench.py --workload c3
--converged` and the GPU suite did not see.  Round 5 found what that build did (profiles/r05_hunt_*.log): its
error-feedback sweep k_sweep_lean4fb -- the same source with the exact-pass lanes kept as a scalar wave mask, a different register
assignment of the same instructions -- computed RANDOMLY WRONG results whenever three workgroups shared a CU: a few dozen waves per
sweep picked up garbage, different waves in every run, J reached 1e37 and NaN within a few sweeps, and the fault was the downstream
consequence.  The suite had passed because its feedback grids were too small to fill the CUs (25^4) or compared with a tolerance
a few wrong waves in 10^5 slip under.  What such a build cannot do is compute the same bits twice -- so that is what is asserted
here, on grids that reach full occupancy, for every sweep family:

  * one backup repeated from the same cost-to-go is bit-identical (J, pi, the statistics);
  * the stop-test solve bench.py runs -- a tuned create, timed sweeps, close, then three live handles (float32, float64, float32
    with error-feedback storage) advanced in lockstep batches with a device-side stop, five create / close cycles -- gives the
    same sweep counts and the same bits in every cycle, stays finite, and stays within the accuracy contract.

`python tools/r05_hunt/build_hunt.py e1a; PYROVI_LIB=$PWD/pyro_amd/libpyrovi_e1a.so python -m pytest tests/test_gpu_stress.py -m gpu`
is the build these tests were written against: test_one_backup_is_deterministic[fb-*] and the lockstep test fail on it
(profiles/r05_hunt_tests_on_e1a.log), every test passes on the product library.
"""
import contextlib
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dp(name, dtype=None, fb=False):
    from pyro_amd import configs
    from pyro_amd.planning import dynamicprogramming as DP
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build(name)
        dp = DP.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"], dtype=dtype or cfg["dtype"], f32_feedback=fb)
    dp.save_time_history = False
    dp.verbose = False
    return dp


# (kind, workload): 4-D grids large enough for three 512-thread workgroups on every CU (8 528 tiles on 41^4), a remainder of one
# and of three actions behind the groups of four, ragged axes; the 2-D float32 kernel, the float64 kernels, the n = 3 kernel
_DET = [("fb", "cartpole:41,41,41,41:21:float32"), ("fb", "cartpole:41,41,41,41:31:float32"), ("fb", "cartpole:31,33,35,37:21:float32"),
        ("f32", "cartpole:41,41,41,41:21:float32"), ("f32", "cartpole:41,41,41,41:31:float32"), ("f32", "cartpole:31,33,35,37:21:float32"),
        ("f64", "cartpole:31,31,31,31:21:float64"), ("f32", "pendulum:1001,1001:51:float32"), ("f64", "pendulum:401,401:21:float64"),
        ("f32", "twolink:21,21,21,21:5,5:float32"), ("f64", "twolink:21,21,21,21:5,5:float64"), ("f32", "h3s")]


@pytest.mark.parametrize("kind,name", _DET, ids=["%s-%s" % (k, n.split(":float")[0]) for k, n in _DET])
def test_one_backup_is_deterministic(kind, name):
    """The third backup from J0, repeated from the same J_2 (pvi_set_J clears the residuals of an error-feedback handle): J, pi
    and the three statistics are the same bits every time."""
    reps = 12
    dp = _dp(name, dtype="float64" if kind == "f64" else "float32", fb=kind == "fb")
    p = dp._p
    p.sweep(2, 1.0, -1.0)
    J2 = p.get_J()
    first = None
    for r in range(reps):
        p.set_J(J2)
        st, n = p.sweep(1, 1.0, -1.0)
        assert n == 1
        J, pi = p.get_J(), p.get_pi()
        assert np.isfinite(J).all(), (name, r, p.describe())
        if first is None:
            first = (J, pi, np.array(st[-1]))
        else:
            nbad = int(((J != first[0]) | (pi != first[1])).sum())
            assert nbad == 0, "%s %s: repeat %d differs from the first in %d nodes (%s)" % (kind, name, r, nbad, p.describe()[:160])
            assert np.array_equal(np.array(st[-1]), first[2]), (name, r, st[-1], first[2])
    p.close()


def test_lockstep_solves_of_three_live_handles_repeat_bit_for_bit():
    """What `bench.py --workload c3 --converged` does, on a 41^4 cart-pole: measure (a tuned create, 60 sweeps, close), then
    float32 / float64 / float32-with-feedback handles alive together, advanced in lockstep batches of 50 sweeps with the stop test
    on the device, closed; five cycles.  Every cycle: the same stop sweep in all three dtypes, finite J, float32 within 2.5e-5
    of float64 at every checkpoint and within 1e-5 at the end, feedback within 1e-6 at every checkpoint -- and the same bits as the
    first cycle."""
    name, tol, every = "cartpole:41,41,41,41:21:float32", 0.1, 50
    ref = None
    for cycle in range(5):
        pre = _dp(name)                                       # bench.py's `measure`: the create that times the tile candidates
        pre._p.sweep(60, 1.0, -1.0)
        pre._p.synchronize()
        pre._p.close()
        hs = {"f32": _dp(name)._p, "f64": _dp(name, dtype="float64")._p, "fb": _dp(name, fb=True)._p}
        assert "kernel=" in hs["fb"].describe()
        done = {k: 0 for k in hs}
        stop = {k: False for k in hs}
        worst = {"f32": 0.0, "fb": 0.0}
        while not all(stop.values()) and max(done.values()) < 4000:
            for k, h in hs.items():
                if not stop[k]:
                    st, n = h.sweep(every, 1.0, tol)
                    done[k] += n
                    stop[k] = n < every or (n and st[-1][3] <= tol)
            J64 = hs["f64"].get_J()
            m = np.abs(J64).max()
            for k in ("f32", "fb"):
                J = hs[k].get_J()
                assert np.isfinite(J).all(), (cycle, k, done)
                worst[k] = max(worst[k], float(np.abs(J - J64).max() / m))
        assert done["f32"] == done["f64"] == done["fb"] and all(stop.values()), (cycle, done, stop)
        assert worst["f32"] <= 2.5e-5 and worst["fb"] <= 1e-6, (cycle, worst)
        out = {k: (h.get_J(), h.get_pi()) for k, h in hs.items()}
        assert np.abs(out["f32"][0] - out["f64"][0]).max() / np.abs(out["f64"][0]).max() <= 1e-5
        assert "kernel=k_sweep_lean4fb<" in hs["fb"].describe() and "kernel=k_sweep_lean4<" in hs["f32"].describe()
        for h in hs.values():
            h.close()
        if ref is None:
            ref = (dict(done), out)
            print("stop sweep %d; worst |J32 - J64| / max %.2e, feedback %.2e" % (done["f32"], worst["f32"], worst["fb"]))
        else:
            assert done == ref[0], (cycle, done, ref[0])
            for k in out:
                assert np.array_equal(out[k][0], ref[1][k][0]) and np.array_equal(out[k][1], ref[1][k][1]), (cycle, k)


def test_multi_sweep_launch_under_stress():
    """ADVICE r4 (f64.hip grid_barrier_wt: write-through hand-off between XCDs without fences): many batches of the multi-sweep
    launch of small float64 grids, three handles of different shapes alive and interleaved, against one launch per sweep
    (pvi_override MULTI=0) -- the same bits after every batch, 30 batches of odd sizes."""
    from pyro_amd import _native
    names = ["pendulum:101,101:11:float64", "pendulum:201,201:21:float64", "pendulum:51,401:24:float64"]
    multi = [_dp(n)._p for n in names]
    for h in multi:
        h.sweep(1, 1.0, -1.0)
    with _native.overrides(MULTI="0"):                        # (the form of a handle's batches is decided at its first sweep)
        single = [_dp(n)._p for n in names]
        for h in single:
            h.sweep(1, 1.0, -1.0)
    assert all("multi=1" in h.describe() for h in multi) and all("multi=0" in h.describe() for h in single), [h.describe() for h in multi + single]
    sizes = [1, 2, 3, 7, 40, 5, 11, 64, 9, 13]
    for b in range(30):
        n = sizes[b % len(sizes)]
        for hm, hs_ in zip(multi, single):                    # interleaved: the three multi-sweep kernels alternate on the device
            sm, nm = hm.sweep(n, 1.0, -1.0)
            ss, ns = hs_.sweep(n, 1.0, -1.0)
            assert nm == ns == n
            assert np.array_equal(np.array(sm), np.array(ss)), (b, n)
        if b % 5 == 4:
            for hm, hs_ in zip(multi, single):
                assert np.array_equal(hm.get_J(), hs_.get_J()) and np.array_equal(hm.get_pi(), hs_.get_pi()), b
    for h in multi + single:
        h.close()
