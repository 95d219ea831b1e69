"""
The multi-rank path (slab partition + halo exchange + statistics all-reduce) on CPU: two processes,
`gloo` backend, the oracle as the per-slab compute (the product's HipSlab needs a GPU; the exchange
logic in pyro_amd/parallel.py is the same).  Rows outside a rank's stored slab are NaN, so a halo
that is too small, or a wrong exchange, poisons the result.
"""
import contextlib
import io
import os
import socket

import numpy as np
import pytest

from oracle import vi_oracle as O


class OracleSlab:
    def __init__(self, grid_sys, cost, dtype, rows, halo, device):
        import torch
        s = grid_sys.sys
        dyn_id, params = s.device_dynamics()
        self.p = O.Problem(grid_sys.x_level, grid_sys.u_level, grid_sys.dt, dyn_id, np.array(params), cost["Q"], cost["R"],
                           cost["S"], cost["xbar"], cost["ubar"], cost["INF"], cost["EPS"])
        n0 = self.p.dims[0]
        self.rows = rows
        self.store_rows = (max(0, rows[0] - halo), min(n0, rows[1] + halo))
        self.plane = self.p.nodes_n // n0
        self.buf = [np.full(self.p.nodes_n, np.nan), np.full(self.p.nodes_n, np.nan)]
        self.t = [torch.from_numpy(b) for b in self.buf]
        self.cur = 0
        self.ids = np.arange(rows[0] * self.plane, rows[1] * self.plane)
        self.pi = np.zeros(len(self.ids), dtype=np.int64)
        self._st = None

    def terminal_cost(self):
        a, b = self.store_rows[0] * self.plane, self.store_rows[1] * self.plane
        self.buf[self.cur][a:b] = O.terminal_cost(self.p)[a:b]

    def sweep(self, alpha):
        Jc = self.buf[self.cur]
        Jn, self.pi = O.sweep(self.p, Jc, alpha, ids=self.ids)
        nxt = self.buf[self.cur ^ 1]
        nxt[:] = np.nan
        nxt[self.ids] = Jn
        d = Jn - Jc[self.ids]
        self._st = np.array([Jn.max(), d.max(), d.min()])
        self.cur ^= 1

    def stats(self):
        return self._st

    def rows_view(self, row0, nrows):
        return self.t[self.cur][row0 * self.plane:(row0 + nrows) * self.plane]

    def owned_J(self, prev=False):
        return self.buf[self.cur ^ (1 if prev else 0)][self.ids].copy()

    def set_owned_J(self, J):
        self.buf[self.cur][:] = np.nan
        self.buf[self.cur][self.ids] = J

    def owned_pi(self):
        return self.pi


class COracleSlab(OracleSlab):
    """The same slab on the oracle's C twin (bit-identical to the NumPy oracle, tests/test_c_oracle.py): for the
    618-sweep solve."""

    def __init__(self, *a):
        super().__init__(*a)
        from oracle import c_oracle as CO
        self.c = CO.CProblem(self.p)

    def sweep(self, alpha):
        Jc = self.buf[self.cur]
        Jn, self.pi = self.c.sweep(np.nan_to_num(Jc, nan=np.nan), alpha, int(self.ids[0]), int(self.ids[-1]) + 1, threads=1)
        nxt = self.buf[self.cur ^ 1]
        nxt[:] = np.nan
        nxt[self.ids] = Jn
        d = Jn - Jc[self.ids]
        self._st = np.array([Jn.max(), d.max(), d.min()])
        self.cur ^= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(case):
    from pyro_amd import configs
    with contextlib.redirect_stdout(io.StringIO()):
        return configs.build(case)


def _worker(rank, world, port, case, sweeps, halo, out):
    import torch.distributed as dist
    from pyro_amd import parallel
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        cfg = _build(case)
        vi = parallel.ShardedValueIteration(cfg["grid_sys"], cfg["cf"], dist, dtype="float64", slab_factory=OracleSlab,
                                            halo=halo)
        stats = [vi.sweep(1.0) for _ in range(sweeps)]
        J, pi = vi.gather()
        if rank == 0:
            np.savez(out, J=J, pi=pi, stats=np.array(stats), p2p=vi.p2p, halo=vi.halo)
    finally:
        dist.destroy_process_group()


def _reference(case, sweeps):
    cfg = _build(case)
    g, cf, s = cfg["grid_sys"], cfg["cf"], cfg["sys"]
    dyn_id, params = s.device_dynamics()
    p = O.Problem(g.x_level, g.u_level, g.dt, dyn_id, np.array(params), cf.Q, cf.R, cf.S, cf.xbar, cf.ubar, cf.INF, cf.EPS)
    J = O.terminal_cost(p)
    stats = []
    for _ in range(sweeps):
        Jn, pi = O.sweep(p, J)
        st, delta = O.sweep_stats(Jn, J)
        stats.append(list(st) + [delta])
        J = Jn
    return J, pi, np.array(stats)


@pytest.mark.parametrize("case,world,halo", [
    ("pendulum:41,21:5:float64", 2, None),        # neighbour exchange (halo 2 rows < slab)
    ("pendulum:12,31:5:float64", 3, 5),           # slabs thinner than the halo -> all-gather fall-back
    ("cartpole:9,7,9,7:3:float64", 2, None),      # 4-D
])
def test_sharded_sweeps_equal_single_process(tmp_path, case, world, halo):
    import torch.multiprocessing as mp
    out = str(tmp_path / "res.npz")
    sweeps = 5
    mp.spawn(_worker, args=(world, _free_port(), case, sweeps, halo, out), nprocs=world, join=True)
    r = np.load(out)
    J, pi, stats = _reference(case, sweeps)
    assert not np.isnan(r["J"]).any()
    assert np.array_equal(r["J"], J)                       # bit-identical: same arithmetic per node
    assert np.array_equal(r["pi"], pi)
    np.testing.assert_allclose(r["stats"], stats, rtol=0, atol=0)
    assert bool(r["p2p"]) == (halo is None)


def test_partition_and_halo():
    from pyro_amd import parallel
    assert parallel.partition_rows(151, 8) == [(0, 19), (19, 38), (38, 57), (57, 76), (76, 95), (95, 114), (114, 133), (133, 151)]
    assert parallel.partition_rows(10, 3) == [(0, 4), (4, 7), (7, 10)]
    cfg = _build("cartpole:151,5,5,5:3:float32")
    # C4: the largest displacement along axis 0 is 2*pi*0.05 / (4*pi/150) = 3.75 cells: lower corner 3 rows away, upper 4.
    # (SURVEY 8(e) counted ceil(3.75) + 1 = 5: one row more than the interpolation reads)
    assert parallel.halo_rows(cfg["grid_sys"]) == 4
    assert parallel._rows_for_reach(3.0) == 4 and parallel._rows_for_reach(2.9999999999999996) == 4 and parallel._rows_for_reach(0.2) == 1


def _harness_worker(rank, world, port, out):
    """bench.py --gpus N harness on CPU: the timed region is max-over-ranks, every rank runs the same number of batches,
    a failure on ONE rank is seen by all."""
    import time
    import torch
    import torch.distributed as dist
    from pyro_amd import parallel_bench as PB
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        class Drv:
            calls = 0

            def run(self, n):
                Drv.calls += 1
                time.sleep(0.004 * n * (1 + 2 * rank))          # rank 1 is three times slower
                return [1.0, 2.0, 3.0, 4.0]
        drv = Drv()
        elapsed, batches, st = PB._timed(drv, dist, torch, steps=5, warmup=1)
        ok_all = PB._agree(dist, torch, True)
        ok_one = PB._agree(dist, torch, rank != 1)               # rank 1 reports a failure
        res = [None] * world
        dist.all_gather_object(res, (elapsed, batches, Drv.calls, ok_all, ok_one))
        if rank == 0:
            np.save(out, np.array(res, dtype=float))
    finally:
        dist.destroy_process_group()


def test_bench_harness_times_the_slowest_rank(tmp_path):
    import torch.multiprocessing as mp
    from pyro_amd import parallel_bench as PB
    out = str(tmp_path / "h.npy")
    mp.spawn(_harness_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = np.load(out)
    assert r[0, 0] == r[1, 0] and r[0, 1] == r[1, 1] and r[0, 2] == r[1, 2]     # same region, batches and calls on both ranks
    assert r[0, 0] >= PB.MIN_REGION_S                                           # batches repeat until the region is long enough
    assert r[0, 0] >= r[0, 1] * 5 * 0.004 * 3 * 0.9                             # ... and it is the SLOW rank's time
    assert r[:, 3].all() and not r[:, 4].any()                                  # one failing rank flips the decision everywhere


def test_weak_scaling_family_is_c3_per_rank():
    """bench.py --gpus N: 100 N + 1 rows on a rail N times as long -- C3's spacing, N = 1 is C3, 100-101 rows per rank."""
    from pyro_amd import parallel
    c3 = _build("c3")["grid_sys"]
    from pyro_amd import configs
    for world in (1, 2, 8):
        with contextlib.redirect_stdout(io.StringIO()):
            g = configs.build("c3w", world=world)["grid_sys"]
        assert list(g.x_grid_dim) == [100 * world + 1, 101, 101, 101] and list(g.u_grid_dim) == [21]
        np.testing.assert_allclose(g.x_step_size, c3.x_step_size, rtol=1e-14)
        for ax in (1, 2, 3):
            assert np.array_equal(g.x_level[ax], c3.x_level[ax])
        rows = [b - a for a, b in parallel.partition_rows(g.x_grid_dim[0], world)]
        assert set(rows) <= {100, 101} and parallel.halo_rows(g) == parallel.halo_rows(c3)
    with contextlib.redirect_stdout(io.StringIO()):
        g1 = configs.build("c3w", world=1)["grid_sys"]
    assert all(np.array_equal(a, b) for a, b in zip(g1.x_level, c3.x_level))


# ---- the reference's class surface over a sharded grid (VERDICT r2 row b') ---------------------------------------------
def _surface_worker(rank, world, port, out):
    import torch.distributed as dist
    from pyro_amd import parallel
    from pyro_amd.planning import dynamicprogramming
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        cfg = _build("c1")
        with contextlib.redirect_stdout(io.StringIO()) as log:
            dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(
                cfg["grid_sys"], cfg["cf"], comm=parallel.TorchDistComm(dist, slab_factory=COracleSlab))
            dp.save_time_history = False
            assert dp.sharded and dp.tier == "fused"
            J0 = dp.J.copy()
            dp.compute_steps(3)
            k3, J3 = dp.k, dp.J.copy()
            dp.solve_bellman_equation(tol=0.1)
            J, pi, Jn, k = dp.J.copy(), dp.pi.copy(), dp.J_next.copy(), dp.k
            ctl = dp.get_lookup_table_controller()
            u = [float(ctl.c(np.array(x), None)[0]) for x in ([-1.0, 0.5], [2.0, -3.0], [7.0, 0.0])]
            dp.clean_infeasible_set()
            Jc, pic = dp.J.copy(), dp.pi.copy()
            dp.save_latest(out + ".r%d" % rank)
            # a host edit of J goes back to every slab: one more backup from the cleaned cost-to-go
            dp.compute_steps(1)
            J_after_clean = dp.J.copy()
        np.savez(out + ".rank%d.npz" % rank, J0=J0, J3=J3, k3=k3, J=J, pi=pi, Jn=Jn, k=k, u=np.array(u), Jc=Jc, pic=pic,
                 J_after_clean=J_after_clean, lines=np.array(log.getvalue().count(" max: ")))
    finally:
        dist.destroy_process_group()


def test_class_surface_on_two_ranks_solves_config1_like_the_reference(tmp_path):
    """DynamicProgrammingWithLookUpTable(grid_sys, cf, comm=...) on two gloo ranks: compute_steps,
    solve_bellman_equation, J / pi / J_next, get_lookup_table_controller, clean_infeasible_set and save_latest behave
    as on one device -- BASELINE config 1 stops after the reference's 618 sweeps with the reference's J* and pi*
    (dynamicprogramming.py:265-334, 472-485), on every rank."""
    import torch.multiprocessing as mp
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "config1_pendulum_101x101x11.npz"))
    out = str(tmp_path / "surf")
    mp.spawn(_surface_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    cfg = _build("c1")
    gs, cf, s = cfg["grid_sys"], cfg["cf"], cfg["sys"]
    dyn_id, params = s.device_dynamics()
    p = O.Problem(gs.x_level, gs.u_level, gs.dt, dyn_id, np.array(params), cf.Q, cf.R, cf.S, cf.xbar, cf.ubar, cf.INF, cf.EPS)
    for rank in range(2):
        r = np.load(out + ".rank%d.npz" % rank)
        assert int(r["k3"]) == 3 and int(r["k"]) == int(g["sweeps"]) == 618
        assert np.array_equal(r["J0"], O.terminal_cost(p))
        assert np.abs(r["J"] - g["J"]).max() <= 1e-12 * np.abs(g["J"]).max()
        assert np.array_equal(r["pi"], g["pi"].astype(np.int64))
        assert np.abs(r["Jn"] - g["J_prev"]).max() <= 1e-12 * np.abs(g["J"]).max()
        assert int(r["lines"]) == 618 + 1                         # finalize_backward_step printed every sweep
        Jc, pic = O.clean_infeasible_set(p, r["J"], r["pi"], s.ubar)
        assert np.array_equal(r["Jc"], Jc) and np.array_equal(r["pic"], pic)
        for x, u in zip(([-1.0, 0.5], [2.0, -3.0], [7.0, 0.0]), r["u"]):
            assert u == float(np.ravel(O.controller_c(p, r["pi"], np.array(x)))[0])
        Jn, _ = O.sweep(p, Jc)
        assert np.array_equal(r["J_after_clean"], Jn)
        assert np.array_equal(np.load(out + ".r%d_J_inf.npy" % rank), r["Jn"])
        assert np.array_equal(np.load(out + ".r%d_pi_inf.npy" % rank), r["pic"])


def test_interpol_method_is_not_silently_ignored():
    """The reference hands dp.interpol_method to the interpolant of every sweep (dynamicprogramming.py:186-189,
    discretizer.py:570-587); a kind the kernels do not implement must raise, not compute bilinear results."""
    from pyro_amd.planning import dynamicprogramming as DPm

    class Probe(DPm.DynamicProgramming):
        def _make_engine(self):                     # no device needed for the attribute's contract
            self._host, self._dirty, self.tier = {}, False, "fused"

            class P:
                def terminal_cost(self): pass
                def get_J(self, prev=False): return np.zeros(1)
                def get_pi(self): return np.zeros(1, dtype=int)
            self._p = P()
    cfg = _build("pendulum:5,5:3:float64")
    dp = Probe(cfg["grid_sys"], cfg["cf"])
    assert dp.interpol_method == "linear"
    dp.interpol_method = "linear"
    for kind in ("quintic", "pchip", "quintic_legacy"):
        with pytest.raises(NotImplementedError):
            dp.interpol_method = kind
    # 'cubic' / 'cubic_legacy' (round 6): the spline sweep of the table tier on 2-D grids -- sharded engines and grids of other
    # dimensions keep raising, and a refused assignment leaves the attribute as it was
    cfg4 = _build("cartpole:5,5,5,5:3:float64")
    dp4 = Probe(cfg4["grid_sys"], cfg4["cf"])
    for kind in ("cubic", "cubic_legacy"):
        with pytest.raises(NotImplementedError, match="2-D"):
            dp4.interpol_method = kind
        assert dp4.interpol_method == "linear"
    dp.comm = object()
    with pytest.raises(NotImplementedError):
        dp.interpol_method = "cubic"
    assert dp.interpol_method == "linear"
    dp.comm = None
    # 'slinear' (scipy's order-1 spline) IS the linear interpolant: served by the same engine, nothing is rebuilt (round 6;
    # tests/golden/slinear_*.npz: the reference's own solves with the two methods agree to 2e-15)
    engine = dp._p
    dp.interpol_method = "slinear"
    assert dp.interpol_method == "slinear" and dp._p is engine
    dp.interpol_method = "linear"
    assert dp._p is engine
    with pytest.raises(NotImplementedError):        # a linear engine cannot turn into the spline class by assignment
        dp.interpol_method = "bicubic"
    assert dp.interpol_method == "linear"
    # 'nearest' is implemented by the table tier (round 4; the GPU test runs it against the reference's own results): on one
    # device the engine is rebuilt there, a sharded engine keeps the interpolant it was built with and says so
    dp.comm = object()
    with pytest.raises(NotImplementedError):
        dp.interpol_method = "nearest"
    assert dp.interpol_method == "linear"


# ---- one halo width for the whole grid --------------------------------------------------------------
def _contracting_grid():
    from pyro_amd.dynamic import system
    from pyro_amd.planning import discretizer

    class Contracting(system.ContinuousDynamicSystem):
        def __init__(self):
            super().__init__(2, 1, 2)
            self.x_ub, self.x_lb = np.array([3.0, 1.0]), np.array([-3.0, -1.0])

        def f(self, x, u, t=0):
            return np.array([-2.0 * x[0] + x[1], u[0]])

    with contextlib.redirect_stdout(io.StringIO()):
        return discretizer.GridDynamicSystem(Contracting(), [61, 11], [3], dt=0.5)


def test_local_halo_bounds_differ_between_ranks_and_are_reduced():
    """A system whose axis-0 speed depends on x0 (f = [-2 x0 + x1, u]) gives every slab its own reach: the halo handed to
    the exchange must be the LARGEST, the same on every rank (mismatched send / recv counts otherwise)."""
    from pyro_amd import parallel
    gs = _contracting_grid()
    parts = parallel.partition_rows(61, 3)
    local = [parallel.halo_rows(gs, rows) for rows in parts]
    assert len(set(local)) > 1                                    # the per-rank bounds really differ
    assert max(local) == parallel.halo_rows(gs)                   # ... and their maximum is the whole-grid bound
    xn = gs._xnext_rows(parts[1][0] * 11, parts[1][1] * 11)[0]
    assert parallel.halo_rows(gs, parts[1], xn=xn) == local[1]    # a table the caller already has is not rebuilt


def _halo_worker(rank, world, port, out):
    import torch.distributed as dist
    from pyro_amd import parallel
    from pyro_amd.analysis import costfunction
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        gs = _contracting_grid()
        cf = costfunction.QuadraticCostFunction.from_sys(gs.sys)

        class Stub:                                               # no compute: only the constructor's halo agreement
            def __init__(self, grid_sys, cost, dtype, rows, halo, device):
                self.halo = halo

            def terminal_cost(self):
                pass
        vi = parallel.ShardedValueIteration(gs, cf, dist, slab_factory=Stub)
        res = [None] * world
        dist.all_gather_object(res, (vi.halo, vi.p2p, parallel.halo_rows(gs, vi.rows)))
        if rank == 0:
            np.save(out, np.array(res, dtype=float))
    finally:
        dist.destroy_process_group()


def test_python_driver_agrees_on_one_halo(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "halo.npy")
    mp.spawn(_halo_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    r = np.load(out)
    assert len(set(r[:, 0])) == 1 and len(set(r[:, 1])) == 1          # one width, one exchange kind on every rank
    assert len(set(r[:, 2])) > 1 and r[0, 0] == r[:, 2].max()         # ... although the local bounds differ


# ---- base class over the Python-driven schedule -----------------------------------------------------
def _hardinf_worker(rank, world, port, out):
    import torch.distributed as dist
    from pyro_amd import parallel
    from pyro_amd.planning import dynamicprogramming
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        cfg = _build("pendulum:21,11:3:float64")
        seen = {}

        class Recording(OracleSlab):
            def __init__(self, grid_sys, cost, dtype, rows, halo, device, hard_inf=False):
                seen["hard_inf"] = hard_inf
                super().__init__(grid_sys, cost, dtype, rows, halo, device)
        with contextlib.redirect_stdout(io.StringIO()):
            dp = dynamicprogramming.DynamicProgramming(cfg["grid_sys"], cfg["cf"], comm=parallel.TorchDistComm(dist, slab_factory=Recording))
        base_flag = seen["hard_inf"]
        refused = False
        try:        # a back end without the keyword cannot run the base-class recursion: refused, not silently LUT semantics
            with contextlib.redirect_stdout(io.StringIO()):
                dynamicprogramming.DynamicProgramming(cfg["grid_sys"], cfg["cf"], comm=parallel.TorchDistComm(dist, slab_factory=OracleSlab))
        except NotImplementedError:
            refused = True
        with contextlib.redirect_stdout(io.StringIO()):
            lut = dynamicprogramming.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"],
                                                                       comm=parallel.TorchDistComm(dist, slab_factory=Recording))
        lut_flag = seen["hard_inf"]
        np.save(out, np.array([bool(dp.HARD_INF), base_flag, refused, lut_flag], dtype=float))
    finally:
        dist.destroy_process_group()


def test_base_class_flag_reaches_the_python_driven_slabs(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "hi.npy")
    mp.spawn(_hardinf_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    r = np.load(out)
    assert r[0] == 1 and r[1] == 1 and r[2] == 1 and r[3] == 0


def test_torch_dist_comm_refuses_the_table_tier():
    """The Python-driven schedule has no table tier: a system without in-kernel dynamics must raise, not run 'fused'."""
    from pyro_amd import parallel

    class FakeDist:
        def get_rank(self): return 0
        def get_world_size(self): return 1

    class DP:
        grid_sys = _contracting_grid()
        cf = None
        dtype, device, HARD_INF = "float64", 0, False
    with pytest.raises(NotImplementedError):
        parallel._TorchEngine(DP(), parallel.TorchDistComm(FakeDist()))
