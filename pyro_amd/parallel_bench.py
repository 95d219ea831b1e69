"""bench.py for N > 1: one process per GPU (torchrun), axis-0 slabs, halo exchange + statistics all-reduce over RCCL.

Headline (weak scaling, the family the N = 1 line belongs to): cart-pole, 101^3 x 21 actions per row of axis 0 and
100 N + 1 rows on a rail N times as long -- the same spacing on every axis as BASELINE configs[2] (C3), every rank owns
100-101 rows, N = 1 IS C3.  Secondary: BASELINE configs[3] (C4, 151^4 x 31) split over the same ranks (strong scaling,
with its one-GPU rate measured on rank 0 beside it).
"""
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

MIN_REGION_S = 0.3


def _quiet_build(name, **kw):
    from pyro_amd import configs
    with contextlib.redirect_stdout(io.StringIO()):
        return configs.build(name, **kw)


def _one_gpu_rate(cfg, local, sweeps):
    """The same grid on ONE GPU (rank 0 only): cells/s."""
    from pyro_amd.planning import dynamicprogramming
    g = cfg["grid_sys"]
    with contextlib.redirect_stdout(io.StringIO()):
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, cfg["cf"], dtype=cfg["dtype"], device=local)
    p = dp._p
    p.sweep(2, 1.0, -1.0)
    p.synchronize()
    t0 = time.perf_counter()
    p.sweep(sweeps, 1.0, -1.0)
    p.synchronize()
    rate = g.nodes_n * g.actions_n * sweeps / (time.perf_counter() - t0)
    p.close()
    return rate


def _barrier(dist, torch):
    """Host-side barrier (a one-element all-reduce of a CPU tensor: the gloo half of the process group)."""
    dist.all_reduce(torch.zeros(1, dtype=torch.int32))


def _agree(dist, torch, ok):
    """True when every rank says ok (host-side all-reduce)."""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


class _Driver:
    """One sharded solver behind run(n) -> statistics of the last sweep; in-library RCCL first, torch.distributed's nccl
    backend (the Python-driven schedule of pyro_amd.parallel.ShardedValueIteration) when that cannot start on some rank."""

    def __init__(self, cfg, dist, torch, rank, world, local, via_torch):
        from pyro_amd import _native, parallel
        g = cfg["grid_sys"]
        self.fallback_reason = None
        self.vi = None
        if not via_torch:
            # Everything that can fail on ONE rank without a peer -- the library, RCCL's dlopen, the device -- is tried
            # and agreed on BEFORE the first RCCL collective (ncclCommInitRank inside pvi_shard_create): a rank that
            # cannot get there must not leave the others waiting in it.  pvi_shard_create itself then agrees on the halo
            # and on every rank having built its slab, so its failures are seen by all ranks too.
            err, my_id = None, None
            try:
                my_id = _native.comm_unique_id()          # loads librccl and talks to the device on EVERY rank
                if _native.device_count() <= local:
                    raise RuntimeError("no HIP device %d" % local)
            except Exception as e:                        # noqa: BLE001
                err = "%s: %s" % (type(e).__name__, e)
            if _agree(dist, torch, err is None):
                try:
                    ids = [my_id if rank == 0 else None]
                    dist.broadcast_object_list(ids, src=0)
                    self.vi = parallel.RcclValueIteration(g, cfg["cf"], rank, world, comm_id=ids[0], dtype=cfg["dtype"], device=local)
                    self.vi.run(1, 1.0, -1.0)             # one whole sweep: exchange + all-reduce have run
                except Exception as e:                    # noqa: BLE001 -- a failure here is collective (see above)
                    err = "%s: %s" % (type(e).__name__, e)
            if _agree(dist, torch, err is None):
                vi = self.vi
                self.run = lambda n: list(vi.run(n, 1.0, -1.0)[0])
                desc = vi.describe()
                self.halo, self.p2p, self.overlap, self.describe = vi.halo, "send/recv" in desc, "+overlap" in desc, vi.describe
                self.collectives = "RCCL inside libpyrovi (pvi_shard_*)"
                self.timing = vi.shard.timing
                self.rccl_ranks = int(desc.split("rccl_ranks=")[1].split()[0]) if "rccl_ranks=" in desc else None
                return
            errs = [None] * world
            dist.all_gather_object(errs, err)
            self.fallback_reason = next((e for e in errs if e), "unknown")
            if self.vi is not None:
                with contextlib.suppress(Exception):
                    self.vi.close()
        vi = parallel.ShardedValueIteration(g, cfg["cf"], dist, dtype=cfg["dtype"], device=local)
        self.vi = vi
        self.run = lambda n: vi.run(n, 1.0, -1.0)
        self.halo, self.p2p, self.overlap, self.describe = vi.halo, vi.p2p, vi.overlap, vi.slab.describe
        self.collectives = "torch.distributed (nccl)"
        self.timing, self.rccl_ranks = None, None

    def close(self):
        with contextlib.suppress(Exception):
            if hasattr(self.vi, "close"):
                self.vi.close()
            elif hasattr(self.vi, "slab"):
                self.vi.slab.close()


def _selftest(dist, torch, rank, world, local, via_torch, name="cartpole:41,13,17,15:7:float32", sweeps=6):
    """`sweeps` backups of a small cart-pole over all ranks against the same backups on rank 0 alone: J, pi and the
    statistics of the last sweep must be the same bits (same kernels, same arithmetic per node; only the rows a rank does
    not own travel).  Returns {"ok": bool, ...}; every rank takes part, rank 0 holds the verdict."""
    cfg = _quiet_build(name, world=world)
    # The collective section: every step in it is entered by all ranks or (a failure the library's own agreements report on
    # every rank) by none.  What follows it is LOCAL to rank 0; an exception there is caught and travels with the verdict, so
    # that the other ranks are never left waiting in the broadcast below.
    drv = _Driver(cfg, dist, torch, rank, world, local, via_torch)
    st = drv.run(sweeps - 1)                 # (the driver's constructor ran one sweep already when it used the library's RCCL path)
    if drv.fallback_reason is None and hasattr(drv.vi, "shard"):
        J, pi = drv.vi.shard.gather_J(False), drv.vi.shard.gather_pi()
        done = sweeps
    else:                                    # the Python-driven schedule starts from J0
        st = drv.run(1)
        J, pi = drv.vi.gather()
        done = sweeps
    drv.close()
    res = {"ok": True, "grid": name, "sweeps": done, "ranks": world}
    if rank == 0:
        try:
            from pyro_amd.planning import dynamicprogramming
            with contextlib.redirect_stdout(io.StringIO()):
                dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"], dtype=cfg["dtype"], device=local)
            stats, _ = dp._p.sweep(done, 1.0, -1.0)
            J1, pi1 = dp._p.get_J(), dp._p.get_pi()
            dp._p.close()
            res["ok"] = bool(np.array_equal(J, J1) and np.array_equal(pi, pi1) and np.allclose(st[:3], stats[-1][:3], rtol=1e-12, atol=0))
            if not res["ok"]:
                res["max_abs_diff_J"] = float(np.abs(np.asarray(J) - J1).max())
                res["pi_mismatches"] = int((np.asarray(pi) != pi1).sum())
        except Exception as e:                            # noqa: BLE001
            res["ok"] = False
            res["error"] = "rank 0, one-GPU reference: %s: %s" % (type(e).__name__, e)
    flag = [res["ok"] if rank == 0 else None]
    dist.broadcast_object_list(flag, src=0)
    res["ok"] = bool(flag[0])
    return res


def _timed(drv, dist, torch, steps, warmup):
    """W warm-up sweeps, then batches of exactly K sweeps, each bracketed by barrier + synchronize on both sides and
    reduced with max over the ranks, until the region reaches MIN_REGION_S (every rank takes the same decisions: they
    are made on the all-reduced times)."""
    sync = torch.cuda.synchronize if torch.cuda.is_available() else (lambda: None)   # (CPU: the gloo test of this harness)
    if warmup:
        drv.run(warmup)
    sync()
    batches, elapsed, st = 0, 0.0, None
    while batches == 0 or elapsed < MIN_REGION_S:
        _barrier(dist, torch)
        t0 = time.perf_counter()
        st = drv.run(steps)                    # fixed sweep count: statistics (one all-reduce) for the last sweep only
        sync()                                 # (pvi_shard_sweep has synchronised its own streams already)
        _barrier(dist, torch)
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed += float(t.item())
        batches += 1
    return elapsed, batches, st


def _rank_timing(drv, dist, world):
    """Every rank's per-sweep GPU times of the last timed batch (pvi_shard_timing: HIP events on the rank's compute and
    comm streams), gathered so that rank 0 can print them: kernel time split into boundary and interior pieces, the
    exchange, and the part of the exchange the interior kernel did not hide."""
    mine = drv.timing() if drv.timing else None
    out = [None] * world
    dist.all_gather_object(out, mine)
    if any(t is None for t in out):
        return None
    keys = ("boundary_ms", "interior_ms", "exchange_ms", "exposed_exchange_ms", "sweep_ms")
    return {"source": "HIP events per rank (pvi_shard_timing), mean over the last %d sweeps of the last batch" % out[0]["sweeps_timed"],
            **{"kernel_" + k if k in ("boundary_ms", "interior_ms") else k: [round(float(t[k]), 4) for t in out] for k in keys}}


def _fragment(cfg, drv, world, steps, warmup, elapsed, batches, st, setup_s, per_rank=None):
    g = cfg["grid_sys"]
    N, A = g.nodes_n, g.actions_n
    w = 4 if cfg["dtype"] == "float32" else 8
    timed = steps * batches
    alg = N * (2 * w + (1 if A <= 256 else 2))
    par = "axis-0 slabs x%d, halo %d rows, %s%s, collectives: %s" % (
        world, drv.halo, "p2p send/recv" if drv.p2p else "slab broadcast",
        ", exchange overlapped with the interior kernel" if drv.overlap else "", drv.collectives)
    out = {
        "value": N * A * timed / elapsed, "unit": "cells/s", "steps": steps, "warmup": warmup,
        "batches": batches, "timed_steps": timed, "timed_region_s": elapsed, "ms_per_step": elapsed / timed * 1e3,
        "sweeps_per_sec": timed / elapsed, "setup_ms": setup_s * 1e3, "dtype": "f32" if w == 4 else "f64",
        "config": {"workload": "%s: %s" % (cfg["name"], cfg["description"]), "nodes": N, "actions": A,
                   "cells_per_sweep": N * A, "dt": g.dt, "alpha": 1.0, "parallelism": par},
        "roofline": {"bound": "hbm", "achieved": alg * timed / elapsed / 1e9, "peak": 8000.0 * world, "unit": "GB/s",
                     "frac": alg * timed / elapsed / 1e9 / (8000.0 * world), "traffic": None,
                     "algorithmic_bytes_per_launch": alg,
                     "note": "whole-step rate of all ranks incl. halo exchange and all-reduce against N x 8 TB/s; "
                             "per-kernel figures and counters are in the N=1 line"},
        "last_stats": [float(v) for v in st], "kernel_path": drv.describe(),
        "rccl_ranks": drv.rccl_ranks, "per_rank": per_rank,
    }
    if per_rank:
        k = [b + i for b, i in zip(per_rank["kernel_boundary_ms"], per_rank["kernel_interior_ms"])]
        out["kernel_ms_max_rank"] = max(k)
        out["exposed_exchange_ms_max_rank"] = max(per_rank["exposed_exchange_ms"])
        # share of the exchange hidden behind the interior kernel; undefined when there is nothing to exchange (one rank:
        # both numbers are event noise)
        ex = max(per_rank["exchange_ms"])
        out["overlap_efficiency"] = (max(0.0, 1.0 - max(per_rank["exposed_exchange_ms"]) / ex) if ex >= 0.05 else None)
    if drv.fallback_reason:
        out["in_library_rccl_error"] = drv.fallback_reason
    return out


def run(args):
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    has_cuda = torch.cuda.is_available()      # (False only in the CPU tests of this harness: tests/test_emu_cpu.py)
    if has_cuda:
        torch.cuda.set_device(local)
    # gloo and RCCL print banners on stdout from C: the ONE JSON line goes to the real stdout, everything else to stderr
    import sys
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    # torch.distributed is the launcher-side harness (rendezvous, the communicator id, barriers, max of the timed region,
    # all on host tensors over gloo): the data path -- halo exchange and statistics all-reduce -- is RCCL inside libpyrovi
    # (pvi_shard_*).  The process group also carries an nccl backend for device tensors; it is only ever initialised when
    # PVI_TORCH_COLLECTIVES=1 selects the Python-driven schedule, or when the in-library communicator fails to start.
    via_torch = bool(int(os.environ.get("PVI_TORCH_COLLECTIVES", "0")))
    dist.init_process_group("cpu:gloo,cuda:nccl" if world > 1 and has_cuda else "gloo", rank=rank, world_size=world)
    steps = args.steps if args.steps is not None else 20
    warmup = args.warmup if args.warmup is not None else 2
    headline = args.workload or "c3w"
    out = None

    def one(name, steps, warmup, ref_sweeps):
        cfg = _quiet_build(name, world=world)
        ref = None
        if rank == 0 and not args.no_cpu and ref_sweeps:
            ref_cfg = _quiet_build("c3") if name == "c3w" else cfg      # c3w: per-GPU work = C3
            ref = _one_gpu_rate(ref_cfg, local, ref_sweeps)
        t0 = time.perf_counter()
        drv = _Driver(cfg, dist, torch, rank, world, local, via_torch)
        if has_cuda:
            torch.cuda.synchronize()
        setup_s = time.perf_counter() - t0
        elapsed, batches, st = _timed(drv, dist, torch, steps, warmup)
        per_rank = _rank_timing(drv, dist, world)
        frag = _fragment(cfg, drv, world, steps, warmup, elapsed, batches, st, setup_s, per_rank)
        drv.close()
        return frag, ref

    # ---- self-test before anything is timed: a small grid through the SAME sharded path (slabs, halo exchange, statistics
    # all-reduce, gathers) must equal one rank's result bit for bit.  Until an 8-GPU node has run this, it is the first real
    # RCCL send / recv between ranks the library ever executes -- a wrong exchange should say so, not print a fast number.
    selftest = None
    if not getattr(args, "no_selftest", False):
        err = None
        try:
            grid = getattr(args, "selftest_grid", None)
            selftest = _selftest(dist, torch, rank, world, local, via_torch, **({"name": grid} if grid else {}))
        except Exception as e:                            # noqa: BLE001
            err = "%s: %s" % (type(e).__name__, e)
        errs = [None] * world
        dist.all_gather_object(errs, err)
        if any(errs):
            selftest = {"ok": False, "error": "; ".join("rank %d: %s" % (r, e) for r, e in enumerate(errs) if e)}
    frag, ref = one(headline, steps, warmup, max(3, steps // 4))
    if rank == 0:
        weak = headline == "c3w"
        out = {"metric": "vi_state_action_cell_updates_per_sec", "value": frag.pop("value"), "unit": frag.pop("unit"),
               "n_gpus": world, "steps": frag.pop("steps"), "warmup": frag.pop("warmup"),
               "ms_per_step": frag.pop("ms_per_step"), "higher_is_better": True,
               "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": frag.pop("dtype"),
               "data": "synthetic", "config": frag.pop("config")}
        out.update(frag)
        out["selftest"] = selftest
        if selftest is not None and not selftest.get("ok"):
            # a sharded result that differs from one GPU's is not a measurement of this path: no number in `value`
            out["value_unverified"] = out["value"]
            out["value"] = None
            out["invalid"] = "self-test failed: the sharded solve of %s differs from one rank's (see selftest)" % selftest.get("grid", "the small grid")
            print("bench.py: SELF-TEST FAILED -- value withheld: %r" % (selftest,), file=sys.stderr, flush=True)
        if weak:
            out["value_1gpu_c3"] = ref                # one rank's share of the work, alone on one GPU (N = 1 line)
        else:
            out["value_1gpu_same_workload"] = ref
            out["strong_scaling_speedup"] = out["value"] / ref if ref and out["value"] else None
    if args.workload is None and not args.no_secondary:
        # BASELINE configs[3] over the same ranks: strong scaling.  A secondary line must not take the headline down,
        # and the ranks must leave it TOGETHER: whatever one rank caught, all of them agree on before going on.
        frag, ref, err = None, None, None
        try:
            frag, ref = one("c4", max(5, steps // 2), 2, 3)
        except Exception as e:                            # noqa: BLE001
            err = "%s: %s" % (type(e).__name__, e)
        errs = [None] * world
        dist.all_gather_object(errs, err)
        if rank == 0:
            if any(errs):
                out["secondary"] = {"c4": {"error": "; ".join("rank %d: %s" % (r, e) for r, e in enumerate(errs) if e)}}
            else:
                frag["scaling"] = "strong"
                frag["value_1gpu_same_workload"] = ref
                frag["strong_scaling_speedup"] = frag["value"] / ref if ref else None
                out["secondary"] = {"c4": frag}
    if rank == 0:
        from pyro_amd import _build, _native, benchline
        if os.path.realpath(_native.LIB_PATH) != os.path.realpath(_build.OUT):     # (a sanitizer, experiment or test build)
            out["invalid"] = "PYROVI_LIB=%s: not the product library pyro_amd/libpyrovi.so" % _native.LIB_PATH
        benchline.emit(out, fd=real_stdout)       # compact headline on stdout, the full record on stderr / gpurun_out
    _barrier(dist, torch)
    dist.destroy_process_group()
