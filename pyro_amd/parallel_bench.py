"""bench.py for N > 1: one process per GPU (torchrun), axis-0 slabs, halo exchange over RCCL."""
import contextlib
import io
import json
import os
import time

import numpy as np


def run(args):
    import torch
    import torch.distributed as dist
    from pyro_amd import configs, parallel

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    torch.cuda.set_device(local)
    # torch.distributed is the launcher-side harness only (rendezvous, the communicator id, barrier and max of the timed
    # region): the data path -- halo exchange and statistics all-reduce -- is RCCL inside libpyrovi (pvi_shard_*).
    # PVI_TORCH_COLLECTIVES=1 selects the older Python-driven schedule over torch.distributed's nccl backend.
    via_torch = bool(int(os.environ.get("PVI_TORCH_COLLECTIVES", "0")))
    if via_torch:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    name = args.workload or "c4"
    steps = args.steps if args.steps is not None else 20
    warmup = args.warmup if args.warmup is not None else 2
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build(name)
    g = cfg["grid_sys"]
    N, A = g.nodes_n, g.actions_n
    w = 4 if cfg["dtype"] == "float32" else 8

    # single-GPU reference of the SAME workload on rank 0 (strong-scaling denominator)
    one_gpu = None
    if rank == 0 and not args.no_cpu:
        from pyro_amd.planning import dynamicprogramming
        with contextlib.redirect_stdout(io.StringIO()):
            dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, cfg["cf"], dtype=cfg["dtype"], device=local)
        p = dp._p
        p.sweep(2, 1.0, -1.0)
        p.synchronize()
        t0 = time.perf_counter()
        p.sweep(max(3, steps // 4), 1.0, -1.0)
        p.synchronize()
        one_gpu = N * A * max(3, steps // 4) / (time.perf_counter() - t0)
        p.close()
        del dp

    if via_torch:
        vi = parallel.ShardedValueIteration(g, cfg["cf"], dist, dtype=cfg["dtype"], device=local)
        run = lambda n: vi.run(n, 1.0, -1.0)
        halo, p2p, overlap, describe = vi.halo, vi.p2p, vi.overlap, vi.slab.describe
    else:
        from pyro_amd import _native
        ids = [_native.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        vi = parallel.RcclValueIteration(g, cfg["cf"], rank, world, comm_id=ids[0], dtype=cfg["dtype"], device=local)
        run = lambda n: list(vi.run(n, 1.0, -1.0)[0])
        desc = vi.describe()
        halo, p2p, overlap, describe = vi.halo, "send/recv" in desc, "+overlap" in desc, vi.describe
    if warmup:
        run(warmup)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    st = run(steps)                        # fixed sweep count: statistics of the last sweep only
    torch.cuda.synchronize()               # (pvi_shard_sweep has synchronised its own streams already)
    dist.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if via_torch else "cpu")
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    if rank == 0:
        pbytes = 1 if A <= 256 else 2
        alg = N * (2 * w + pbytes)
        out = {
            "metric": "vi_state_action_cell_updates_per_sec", "value": N * A * steps / dt, "unit": "cells/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if w == 4 else "f64", "data": "synthetic",
            "config": {"workload": "%s: %s" % (cfg["name"], cfg["description"]), "nodes": N, "actions": A,
                       "cells_per_sweep": N * A, "dt": g.dt, "alpha": 1.0,
                       "parallelism": "axis-0 slabs x%d, halo %d rows, %s%s, collectives: %s" % (
                           world, halo, "p2p send/recv" if p2p else "slab broadcast",
                           ", exchange overlapped with the interior kernel" if overlap else "",
                           "torch.distributed (nccl)" if via_torch else "RCCL inside libpyrovi (pvi_shard_*)")},
            "sweeps_per_sec": steps / dt,
            "roofline": {"bound": "hbm", "achieved": alg * steps / dt / 1e9, "peak": 8000.0 * world, "unit": "GB/s",
                         "frac": alg * steps / dt / 1e9 / (8000.0 * world), "traffic": None,
                         "note": "whole-step rate incl. halo exchange; per-kernel figures are in the N=1 line"},
            "value_1gpu_same_workload": one_gpu,
            "strong_scaling_speedup": (N * A * steps / dt) / one_gpu if one_gpu else None,
            "last_stats": [float(v) for v in st], "kernel_path": describe(),
        }
        print(json.dumps(out))
    dist.destroy_process_group()
