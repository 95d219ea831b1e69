#!/usr/bin/env python3
"""usage: slab_time.py <workload> <world> [rank ...]   -- what ONE rank of an axis-0 sharded solve costs per sweep, measured on
one GPU: the slab of rank r of `world` (owned rows + agreed halo, boundary pieces first, interior overlapped with the
exchange stream) is built through pvi_shard_create_with_transport with a transport that moves nothing, and pvi_shard_timing
reports the boundary and interior kernel times.  The exchange itself (xGMI) is not in it: this is the kernel side of the
strong-scaling estimate, not a scaling measurement."""
import contextlib, io, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from pyro_amd import configs, parallel

name, world = sys.argv[1], int(sys.argv[2])
ranks = [int(r) for r in sys.argv[3:]] or sorted({0, world // 2, world - 1})
with contextlib.redirect_stdout(io.StringIO()):
    cfg = configs.build(name, world=world) if name == "c3w" else configs.build(name)
g, cf = cfg["grid_sys"], cfg["cf"]
halo = parallel.halo_rows(g)
rows_total = g.x_grid_dim[0]
for rank in ranks:
    t0 = time.time()
    sh = g._shard_problem(rank, world, halo, cost=cf.device_cost(), dtype=cfg["dtype"],
                          transport=(lambda *a: 0, lambda v: 0))
    sh.terminal_cost()
    t1 = time.time()
    sh.sweep(3, 1.0, -1.0)
    sh.sweep(12, 1.0, -1.0)
    tm = sh.timing()
    b, i, whole, n = tm["boundary_ms"], tm["interior_ms"], tm["sweep_ms"], tm["sweeps_timed"]
    desc = sh.describe()
    print("%s world %d rank %d rows [%d, %d) of %d halo %d  setup %.2f s  boundary %.3f ms  interior %.3f ms  sweep %.3f ms  (%d sweeps)  %s"
          % (name, world, rank, sh.rows[0], sh.rows[1], rows_total, sh.halo, t1 - t0, b, i, whole, int(n),
             " ".join(w for w in desc.split() if w.startswith(("pieces", "tile=", "block=", "tiles_per"))) ), flush=True)
    sh.close()
