"""Cart-pole swing-up on a 4-D grid, sharded over the GPUs of one node: axis-0 slabs, halo exchange and statistics
all-reduce by RCCL INSIDE libpyrovi (pvi_shard_*); torch.distributed is only the launcher-side harness that hands
every rank the communicator id.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/cartpole_sharded.py
"""
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyro_amd import _native, parallel
from pyro_amd.analysis import costfunction
from pyro_amd.dynamic import cartpole
from pyro_amd.planning import discretizer

local = int(os.environ.get("LOCAL_RANK", "0"))
dist.init_process_group("gloo")                        # rendezvous only
rank, world = dist.get_rank(), dist.get_world_size()

sys_ = cartpole.CartPole()
grid_sys = discretizer.GridDynamicSystem(sys_, [101, 101, 101, 101], [21])
qcf = costfunction.QuadraticCostFunction.from_sys(sys_)
qcf.xbar = np.array([0, np.pi, 0, 0])
qcf.INF = 1000

ids = [_native.comm_unique_id() if rank == 0 else None]
dist.broadcast_object_list(ids, src=0)
vi = parallel.RcclValueIteration(grid_sys, qcf, rank, world, comm_id=ids[0], dtype="float32", device=local)
for k in range(10):
    (jmax, dmax, dmin, delta), n = vi.run(20, 1.0, -1.0)       # 20 sweeps per call, no host synchronisation in between
    if rank == 0:
        print("%d  max J %.3f  delta %.4f" % (20 * (k + 1), jmax, delta))
J, pi = vi.owned()                                             # this rank's rows
vi.close()
dist.destroy_process_group()
