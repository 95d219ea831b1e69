"""Cart-pole swing-up on a 4-D grid, sharded over the GPUs of one node (axis-0 slabs, halo exchange over RCCL):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/cartpole_sharded.py
"""
import os
import numpy as np
import torch
import torch.distributed as dist

from pyro_amd.analysis import costfunction
from pyro_amd.dynamic import cartpole
from pyro_amd.planning import discretizer
from pyro_amd import parallel

local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device("cuda", local))

sys_ = cartpole.CartPole()
grid_sys = discretizer.GridDynamicSystem(sys_, [101, 101, 101, 101], [21])
qcf = costfunction.QuadraticCostFunction.from_sys(sys_)
qcf.xbar = np.array([0, np.pi, 0, 0])
qcf.INF = 1000

vi = parallel.ShardedValueIteration(grid_sys, qcf, dist, dtype="float32", device=local)
for k in range(200):
    jmax, dmax, dmin, delta = vi.sweep(1.0)
    if dist.get_rank() == 0 and k % 20 == 0:
        print("%d  max J %.3f  delta %.4f" % (k, jmax, delta))
dist.destroy_process_group()
