"""timing of the bicubic-spline mode against the bilinear kernels (pendulum grids)"""
import sys
sys.path.insert(0, "/root/repo")
from pyro_amd import configs
for spec in ("pendulum:1001,1001:51:float64", "pendulum:1001,1001:51:float32", "pendulum:201,201:201:float64", "pendulum:101,101:11:float64"):
    cfg = configs.build(spec)
    cost = cfg["cf"].device_cost()
    for mode in ("linear", "bicubic"):
        p = cfg["grid_sys"]._device_problem(cost=cost, dtype=cfg["dtype"], device=0)
        p.set_interpolation(mode)
        p.terminal_cost()
        p.sweep(5, 1.0, -1.0)
        p.sweep(50, 1.0, -1.0)
        print("%-34s %-8s %.4f ms/sweep" % (spec, mode, p.last_sweep_ms() / 50), flush=True)
        p.close()
