"""Tier-B (table) sweep timing: x_next / G tables of a pendulum grid built on the GPU by the fused handle, then swept
by k_sweep_table.  Reports ms/sweep and the HBM rate on the table bytes (N*A*(n*8 + 8) per sweep)."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from pyro_amd import configs, _native
spec = sys.argv[1] if len(sys.argv) > 1 else "pendulum:1001,1001:51:float64"
dtype = spec.split(":")[-1]
cfg = configs.build(spec)
g = cfg["grid_sys"]
p = g._device_problem(cost=cfg["cf"].device_cost(), dtype="float64", device=0)
xn, xo, ao, G = p.build_tables()
p.terminal_cost(); J0 = p.get_J()
p.sweep(3, 1.0, -1.0); Jref = p.get_J()
p.close()
s = g.sys
for ok in (None, (xo & ao)):
    h = _native.Problem(g.x_level, g.u_level, s.x_lb, s.x_ub, s.u_lb, s.u_ub, g.dt, dtype=dtype, dynamics_id=_native.DYN_TABLE,
                        table_inf=float(cfg["cf"].INF))
    h.set_tables(xn, G, ok)
    h.set_J(J0)
    h.sweep(3, 1.0, -1.0)
    err = np.abs(h.get_J() - Jref).max() / np.abs(Jref).max()
    h.sweep(20, 1.0, -1.0)
    ms = h.last_sweep_ms() / 20
    cells = G.size
    packed = "table-packed" in h.describe()
    n_ = xn.shape[2]
    rec = (4 + 4 * n_ + 4) if dtype == "float32" else (8 + 8 * n_ + 8)           # packed record (f64: padded offset)
    byt = cells * (rec if packed else (n_ * 8 + 8 + (1 if ok is not None else 0)))
    print("%s %s  %.3f ms/sweep  %.1f G cells/s  table stream %.0f GB/s  rel err vs fused %.1e" %
          (spec, "base" if ok is not None else "LUT ", ms, cells / ms / 1e6, byt / ms / 1e6, err), flush=True)
    h.close()
