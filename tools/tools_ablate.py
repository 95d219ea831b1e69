"""Timing ablations of k_sweep_lean (PVI_DBG bits; results are wrong on purpose): kernel time per sweep from HIP events.
usage: tools_ablate.py <workload> [sweeps]"""
import contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyro_amd import configs
name, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 200
with contextlib.redirect_stdout(io.StringIO()):
    cfg = configs.build(name)
    p = cfg["grid_sys"]._device_problem(cost=cfg["cf"].device_cost(), dtype=cfg["dtype"])
p.terminal_cost()
try:
    p.sweep(max(2, n // 10), 1.0, -1.0)
except RuntimeError:
    pass
best = 1e9
for _ in range(3):
    try:
        p.sweep(n, 1.0, -1.0)
    except RuntimeError:                 # PVI_DBG & 8: the sweeps do not record themselves; the events still timed them
        pass
    best = min(best, p.last_sweep_ms() / n)
print("%s PVI_DBG=%s %.2f us/sweep  %s" % (name, os.environ.get("PVI_DBG", "0"), best * 1e3, p.describe().split(" note=")[0]))
