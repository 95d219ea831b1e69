"""How the oracle's C/OpenMP twin scales on this host: cells/s of a slice of a sweep at 1, 2, 4, ... threads.
Diagnostic for bench.py's cpu_baseline (writes one JSON line)."""
import contextlib, io, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_PROC_BIND", "spread")
os.environ.setdefault("OMP_PLACES", "cores")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
import numpy as np
import bench
from oracle import c_oracle as CO
from pyro_amd import configs

name = sys.argv[1] if len(sys.argv) > 1 else "cartpole:61,61,61,61:21:float32"
with contextlib.redirect_stdout(io.StringIO()):
    cfg = configs.build(name)
p = bench.oracle_problem(cfg)
c = CO.CProblem(p)
J = c.terminal_cost()
J, _ = c.sweep(J)
out = {"workload": name, "cpu_count": os.cpu_count(), "affinity": len(CO.ALLOWED_CPUS),
       "usable": bench.usable_cpus(), "physical_cores": CO.physical_cores(), "omp_max_threads": CO.max_threads(), "rates": {}}
try:
    out["cgroup_cpu_max"] = open("/sys/fs/cgroup/cpu.max").read().strip()
except OSError:
    pass
t, top = 1, CO.max_threads()
while True:
    n = min(p.nodes_n, 40000 * t)
    t0 = time.perf_counter()
    c.sweep(J, 1.0, 0, n, threads=t)
    el = time.perf_counter() - t0
    out["rates"][str(t)] = n * p.actions_n / el
    if t >= top:
        break
    t = min(top, t * 2)
print(json.dumps(out))
