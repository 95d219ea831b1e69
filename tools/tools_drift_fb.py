"""GPU: float32 iterates against float64 ones, sweep by sweep, with plain float32 storage and with error-feedback storage
(f32_feedback=True -> PVI_FLAG_F32_FEEDBACK, k_sweep_lean4fb).   usage: tools_drift_fb.py <workload> <sweeps> <every>"""
import contextlib, io, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from pyro_amd import configs
from pyro_amd.planning import dynamicprogramming
name, n, every = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dps = {}
for key, dt, fb in (("f64", "float64", False), ("f32", "float32", False), ("f32fb", "float32", True)):
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build(name)
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"], dtype=dt, f32_feedback=fb)
    dp.save_time_history = False; dp.verbose = False
    dps[key] = dp
    print(key, dp._p.describe()[:200], flush=True)
ms = {k: 0.0 for k in dps}
worst = {"f32": 0.0, "f32fb": 0.0}
for k in range(every, n + 1, every):
    for key, dp in dps.items():
        dp._p.synchronize(); t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            dp.compute_steps(every)
        dp._p.synchronize(); ms[key] += (time.perf_counter() - t0) * 1e3
    J64 = dps["f64"].J; m = np.abs(J64).max()
    e = {key: np.abs(dps[key].J - J64).max() / m for key in ("f32", "f32fb")}
    for key in e: worst[key] = max(worst[key], e[key])
    print("%5d  plain %.3e  feedback %.3e   maxJ %.2f" % (k, e["f32"], e["f32fb"], m), flush=True)
pm = {key: float((dps[key].pi != dps["f64"].pi).mean()) for key in ("f32", "f32fb")}
print("DRIFT %s sweeps %d  worst plain %.3e  worst feedback %.3e  pi mismatch plain %.5f feedback %.5f  ms/sweep f64 %.3f plain %.3f feedback %.3f  kernel %s"
      % (name, n, worst["f32"], worst["f32fb"], pm["f32"], pm["f32fb"], ms["f64"] / n, ms["f32"] / n, ms["f32fb"] / n,
         [t for t in dps["f32fb"]._p.describe().split() if t.startswith("kernel=")]))
