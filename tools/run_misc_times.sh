#!/bin/bash
# timing of the paths outside the BASELINE configs: exact-f32 (two-link f32), n = 3 explicit systems, spline
cd /root/repo
python - <<'PY'
import time, contextlib, io, numpy as np
from pyro_amd import configs
from pyro_amd.planning import dynamicprogramming, discretizer
from pyro_amd.analysis import costfunction
def run(name, dp, n=10):
    p = dp._p
    p.sweep(2, 1.0, -1.0); p.synchronize()
    t0 = time.perf_counter(); p.sweep(n, 1.0, -1.0); p.synchronize()
    print("%s: %.3f ms/sweep  %s" % (name, (time.perf_counter()-t0)/n*1e3, p.describe()[:70]))
with contextlib.redirect_stdout(io.StringIO()):
    c = configs.build("twolink:51,51,51,51:11,11:float32")
    dp1 = dynamicprogramming.DynamicProgrammingWithLookUpTable(c["grid_sys"], c["cf"], dtype="float32")
run("twolink 51^4 x121 f32 (exact-f32)", dp1)
from pyro_amd.dynamic import drone, vehicle_steering, suspension
with contextlib.redirect_stdout(io.StringIO()):
    s = drone.ConstantSpeedHelicopterTunnel()
    g = discretizer.GridDynamicSystem(s, [101, 101, 101], [21])
    cf = costfunction.QuadraticCostFunctionWithDomainCheck.from_sys(s) if hasattr(costfunction.QuadraticCostFunctionWithDomainCheck, "from_sys") else costfunction.QuadraticCostFunction.from_sys(s)
    dp2 = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, cf)
run("helicopter 101^3 x21 f64 (k_sweep3)", dp2)
with contextlib.redirect_stdout(io.StringIO()):
    dp3 = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, cf, dtype="float32")
run("helicopter 101^3 x21 f32 (k_sweep3)", dp3)
PY
