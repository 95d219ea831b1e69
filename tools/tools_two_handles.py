# two handles whose lean windows need different amounts of LDS (> 48 KB), used alternately
import sys, io, contextlib
sys.path.insert(0, "/root/repo")
import numpy as np
from pyro_amd import configs
from pyro_amd.planning import dynamicprogramming
with contextlib.redirect_stdout(io.StringIO()):
    a = configs.build("cartpole:61,61,61,61:21:float32"); b = configs.build("c3")
    B = dynamicprogramming.DynamicProgrammingWithLookUpTable(b["grid_sys"], b["cf"], dtype="float32")   # larger window first
    A = dynamicprogramming.DynamicProgrammingWithLookUpTable(a["grid_sys"], a["cf"], dtype="float32")
print(A._p.describe()[:90]); print(B._p.describe()[:90])
for k in range(3):
    A._p.sweep(2, 1.0, -1.0); B._p.sweep(2, 1.0, -1.0)
print("ok", float(A._p.get_J().max()), float(B._p.get_J().max()))
