import sys, io, contextlib
sys.path.insert(0, "/root/repo")
from pyro_amd import configs
from pyro_amd.planning import dynamicprogramming
for name in sys.argv[1:]:
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build(name)
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"], dtype=cfg["dtype"])
    print(name, dp._p.describe())
    dp._p.close()
