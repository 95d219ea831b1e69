#!/usr/bin/env python3
"""VERDICT r5 next #7 (small grids): microseconds per sweep of the 2-D float32 window sweep as one launch per sweep against the
multi-sweep cooperative launch (k_sweep_leanm, opt-in: MULTI32=1) -- for the lane splits at which the multi-sweep launch APPLIES.
It needs every workgroup resident and at most 512 of them; the production choice for C2' (201 x 201 x 201 actions: 8 lanes per
node, 1 407 workgroups of 232 threads) does not qualify, LSPLIT=1 (2 lanes per node, 316 workgroups) and LSPLIT=0 (158) do: the
question on hardware is whether a batch without launch boundaries at fewer lanes per node beats one launch per sweep at eight.

    python tools/time_multi32.py [config ...]          # default: c2p pendulum:101,101:11:float32 pendulum:301,151:51:float32
"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pyro_amd import _native, configs
from pyro_amd.planning import dynamicprogramming as DP

N = int(os.environ.get("SWEEPS", "2000"))


def handle(name, **ov):
    with contextlib.redirect_stdout(io.StringIO()), _native.overrides(**ov):
        cfg = configs.build(name)
        dp = DP.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"], dtype="float32")
        dp._p.sweep(1, 1.0, -1.0)          # (the form of a handle's batches is decided at its first sweep)
    return dp._p


for name in sys.argv[1:] or ["c2p", "pendulum:101,101:11:float32", "pendulum:301,151:51:float32"]:
    ref = None
    for tag, ov in (("production choice", {}), ("LSPLIT=1", {"LSPLIT": "1", "MULTI": "0"}), ("LSPLIT=1 MULTI32=1", {"LSPLIT": "1", "MULTI32": "1"}),
                    ("LSPLIT=0", {"LSPLIT": "0", "MULTI": "0"}), ("LSPLIT=0 MULTI32=1", {"LSPLIT": "0", "MULTI32": "1"})):
        try:
            p = handle(name, **ov)
        except _native.NativeError as e:
            print("%-28s %-20s refused: %s" % (name, tag, str(e)[:100]))
            continue
        p.terminal_cost()
        p.sweep(N // 10 + 1, 1.0, -1.0)
        best = 1e30
        for _ in range(3):
            p.terminal_cost()
            t0 = time.perf_counter()
            p.sweep(N, 1.0, -1.0)
            best = min(best, time.perf_counter() - t0)
        J = p.get_J()
        if ref is None:
            ref = J
        tok = dict(t.split("=", 1) for t in p.describe().split() if "=" in t)
        print("%-28s %-20s %8.2f us/sweep  multi=%s lsplit=%s grid=%s block=%s kernel=%s  same bits as the production choice: %s" % (
            name, tag, best / N * 1e6, tok.get("multi"), tok.get("lsplit"), tok.get("grid"), tok.get("block"), tok.get("kernel", "")[:36],
            np.array_equal(J, ref)), flush=True)
        p.close()
