#!/usr/bin/env python3
"""LDS bank conflicts of the 4-D window sweep's gathers, counted on the production kernel's OWN addresses -- under emulation
(tests/emu, `build_emu.py --ldstrace`: every ds_read_b64 of lean4_gather reports its address; the k-th read of the 64 lanes of a wave
is one wave instruction) and under the bank model calibrated in round 4 (tools/ldsbank.hip, tools/lds_conflict_model4.py: a group
of 32 lanes takes one LDS cycle per distinct 8-byte slot that shares a bank pair).  Not a timing and not a hardware counter: it is
the quantity SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE measured as 0.42 on C3 in round 4 (profiles/r04_counters_c3.json), computed
exactly for a candidate layout BEFORE it can be measured.

The grid: BASELINE configs[2]'s steps and velocity planes (101 x 101: the production tile shapes apply unchanged) over a few
position nodes around chosen angles -- the displacement of a cart-pole node depends on (theta, dtheta, u) only, so a slice of the
position axes sees the same gather patterns as the whole grid does at those angles.

    python tools/emu_lds_conflicts.py [angle_index ...]       # reference order, swapped order, swapped + RS_CONG=1, pinned tiles
"""
import contextlib
import ctypes
import io
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))

if os.environ.get("PYROVI_LIB") is None:                  # build the traced library and re-enter with it selected
    import build_emu
    lib = build_emu.build(ldstrace=True)
    env = dict(os.environ, PYROVI_LIB=lib, PYTHONPATH=ROOT)
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))

import numpy as np  # noqa: E402

from pyro_amd import _native  # noqa: E402
from pyro_amd.analysis import costfunction  # noqa: E402
from pyro_amd.dynamic import cartpole  # noqa: E402
from pyro_amd.planning import discretizer  # noqa: E402
from pyro_amd.planning import dynamicprogramming as DP  # noqa: E402

L = ctypes.CDLL(_native.LIB_PATH)
N, A = 101, 21


def stats(reset=True):
    out = (ctypes.c_ulonglong * 3)()
    L.emu_lds_stats(out, int(reset))
    return [int(v) for v in out]


def full_grid():
    """BASELINE configs[2]: cart-pole 101^4 x 21, float32 (pyro_amd/configs.py `c3`)."""
    from pyro_amd import configs
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build("c3")
    return cfg["grid_sys"], cfg["cf"]


def run(g, cf, order, rows, **ov):
    """One sweep of a SLAB of the full grid -- axis-0 rows [rows[0], rows[1]) with four halo rows, the other three axes whole: the
    box is the full problem's, so no lane's position leaves it for want of neighbours (a sub-BOX pins such lanes to one address
    and flatters the count).  Reference order: axis 0 is x, which the dynamics do not depend on -- any three rows see every angle.
    Swapped order: axis 0 is theta -- `rows` picks the angles."""
    from pyro_amd.planning import permuted
    cost = DP.device_cost_of(cf, g.sys)
    kw = g._problem_kwargs(cost, "float32")
    if order == "swapped":
        kw = permuted.swap_problem_kwargs(kw)
    kw.update(rows=(int(rows[0]), int(rows[1])), halo=(4, 4))
    with contextlib.redirect_stdout(io.StringIO()), _native.overrides(TUNE="0", **ov):
        p = _native.Problem(**kw)
    p.terminal_cost()
    stats()
    p.sweep_async(1.0)
    p.sweep_stats()
    r, c, i = stats()
    tok = dict(t.split("=", 1) for t in p.describe().split() if "=" in t)
    p.close()
    return r, c, i, tok


if __name__ == "__main__":
    g, cf = full_grid()
    # theta = -2 pi ... 2 pi over 101 levels (index 50 = hanging down, 75 = upright): rows of the swapped order's axis 0
    th_rows = [(int(a), int(a) + 2) for a in sys.argv[1:]] or [(52, 54), (62, 64), (70, 72), (76, 78), (88, 90)]
    variants = [("reference order, 19 x 51 tiles (C3's choice)", "reference", {"L4PIN": "19/512/101"}),
                ("reference order, 15 x 34 tiles", "reference", {"TV0": "15", "TV1": "34"}),
                ("swapped order, 19 x 51 tiles", "swapped", {"L4PIN": "19/512/101"}),
                ("swapped order, 15 x 34 tiles", "swapped", {"TV0": "15", "TV1": "34"}),
                ("swapped order, 15 x 34 tiles, pitch = width mod 32", "swapped", {"TV0": "15", "TV1": "34", "RS_CONG": "1"}),
                ("swapped order, 16 x 32 tiles", "swapped", {"TV0": "16", "TV1": "32", "TV_EXACT": "1"})]
    plane32 = ""
    for name, order, ov in variants:
        tot = [0, 0, 0]
        tile = None
        for rows in ([(49, 52)] if order == "reference" else th_rows):
            r, c, i, tok = run(g, cf, order, rows, **ov)
            tile = tok.get("tile")
            tot = [a + b for a, b in zip(tot, (r, c, i))]
        r, c, i = tot
        print("%-52s%s tile %-6s wave reads %10d  LDS cycles per ds_read_b64 %.3f (2 = no conflict)  conflict share %.3f" % (
            name, plane32, tile, r, 2.0 * c / max(i, 1), (c - i) / max(c, 1)), flush=True)
