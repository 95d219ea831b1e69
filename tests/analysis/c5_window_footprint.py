#!/usr/bin/env python3
"""VERDICT r5 next #8 ("C5: build the 8x8-patch J window ... in LDS"), decided on the CPU before a kernel is written: WHERE do the
gathers of the sparse float64 walk of BASELINE configs[4] (two-link arm, 101^4 nodes x 11x11 torques) land, per wave (an 8 x 8 patch
of the velocity plane of one position node -- k_sweep64's PATCH mapping) and per workgroup (4 waves), and how many bytes of J would
a window that serves them have to hold?

For sampled position nodes (i0, i1) and every velocity node and action, the cell of x_next by the CPU twin's own expressions
(oracle/vi_oracle.py cells(): analysis only -- nothing here is on the product path), then per wave:

  planes   distinct (c0, c1) position cells its lanes touch (each cell = 2 x 2 position rows, whole velocity planes of 82 KB);
  box      the bounding box of the velocity cells (c2, c3) of its in-box (lane, action) pairs, + 1 for the upper corners;
  window   planes' rows x box doubles: the smallest rectangular LDS window that serves every gather of the wave;
  unique   distinct 16-byte pairs actually read, against gathers issued (reuse inside the wave);
  lines    128-byte lines a wave asks its L1 for, summed over its trips -- the walk as built (every lane walks the set bits of ITS
           mask, two cells per trip: the lanes of a trip are at different actions) against a walk over the UNION of the wave's
           masks (all lanes at one action per trip; lanes whose cell is outside sit the trip out).

    python tests/analysis/c5_window_footprint.py [n_position_samples] [seed]
"""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))      # the repository root
import numpy as np

from oracle import vi_oracle as O
from pyro_amd import configs

nsamp = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
with contextlib.redirect_stdout(io.StringIO()):
    cfg = configs.build("c5")
s, g, cf = cfg["sys"], cfg["grid_sys"], cfg["cf"]
dyn_id, params = s.device_dynamics()
p = O.Problem(g.x_level, g.u_level, g.dt, dyn_id, np.array(params), cf.Q, cf.R, cf.S, cf.xbar, cf.ubar, float(cf.INF), float(cf.EPS),
              x_lb=s.x_lb, x_ub=s.x_ub, u_lb=s.u_lb, u_ub=s.u_ub)
dims = list(p.dims)
D0, D1, D2, D3 = dims
A = p.actions_n
np2, np3 = (D2 + 7) // 8, (D3 + 7) // 8


def cell_of(levels, x):
    n = len(levels)
    return np.clip(np.searchsorted(levels, x, side="right") - 1, 0, n - 2)


rows = []
walk = []
per_wg = []
tot_pairs = tot_in = 0
for k in range(nsamp):
    i0, i1 = int(rng.integers(0, D0)), int(rng.integers(0, D1))
    ids = ((i0 * D1 + i1) * D2 + np.arange(D2)[:, None]) * D3 + np.arange(D3)[None, :]
    xn, x_ok, a_ok, G = O.cells(p, ids.ravel())
    ok = (x_ok & a_ok).reshape(D2, D3, A)
    c = [cell_of(p.levels[d], xn[:, :, d]).reshape(D2, D3, A) for d in range(4)]
    waves = []
    for p2 in range(np2):
        for p3 in range(np3):
            sl = (slice(8 * p2, min(8 * p2 + 8, D2)), slice(8 * p3, min(8 * p3 + 8, D3)))
            m = ok[sl]
            n_in = int(m.sum())
            lanes = m.shape[0] * m.shape[1]
            if n_in == 0:
                waves.append(None)
                rows.append((lanes, 0, 0, 0, 0, 0, 0, 0, 0))
                continue
            cc = [c[d][sl][m] for d in range(4)]
            planes = len(set(zip(cc[0].tolist(), cc[1].tolist())))
            r0 = (int(cc[0].min()), int(cc[0].max()) + 1)
            r1 = (int(cc[1].min()), int(cc[1].max()) + 1)
            b2 = (int(cc[2].min()), int(cc[2].max()) + 1)
            b3 = (int(cc[3].min()), int(cc[3].max()) + 1)
            nrows = (r0[1] - r0[0] + 1) * (r1[1] - r1[0] + 1)
            win = nrows * (b2[1] - b2[0] + 1) * (b3[1] - b3[0] + 1) * 8
            # 16-byte pairs: (row0, row1, v2 row, c3) for the 8 corners of (c0, c1, c2)
            pairs = set()
            for d0 in (0, 1):
                for d1 in (0, 1):
                    for d2 in (0, 1):
                        key = (((cc[0] + d0) * D1 + (cc[1] + d1)) * D2 + (cc[2] + d2)) * D3 + cc[3]
                        pairs.update(key.tolist())
            trips = int(m.reshape(-1, A).sum(axis=1).max())
            mm = m.reshape(-1, A)
            cl = [c[d][sl].reshape(-1, A) for d in range(4)]

            def lines_of(sel_lane, sel_act):
                """distinct 128-byte lines of the 8 gathers of the (lane, action) pairs given"""
                k = [cl[d][sel_lane, sel_act] for d in range(4)]
                out = set()
                for d0 in (0, 1):
                    for d1 in (0, 1):
                        for d2 in (0, 1):
                            e = (((k[0] + d0) * D1 + (k[1] + d1)) * D2 + (k[2] + d2)) * D3 + k[3]
                            out.update((e >> 4).tolist())          # 16 doubles per line
                            out.update(((e + 1) >> 4).tolist())
                return len(out)
            # as built: trip t takes every lane's set bits 2t and 2t + 1
            order = np.argsort(~mm, axis=1, kind="stable")          # set bits first, ascending action
            cnt = mm.sum(axis=1)
            l_lane = 0
            for t in range(0, trips, 2):
                ln, ac = [], []
                for u in (t, t + 1):
                    has = np.nonzero(cnt > u)[0]
                    ln.append(has)
                    ac.append(order[has, u])
                l_lane += lines_of(np.concatenate(ln), np.concatenate(ac))
            l_union, union = 0, np.nonzero(mm.any(axis=0))[0]
            for a in union:
                has = np.nonzero(mm[:, a])[0]
                l_union += lines_of(has, np.full(len(has), a))
            walk.append((trips, len(union), l_lane, l_union))
            waves.append((r0, r1, b2, b3))
            rows.append((lanes, n_in, planes, nrows, (b2[1] - b2[0] + 1), (b3[1] - b3[0] + 1), win, len(pairs), trips))
            tot_pairs += len(pairs)
            tot_in += n_in
    # workgroups of 4 consecutive waves (256 threads)
    for w0 in range(0, len(waves), 4):
        ws = [w for w in waves[w0:w0 + 4] if w is not None]
        if not ws:
            continue
        lo = [min(w[j][0] for w in ws) for j in range(4)]
        hi = [max(w[j][1] for w in ws) for j in range(4)]
        per_wg.append(np.prod([hi[j] - lo[j] + 1 for j in range(4)]) * 8)

R = np.array(rows, dtype=np.float64)
busy = R[:, 1] > 0
print("two-link 101^4 x 121, dt %.2f: %d position nodes sampled, %d waves (8 x 8 velocity patches), %.1f %% with a cell in the box"
      % (p.dt, nsamp, len(R), 100.0 * busy.mean()))
print("in-box (lane, action) pairs per wave: mean %.0f of %d   trips of the busiest lane: mean %.1f max %d"
      % (R[busy, 1].mean(), 64 * A, R[busy, 8].mean(), R[busy, 8].max()))
print("distinct position cells (c0, c1) per wave: mean %.2f max %d   position rows spanned: mean %.1f max %d"
      % (R[busy, 2].mean(), R[busy, 2].max(), R[busy, 3].mean(), R[busy, 3].max()))
print("velocity bounding box per wave: %.1f x %.1f cells on average (of %d x %d), max %d x %d"
      % (R[busy, 4].mean(), R[busy, 5].mean(), D2, D3, R[busy, 4].max(), R[busy, 5].max()))
q = np.percentile(R[busy, 6], [50, 90, 99, 100]) / 1024.0
print("rectangular LDS window that serves a WAVE: median %.0f KB, 90 %% %.0f KB, 99 %% %.0f KB, max %.0f KB  (LDS: 160 KB per CU)"
      % tuple(q))
q = np.percentile(np.array(per_wg, dtype=np.float64), [50, 90, 100]) / 1024.0
print("                       ... a WORKGROUP of 4 waves: median %.0f KB, 90 %% %.0f KB, max %.0f KB" % tuple(q))
print("16-byte pairs: %.0f gathers issued per wave, %.0f distinct (reuse inside a wave %.2fx): a window would be filled with %.1fx the"
      " bytes\n               the wave then reads from it" % (8 * R[busy, 1].mean(), R[busy, 7].mean(), 8 * tot_in / max(tot_pairs, 1),
                                                           R[busy, 6].mean() / (16.0 * R[busy, 7].mean())))
W = np.array(walk, dtype=np.float64)
print("walk as built: %.1f trips of two cells (busiest lane %.1f set bits), %.0f lines per wave;  union walk: %.1f actions per wave, %.0f"
      " lines per wave (%.2fx fewer)" % (np.ceil(W[:, 0] / 2).mean(), W[:, 0].mean(), W[:, 2].mean(), W[:, 1].mean(), W[:, 3].mean(),
                                        W[:, 2].sum() / W[:, 3].sum()))
