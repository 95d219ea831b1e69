"""Bank-conflict model of the 4-D lean sweep's PAIR window (ds_read_b64; development tool, CPU only).

A wave's lanes are nodes of a velocity-plane tile; a lane's gather for action a reads the 8-byte slot
(row iv0 + c0(a), column iv1 + c1(a)) of its window plane.  gfx950 services a ds_read_b64 in two groups of 32 lanes, one LDS
cycle per group when no two DISTINCT slots of the group fall on the same pair of banks (slot mod 32), N cycles for N of them
(MI355X_MICROARCH.md, LDS).  The model counts cycles per wave-instruction (ideal 2) for the C3 cart-pole grid under
 * the lane map:   dense (lane s -> row s / ncols, column s % ncols) or padded to W = 32 / 64 lanes per tile row
 * the row pitch:  a fixed multiple of 4 >= window width, a multiple of 32, or congruent to ncols modulo 32
and reports cycles x waves (the LDS time of the tile's action loop, relative).

Round 5 adds the TRANSPOSED lane map (`model_t`): on the cart-pole the displacement of a node does not depend on its index along
axis 2 (cartpole.py:369-437: the acceleration is a function of theta, dtheta and the input), so lanes that run along axis 2
-- window stored with axis 2 fastest -- all shift by the same amount and a 32-lane group that lies in ONE window column reads 32
consecutive slots: no conflict at any pitch.  Groups that straddle two columns are conflict-free when the column pitch is
congruent to the tile's rows modulo 32.  Not built: the model says what the layout would buy.

    python tools/lds_conflict_model4.py
"""
import numpy as np

N = 101
A = 21
DT = 0.05
LB, UB = -2 * np.pi, 2 * np.pi
DX = (UB - LB) / (N - 1)
LEV = np.linspace(LB, UB, N)
ULEV = np.linspace(-20, 20, A)
M1, M2, LCG, GRAV = 1.0, 0.1, 0.5, 9.81


def ddq(q1, dq1, u):
    h00, h01, h11 = M1 + M2, M2 * LCG * np.cos(q1), M2 * LCG ** 2
    c01 = -M2 * LCG * np.sin(q1) * dq1
    g1 = M2 * GRAV * LCG * np.sin(q1)
    r0 = u - c01 * dq1
    r1 = -g1 + 0 * u
    det = h00 * h11 - h01 * h01
    return (h11 * r0 - h01 * r1) / det, (-h01 * r0 + h00 * r1) / det


def cells(i1, r2, r3):
    i2, i3 = np.meshgrid(r2, r3, indexing="ij")
    q1, v0, v1 = LEV[i1], LEV[i2], LEV[i3]
    a0, a1 = ddq(q1, v1[..., None], ULEV[None, None, :])
    c2 = (v0[..., None] + a0 * DT - LB) / DX
    c3 = (v1[..., None] + a1 * DT - LB) / DX
    inb = (c2 >= 0) & (c2 <= N - 1) & (c3 >= 0) & (c3 <= N - 1)
    return np.clip(np.floor(c2).astype(int), 0, N - 2), np.clip(np.floor(c3).astype(int), 0, N - 2), inb


def group_cycles(slots, live):
    tot = 0
    for g in (slice(0, 32), slice(32, 64)):
        s = np.unique(slots[g][live[g]])
        if s.size:
            tot += np.bincount(s % 32, minlength=32).max()
    return tot


def model(rows, ncols, lane_w, pitch, samples=60, seed=1):
    """tiles of rows x ncols nodes; lane_w: 0 dense, else lanes per tile row; pitch(window width, ncols) -> slots per row"""
    rng = np.random.default_rng(seed)
    cyc = instr = waves = nodes = 0
    lds = 0
    for _ in range(samples):
        i1 = rng.integers(0, N)
        t2 = rng.integers(0, (N + rows - 1) // rows) * rows
        t3 = rng.integers(0, (N + ncols - 1) // ncols) * ncols
        r2, r3 = np.arange(t2, min(N, t2 + rows)), np.arange(t3, min(N, t3 + ncols))
        f2, f3, inb = cells(i1, r2, r3)
        if not inb.any():
            continue
        lo2, lo3 = f2[inb].min(), f3[inb].min()
        w3 = f3[inb].max() + 2 - lo3
        rs = pitch(w3, len(r3))
        lds = max(lds, (f2[inb].max() + 2 - lo2) * rs * 8)
        slot = (f2 - lo2) * rs + (f3 - lo3)                      # [n2, n3, A]
        n2, n3 = len(r2), len(r3)
        W = lane_w if lane_w else n3
        lane_node = np.full(((n2 * W + 63) // 64) * 64, -1)
        for j in range(n2):
            lane_node[j * W:j * W + n3] = j * n3 + np.arange(n3)
        slot = slot.reshape(n2 * n3, A)
        live = inb.reshape(n2 * n3, A)
        nodes += n2 * n3
        for wv in range(len(lane_node) // 64):
            ids = lane_node[wv * 64:(wv + 1) * 64]
            ok = ids >= 0
            if not ok.any():
                continue
            waves += 1
            for a in range(A):
                lv = ok & live[np.maximum(ids, 0), a]
                if not lv.any():
                    continue
                cyc += group_cycles(slot[np.maximum(ids, 0), a], lv)
                instr += 1
    return cyc / instr, waves * 64 / nodes, cyc / nodes * 64 / A, lds


def model_t(rows, ncols, pitch, samples=60, seed=1):
    """lanes along axis 2 (lane s -> column s / rows, row s % rows); LDS slot = window column * pitch + window row;
    pitch(window rows, tile rows) -> slots per window column"""
    rng = np.random.default_rng(seed)
    cyc = instr = waves = nodes = 0
    lds = 0
    for _ in range(samples):
        i1 = rng.integers(0, N)
        t2 = rng.integers(0, (N + rows - 1) // rows) * rows
        t3 = rng.integers(0, (N + ncols - 1) // ncols) * ncols
        r2, r3 = np.arange(t2, min(N, t2 + rows)), np.arange(t3, min(N, t3 + ncols))
        f2, f3, inb = cells(i1, r2, r3)
        if not inb.any():
            continue
        lo2, lo3 = f2[inb].min(), f3[inb].min()
        ps = pitch(f2[inb].max() + 2 - lo2, len(r2))
        lds = max(lds, (f3[inb].max() + 2 - lo3) * ps * 8)
        slot = ((f3 - lo3) * ps + (f2 - lo2)).reshape(len(r2) * len(r3), A)
        live = inb.reshape(len(r2) * len(r3), A)
        n2, n3 = len(r2), len(r3)
        lane_node = np.full(((n2 * n3 + 63) // 64) * 64, -1)
        lane_node[:n2 * n3] = (np.arange(n2)[None, :] * n3 + np.arange(n3)[:, None]).reshape(-1)   # column-major over the tile
        nodes += n2 * n3
        for wv in range(len(lane_node) // 64):
            ids = lane_node[wv * 64:(wv + 1) * 64]
            ok = ids >= 0
            if not ok.any():
                continue
            waves += 1
            for a in range(A):
                lv = ok & live[np.maximum(ids, 0), a]
                if not lv.any():
                    continue
                cyc += group_cycles(slot[np.maximum(ids, 0), a], lv)
                instr += 1
    return cyc / instr, waves * 64 / nodes, cyc / nodes * 64 / A, lds


if __name__ == "__main__":
    mul4 = lambda w, n: max(4, (w + 3) // 4 * 4)
    mul32 = lambda w, n: (w + 31) // 32 * 32
    cong = lambda w, n: next(r for r in range(mul4(w, n), mul4(w, n) + 33) if r % 32 == n % 32)
    print("%-44s %8s %8s %10s %8s" % ("layout", "clk/b64", "lanes/nd", "clk/nd-act", "LDS B"))
    for name, rows, ncols, lw, pitch in [
        ("10x51 dense, pitch mult of 4 (production)", 10, 51, 0, mul4),
        ("10x51 dense, pitch = ncols mod 32", 10, 51, 0, cong),
        ("8x51 padded to 64 lanes, pitch mult of 32", 8, 51, 64, mul32),
        ("19x26 dense, pitch mult of 4", 19, 26, 0, mul4),
        ("19x26 dense, pitch = ncols mod 32", 19, 26, 0, cong),
        ("16x26 padded to 32 lanes, pitch mult of 32", 16, 26, 32, mul32),
        ("16x26 padded to 32 lanes, pitch mult of 4", 16, 26, 32, mul4),
        ("16x32 dense = padded, pitch mult of 32", 16, 32, 32, mul32),
        ("16x32 dense = padded, pitch mult of 4", 16, 32, 32, mul4),
        ("8x64 dense = padded, pitch mult of 32", 8, 64, 64, mul32),
        ("15x34 dense, pitch mult of 4", 15, 34, 0, mul4),
        ("15x34 dense, pitch = ncols mod 32", 15, 34, 0, cong),
    ]:
        c, l, t, b = model(rows, ncols, lw, pitch)
        print("%-44s %8.2f %8.2f %10.2f %8d" % (name, c, l, t, b))
    odd = lambda w, n: w | 1
    congt = lambda w, n: next(r for r in range(w, w + 33) if r % 32 == n % 32)
    for name, rows, ncols, pitch in [
        ("T 10x51, pitch odd", 10, 51, odd),
        ("T 10x51, pitch = rows mod 32", 10, 51, congt),
        ("T 19x26, pitch odd", 19, 26, odd),
        ("T 19x26, pitch = rows mod 32", 19, 26, congt),
        ("T 16x32, pitch = rows mod 32", 16, 32, congt),
        ("T 32x16, pitch odd (a group = one column)", 32, 16, odd),
        ("T 34x15, pitch = rows mod 32", 34, 15, congt),
    ]:
        c, l, t, b = model_t(rows, ncols, pitch)
        print("%-44s %8.2f %8.2f %10.2f %8d" % (name, c, l, t, b))
    # the cart-pole with q = (theta, x) (tools/tools_swapped_cartpole.py): tile rows run along dtheta, lanes along dx -- TODAY'S kernel and
    # lane map, the displacement now the same for every lane of a tile row
    reference_order = cells
    def cells_swapped(i1, r2, r3):
        f2, f3, inb = reference_order(i1, r3, r2)
        return f3.transpose(1, 0, 2), f2.transpose(1, 0, 2), inb.transpose(1, 0, 2)
    globals()["cells"] = cells_swapped
    for name, rows, ncols, pitch in [
        ("swapped q: 10x51, pitch = ncols mod 32", 10, 51, cong),
        ("swapped q: 19x26, pitch = ncols mod 32", 19, 26, cong),
        ("swapped q: 15x34, pitch = ncols mod 32", 15, 34, cong),
        ("swapped q: 10x51, pitch mult of 4", 10, 51, mul4),
    ]:
        c, l, t, b = model(rows, ncols, 0, pitch)
        print("%-44s %8.2f %8.2f %10.2f %8d" % (name, c, l, t, b))
