"""Bank-conflict model of the 4-D lean sweep's LDS gathers (development tool; CPU only).

For sampled workgroup tiles of the C3 cart-pole grid it forms the byte addresses the action loop's ds_read2_b32 pairs
touch (pyro_amd/csrc/sweep_lean.inc, lean_q4) and counts LDS cycles under the gfx950 bank rules
(MI355X_MICROARCH.md, LDS): a ds_read2_b32 is two accesses, each serviced per 32-lane half, bank = dword address mod 32,
identical addresses broadcast, N distinct addresses on one bank cost N cycles.  It then tries other layouts
(row pitch, tile shape, lane order) before any of them is written into the kernel.

    python tools/lds_conflict_model.py
"""
import itertools
import sys

import numpy as np

N = 101
A = 21
DT = 0.05
LB, UB = -2 * np.pi, 2 * np.pi
DX = (UB - LB) / (N - 1)
LEV = np.linspace(LB, UB, N)
ULEV = np.linspace(-10, 10, A)
M1, M2, LCG, GRAV = 1.0, 0.1, 0.5, 9.81


def ddq(q1, dq1, u):
    """cart-pole accelerations (closed form of H^-1 (B u - C dq - g)); arrays broadcast."""
    h00, h01, h11 = M1 + M2, M2 * LCG * np.cos(q1), M2 * LCG ** 2
    c01 = -M2 * LCG * np.sin(q1) * dq1
    g1 = M2 * GRAV * LCG * np.sin(q1)
    r0 = u - c01 * dq1
    r1 = -g1 + 0 * u
    det = h00 * h11 - h01 * h01
    return (h11 * r0 - h01 * r1) / det, (-h01 * r0 + h00 * r1) / det


def tile_cells(i0, i1, r2, r3):
    """x_next in grid-cell units for the nodes (i0, i1, r2, r3) x all actions: arrays [n2, n3, A, 4] and in-box mask."""
    i2, i3 = np.meshgrid(r2, r3, indexing="ij")
    q0, q1, v0, v1 = LEV[i0], LEV[i1], LEV[i2], LEV[i3]
    a0, a1 = ddq(q1, v1[..., None], ULEV[None, None, :])
    xn = np.stack([np.broadcast_to((q0 + v0 * DT)[..., None], a0.shape), np.broadcast_to((q1 + v1 * DT)[..., None], a0.shape),
                   v0[..., None] + a0 * DT, v1[..., None] + a1 * DT], -1)
    cell = (xn - LB) / DX
    inb = np.all((xn >= LB) & (xn <= UB), -1)
    return cell, inb


def cycles(addr, live, groups, mod):
    """LDS cycles of one access: addr [64] dword addresses, live [64] bool, per lane group max distinct addresses per bank."""
    tot = 0
    for g in groups:
        a = np.unique(addr[g][live[g]])
        if a.size == 0:
            continue
        tot += np.bincount(a % mod, minlength=mod).max()
    return tot


HALVES = [np.arange(0, 32), np.arange(32, 64)]


def model(tv0, tv1, pitch_of, lane_order="row", samples=40, seed=0, wide=False, verbose=False):
    """tile tv0 x tv1 in the (i2, i3) plane; pitch_of(row_len) -> row pitch in dwords.  Returns (cycles per ds_read2,
    ideal cycles, LDS dwords of the largest window)."""
    rng = np.random.default_rng(seed)
    tot = ideal = 0
    maxwin = 0
    for _ in range(samples):
        i0, i1 = rng.integers(0, N, 2)
        t2 = rng.integers(0, (N + tv0 - 1) // tv0) * tv0
        t3 = rng.integers(0, (N + tv1 - 1) // tv1) * tv1
        r2, r3 = np.arange(t2, min(N, t2 + tv0)), np.arange(t3, min(N, t3 + tv1))
        cell, inb = tile_cells(i0, i1, r2, r3)
        if not inb.any():
            continue
        fl = np.clip(np.floor(cell).astype(int), 0, N - 2)
        lo = [fl[..., d][inb].min() for d in range(4)]
        hi = [fl[..., d][inb].max() + 1 for d in range(4)]
        w = [hi[d] - lo[d] + 1 for d in range(4)]
        rs = pitch_of(w[3])
        plane = w[2] * rs
        maxwin = max(maxwin, w[0] * w[1] * plane)
        n2, n3 = len(r2), len(r3)
        base = ((fl[..., 0] - lo[0]) * w[1] + (fl[..., 1] - lo[1])) * plane + (fl[..., 2] - lo[2]) * rs + (fl[..., 3] - lo[3])
        base = base.reshape(n2 * n3, A)
        live = inb.reshape(n2 * n3, A)
        order = np.arange(n2 * n3)
        if lane_order == "col":       # lanes run along i2 first
            order = order.reshape(n2, n3).T.reshape(-1)
        nw = (len(order) + 63) // 64
        pad = np.full(nw * 64, -1)
        pad[:len(order)] = order
        for wv in range(nw):
            ids = pad[wv * 64:(wv + 1) * 64]
            ok = ids >= 0
            for a in range(A):
                lv = ok & live[np.maximum(ids, 0), a]
                if not lv.any():
                    continue
                b = base[np.maximum(ids, 0), a]
                for c0, c1, c2 in itertools.product((0, 1), (0, 1), (0, 1)):
                    adr = b + (c0 * w[1] + c1) * plane + c2 * rs
                    if wide:          # one ds_read_b64 (bank mod 64 of the first dword; needs 8-byte alignment)
                        tot += cycles(adr // 2 * 2, lv, HALVES, 64)
                        ideal += 2
                    else:
                        tot += cycles(adr, lv, HALVES, 32) + cycles(adr + 1, lv, HALVES, 32)
                        ideal += 4
    return tot, ideal, maxwin


if __name__ == "__main__":
    odd = lambda n: n | 1
    mul4 = lambda n: (n + 3) // 4 * 4
    mul32 = lambda n: (n + 31) // 32 * 32
    for name, tv0, tv1, pitch, order in [
        ("10x51 pitch mult-of-4 (production, dma16)", 10, 51, mul4, "row"),
        ("10x51 pitch odd", 10, 51, odd, "row"),
        ("10x51 pitch mult-of-32", 10, 51, mul32, "row"),
        ("10x51 pitch 4k lanes along i2", 10, 51, mul4, "col"),
        ("16x32 pitch mult-of-4", 16, 32, mul4, "row"),
        ("16x32 pitch mult-of-32", 16, 32, mul32, "row"),
        ("8x64 pitch mult-of-4", 8, 64, mul4, "row"),
        ("15x34 pitch mult-of-4", 15, 34, mul4, "row"),
        ("15x34 pitch mult-of-32", 15, 34, mul32, "row"),
    ]:
        t, i, mw = model(tv0, tv1, pitch, order)
        print("%-48s cycles/ds_read2 %.2f (ideal 4)  conflict share %.2f  window %d B" % (name, 4.0 * t / i, 1 - i / t, 4 * mw))
