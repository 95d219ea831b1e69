#!/usr/bin/env python3
"""Round-5 fault hunt: where do two builds differ?  usage: hunt_cmp.py <tagA> <tagB> <dims comma list> [sweeps]"""
import sys
import numpy as np
a, b = sys.argv[1], sys.argv[2]
dims = [int(x) for x in sys.argv[3].split(",")]
K = int(sys.argv[4]) if len(sys.argv) > 4 else 3
for k in range(1, K + 1):
    Ja, Jb = np.load("/tmp/hunt_J_%s_k%d.npy" % (a, k)), np.load("/tmp/hunt_J_%s_k%d.npy" % (b, k))
    pa, pb = np.load("/tmp/hunt_pi_%s_k%d.npy" % (a, k)), np.load("/tmp/hunt_pi_%s_k%d.npy" % (b, k))
    bad = np.flatnonzero(~((Ja == Jb) | (np.isnan(Ja) & np.isnan(Jb))))
    print("sweep %d: %d of %d nodes differ in J (%d NaN in %s, %d in %s), %d in pi" % (k, bad.size, Ja.size, np.isnan(Ja).sum(), a, np.isnan(Jb).sum(), b, (pa != pb).sum()))
    if bad.size:
        idx = np.array(np.unravel_index(bad, dims)).T
        for d in range(len(dims)):
            u, c = np.unique(idx[:, d], return_counts=True)
            print("  axis %d: %d distinct indices; most frequent %s" % (d, u.size, sorted(zip(c.tolist(), u.tolist()), reverse=True)[:12]))
        for j in bad[:24]:
            print("   node", j, np.unravel_index(j, dims), a, Ja[j], pa[j], b, Jb[j], pb[j])
        d = np.abs(Ja[bad].astype(np.float64) - Jb[bad])
        print("  |dJ|: min %g median %g max %g" % (np.nanmin(d), np.nanmedian(d), np.nanmax(d)))
