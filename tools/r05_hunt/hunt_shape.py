#!/usr/bin/env python3
"""Round-5 fault hunt: shape of the wrong-node set.  usage: hunt_shape.py <ref> <tag> <k> <dims>"""
import sys
import numpy as np
ref, tag, k = sys.argv[1], sys.argv[2], int(sys.argv[3])
dims = [int(x) for x in sys.argv[4].split(",")]
J0, p0 = np.load("/tmp/hunt_J_%s_k%d.npy" % (ref, k)), np.load("/tmp/hunt_pi_%s_k%d.npy" % (ref, k))
J, p = np.load("/tmp/hunt_J_%s_k%d.npy" % (tag, k)), np.load("/tmp/hunt_pi_%s_k%d.npy" % (tag, k))
bad = np.flatnonzero((J != J0) | (p != p0))
idx = np.array(np.unravel_index(bad, dims)).T
groups = {}
for (a, b, c, d), j in zip(idx.tolist(), bad.tolist()):
    groups.setdefault((a, b), []).append((c, d, j))
print("%d wrong nodes in %d position nodes (i0, i1)" % (bad.size, len(groups)))
for (a, b), v in sorted(groups.items())[:40]:
    c = np.array([x[0] for x in v]); d = np.array([x[1] for x in v])
    print(" (i0=%d, i1=%d): %d nodes, i2 in [%d, %d], i3 in [%d, %d]; dJ %s; pi ref %s got %s" % (
        a, b, len(v), c.min(), c.max(), d.min(), d.max(),
        np.array2string((J[[x[2] for x in v]] - J0[[x[2] for x in v]])[:6], precision=3),
        p0[[x[2] for x in v]][:6].tolist(), p[[x[2] for x in v]][:6].tolist()))
    if len(v) <= 64:
        print("    ", sorted((x[0], x[1]) for x in v))
