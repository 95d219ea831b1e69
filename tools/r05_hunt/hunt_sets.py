#!/usr/bin/env python3
"""Round-5 fault hunt: node sets that differ from the reference build after sweep k.  usage: hunt_sets.py <ref> <k> <tag> <tag> ..."""
import sys
import numpy as np
ref, k, tags = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
J0, p0 = np.load("/tmp/hunt_J_%s_k%d.npy" % (ref, k)), np.load("/tmp/hunt_pi_%s_k%d.npy" % (ref, k))
sets = {}
for t in tags:
    J, p = np.load("/tmp/hunt_J_%s_k%d.npy" % (t, k)), np.load("/tmp/hunt_pi_%s_k%d.npy" % (t, k))
    sets[t] = set(np.flatnonzero((J != J0) | (p != p0)).tolist())
    print("%s: %d nodes differ from %s after sweep %d" % (t, len(sets[t]), ref, k))
for a in tags:
    for b in tags:
        if a < b:
            print("  %s & %s: %d common, %d only %s, %d only %s" % (a, b, len(sets[a] & sets[b]), len(sets[a] - sets[b]), a, len(sets[b] - sets[a]), b))
