#!/usr/bin/env python3
"""Timing experiment, NOT a product build: pyro_amd/libpyrovi_ablate.so = the library with -DPVI_ABLATE_UNIFORM (sweep_lean4.inc
k_lean4_node: every node gets the same displacement, so a wave's gathers are one shifted copy of its lanes and the LDS reads of
the 4-D float32 sweep have no bank conflicts; results wrong on purpose).  The sweep time of this build is the bound of every
layout that removes the conflicts:

    python tools/build_ablate.py && PYROVI_LIB=$PWD/pyro_amd/libpyrovi_ablate.so python tools/tools_time.py c3 200
"""
import sys
sys.path.insert(0, "/root/repo")
from pyro_amd import _build
print("built", _build.build_variant("ablate", ["PVI_ABLATE_UNIFORM"], verbose=False), flush=True)
