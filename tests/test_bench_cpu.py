"""CPU-side checks of bench.py: the cpu_baseline leg (oracle C twin, bounded sample) and the helpers around it."""
import contextlib
import io
import os
import sys

import numpy as np

from conftest import ROOT


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_cpu_baseline_whole_sweeps_and_slice_modes():
    bench = _bench()
    from oracle import vi_oracle as O
    from pyro_amd import configs
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build("pendulum:61,61:5:float32")
    rec, J, n, rows, _twin = bench.cpu_baseline(cfg, budget_s=3.0)
    assert rec["kind"] == "port" and rec["unit"] == "cells/s" and rec["value"] > 0 and rec["per_core_value"] > 0
    assert 1 <= rec["cores"] <= (os.cpu_count() or 1) and rows is None and n >= 1
    assert "whole sweeps" in rec["sample"] and 0 < rec["efficiency"] < 4
    p = bench.oracle_problem(cfg)
    Jr = O.terminal_cost(p)
    for _ in range(n):
        Jr, _ = O.sweep(p, Jr)
    assert np.array_equal(J, Jr)                         # the timed sweeps are the oracle's sweeps, bit for bit
    # a budget too small for whole sweeps falls back to a slice of one sweep from J0
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build("cartpole:21,21,21,21:7:float32")
    rec, Js, n, rows, _twin = bench.cpu_baseline(cfg, budget_s=0.05)
    assert n == 1 and rows is not None and "nodes [" in rec["sample"]
    p = bench.oracle_problem(cfg)
    Jr, _ = O.sweep(p, O.terminal_cost(p), ids=np.arange(rows[0], rows[1]))
    assert np.array_equal(Js, Jr)


def test_usable_cpus_and_counter_sources():
    bench = _bench()
    assert 1 <= bench.usable_cpus() <= (os.cpu_count() or 1)
    import json
    ctr = json.load(open(os.path.join(ROOT, "profiles", "counters.json")))
    for name, rec in ctr.items():
        if isinstance(rec, dict):
            assert "source" in rec and "profiles/" in rec["source"], name      # counters are labelled, never anonymous
