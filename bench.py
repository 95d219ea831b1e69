#!/usr/bin/env python3
"""
bench.py -- value-iteration sweep throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c2p|c3|c4|c5|...]

A "step" is one VI sweep (one Bellman backup of every state-action cell of the grid); J is
resident in HBM before the timed region.  W untimed sweeps, then exactly K timed sweeps bracketed
by a device synchronisation (plus a barrier and max over ranks for N > 1).  Rank 0 prints ONE JSON
line: metric = state-action cell updates per second (whole job), plus sweeps/s, the roofline of the
sweep kernel (algorithmic bytes / HIP-event kernel time) and -- at N = 1 -- a CPU baseline (the
oracle's C/OpenMP twin timed on this box's host cores on a bounded sample) and the relative error
of J against it.

Default workload: BASELINE.json configs[1] (pendulum 1001x1001x51, f32) at N = 1; cart-pole
101^4 x 21 (configs[2]), axis-0 slabs with a halo exchange over RCCL, at N > 1 (DESIGN.md
"bench workloads" explains why the 2-D grid is not sharded).
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
HBM_STREAM_GBS = 6300.0      # measured on the GPU box: pure 16-byte read stream, tools/hbmbench.hip (6.0-6.7 TB/s)
VALU_PEAK_F32_TFLOPS = 157.3


def oracle_problem(cfg):
    """Oracle-side description of the same workload (checker / CPU baseline only)."""
    from oracle import vi_oracle as O
    s, g, cf = cfg["sys"], cfg["grid_sys"], cfg["cf"]
    dyn_id, params = s.device_dynamics()
    return O.Problem(g.x_level, g.u_level, g.dt, dyn_id, np.array(params), cf.Q, cf.R, cf.S, cf.xbar, cf.ubar,
                     float(cf.INF), float(cf.EPS), x_lb=s.x_lb, x_ub=s.x_ub, u_lb=s.u_lb, u_ub=s.u_ub)


def cpu_baseline(cfg, budget_s=20.0):
    """Oracle C/OpenMP twin on the host cores, bounded sample: whole sweeps while they fit the
    budget, otherwise a leading slice of nodes of one sweep."""
    from oracle import c_oracle as CO
    p = oracle_problem(cfg)
    c = CO.CProblem(p)
    cores = CO.max_threads()
    J = c.terminal_cost()
    cells_per_sweep = p.nodes_n * p.actions_n
    # calibrate on a slice that runs for at least ~0.3 s (thread start-up would bias a shorter probe)
    n_probe = max(1, min(p.nodes_n, 4096))
    while True:
        t0 = time.perf_counter()
        c.sweep(J, 1.0, 0, n_probe)
        el = time.perf_counter() - t0
        if el >= 0.3 or n_probe >= p.nodes_n:
            break
        n_probe = min(p.nodes_n, n_probe * 4)
    rate = n_probe * p.actions_n / max(el, 1e-9)
    sweep_s = cells_per_sweep / rate
    if sweep_s <= budget_s / 2:
        nsweeps = int(max(1, min(50, budget_s // sweep_s)))
        t0 = time.perf_counter()
        for _ in range(nsweeps):
            J, _ = c.sweep(J, 1.0)
        dt = time.perf_counter() - t0
        return dict(value=nsweeps * cells_per_sweep / dt, unit="cells/s", cores=cores, kind="port",
                    sample="%d full sweeps of the workload in %.1f s (oracle/vi_oracle.c, OpenMP)" % (nsweeps, dt),
                    sweeps_per_sec=nsweeps / dt), J, nsweeps
    nodes = int(min(p.nodes_n, rate * budget_s / p.actions_n))
    t0 = time.perf_counter()
    c.sweep(J, 1.0, 0, nodes)
    dt = time.perf_counter() - t0
    v = nodes * p.actions_n / dt
    return dict(value=v, unit="cells/s", cores=cores, kind="port",
                sample="first %d of %d nodes of one sweep in %.1f s (oracle/vi_oracle.c, OpenMP)" % (nodes, p.nodes_n, dt),
                sweeps_per_sec=v / cells_per_sweep), None, 0


N_SIMD, CLOCK_HZ = 256 * 4, 2.4e9          # 256 CUs x 4 SIMDs, 2.4 GHz
VALU_CLK, LDS_CLK = 2.33, 15.0             # cheapest issue cost per wave64 instruction, measured by tools/valubench.hip


def issue_roofline(ctr, kern_ms, cells):
    """What actually bounds the sweep: SIMD instruction issue.  On gfx950 the vector-ALU, LDS and scalar streams
    of a SIMD add up instead of overlapping (tools/valubench.hip: 64 v_fma + 8 ds_read2_b32 take 153 + 120 clk,
    not max); cheapest costs per wave64 instruction: VALU 2.33 clk (v_fma; v_cmp / v_cndmask / v_cvt 4, v_pk 4.5),
    ds_read2_b32 15 clk.  `frac` = instruction counts (rocprofv3 PMC, profiles/counters.json) priced at those
    cheapest costs / SIMD cycles the kernel was resident: the share of the kernel time that is issue at the
    hardware's best rate; the rest is dearer instruction kinds, scalar work and exposed latency."""
    simd_clk = kern_ms * 1e-3 * CLOCK_HZ
    acc = (ctr["valu_insts_per_launch"] * VALU_CLK + ctr.get("lds_insts_per_launch", 0.0) * LDS_CLK) / N_SIMD
    return {"bound": "simd-issue", "achieved": acc, "peak": simd_clk, "unit": "clk per SIMD", "frac": acc / simd_clk,
            "model": "VALU %.2f clk + LDS %.0f clk per wave64 instruction, additive" % (VALU_CLK, LDS_CLK),
            "valu_insts_per_cell": ctr["valu_insts_per_launch"] * 64.0 / cells,
            "lds_insts_per_cell": ctr.get("lds_insts_per_launch", 0.0) * 64.0 / cells,
            "salu_insts_per_cell": ctr.get("salu_insts_per_launch", 0.0) * 64.0 / cells}


def load_counters(workload):
    """Per-launch hardware counters of the sweep kernel from the committed rocprofv3 PMC passes
    (profiles/counters.json, produced by tools_counters.sh): HBM bytes and VALU instructions."""
    path = os.path.join(ROOT, "profiles", "counters.json")
    try:
        return json.load(open(path)).get(workload, {})
    except Exception:
        return {}


def run_single(args):
    from pyro_amd import configs
    from pyro_amd.planning import dynamicprogramming
    cfg = configs.build(args.workload)
    g = cfg["grid_sys"]
    with contextlib.redirect_stdout(io.StringIO()):
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, cfg["cf"], dtype=cfg["dtype"])
    dp.save_time_history = False
    dp.verbose = False
    p = dp._p
    N, A = g.nodes_n, g.actions_n
    w = 4 if cfg["dtype"] == "float32" else 8
    pbytes = 1 if A <= 256 else 2

    # timed region FIRST: the CPU baseline leaves 256 OpenMP workers spinning for a while, and a descheduled launch
    # thread shows up as GPU idle time inside a region that is only tens of milliseconds long
    p.sweep(args.warmup, 1.0, -1.0)
    p.synchronize()
    t0 = time.perf_counter()
    stats, done = p.sweep(args.steps, 1.0, -1.0)
    p.synchronize()
    dt = time.perf_counter() - t0
    assert done == args.steps
    kern_ms = p.last_sweep_ms() / args.steps           # HIP events on the kernel's stream

    # accuracy + CPU baseline (J0 -> n sweeps on both sides)
    cpu, J_cpu, n_cmp = (None, None, 0)
    if not args.no_cpu:
        cpu, J_cpu, n_cmp = cpu_baseline(cfg, args.cpu_budget)
    rel_err = None
    if J_cpu is not None:
        p.terminal_cost()
        p.sweep(n_cmp, 1.0, -1.0)
        Jg = p.get_J()
        rel_err = float(np.abs(Jg - J_cpu).max() / np.abs(J_cpu).max())

    alg_bytes = N * (2 * w + pbytes)
    ctr = load_counters(cfg["name"])
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    flops_cell = {2: 30, 4: 90}[g.sys.n] if g.sys.m == 1 else 120
    out = {
        "metric": "vi_state_action_cell_updates_per_sec", "value": N * A * args.steps / dt, "unit": "cells/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32" if w == 4 else "f64", "data": "synthetic",
        "config": {"workload": "%s: %s" % (cfg["name"], cfg["description"]), "nodes": N, "actions": A,
                   "cells_per_sweep": N * A, "dt": g.dt, "alpha": 1.0, "parallelism": "1 GPU"},
        "sweeps_per_sec": args.steps / dt,
        "jstar_rel_err_vs_cpu": rel_err, "jstar_rel_err_after_sweeps": n_cmp,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": ctr.get("hbm_bytes_per_launch"),
                     "measured_read_stream_GBps": HBM_STREAM_GBS,
                     "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kern_ms,
                     "note": "nominal: the sweep is instruction-issue bound (A actions per node on 9 B of traffic), see roofline_issue"},
        "roofline_issue": None if not ctr.get("valu_insts_per_launch") else issue_roofline(ctr, kern_ms, N * A),
        "flops_frac_f32_vector_peak": (N * A / (kern_ms * 1e-3)) * flops_cell / (VALU_PEAK_F32_TFLOPS * 1e12),
        "last_stats": [float(v) for v in stats[-1]],
        "kernel_path": p.describe(),
    }
    if cpu is not None:
        out["cpu_baseline"] = cpu
        out["speedup_vs_cpu_baseline"] = out["value"] / cpu["value"]
    if getattr(args, "with_sharded_reference", False):
        out["sharded_workload_1gpu"] = sharded_reference()
        out["table_tier"] = table_tier_reference(cfg)
    print(json.dumps(out))


def table_tier_reference(cfg, sweeps=10):
    """The same workload through tier B (x_next / G tables of the reference's LUT class, streamed once per sweep by
    k_sweep_table): the one HBM-bound sweep of the library, so its HBM fraction is a roofline in the usual sense.
    Tables are built on the GPU by the fused handle (pvi_build_tables) and handed back through pvi_set_tables."""
    from pyro_amd import _native
    g, s = cfg["grid_sys"], cfg["grid_sys"].sys
    with contextlib.redirect_stdout(io.StringIO()):
        p = g._device_problem(cost=cfg["cf"].device_cost(), dtype="float64")
    xn, _, _, G = p.build_tables(x_next_isok=False, action_isok=False)
    p.terminal_cost()
    J0 = p.get_J()
    p.close()
    h = _native.Problem(g.x_level, g.u_level, s.x_lb, s.x_ub, s.u_lb, s.u_ub, g.dt, dtype=cfg["dtype"],
                        dynamics_id=_native.DYN_TABLE, table_inf=float(cfg["cf"].INF))
    h.set_tables(xn, G, None)
    h.set_J(J0)
    h.sweep(2, 1.0, -1.0)
    h.sweep(sweeps, 1.0, -1.0)
    ms = h.last_sweep_ms() / sweeps
    packed = "table-packed" in h.describe()
    h.close()
    # bytes the sweep streams per cell: float32 handles read the packed records (int32 offset, n float32 fractions,
    # float32 G), float64 handles the reference's float64 tables
    n_ = xn.shape[2]
    rec = (4 + 4 * n_ + 4) if cfg["dtype"] == "float32" else (8 + 8 * n_ + 8)
    byt = G.size * (rec if packed else (n_ * 8 + 8))
    return {"kernel": "k_sweep_tablep (packed records)" if packed else "k_sweep_table", "ms_per_step": ms, "table_bytes_per_step": byt, "cells_per_sec": G.size / (ms * 1e-3),
            "roofline": {"bound": "hbm", "achieved": byt / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": byt / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "measured_read_stream_GBps": HBM_STREAM_GBS,
                         "frac_of_measured_stream": byt / (ms * 1e-3) / 1e9 / HBM_STREAM_GBS}}


def sharded_reference(name="c4", sweeps=5):
    """The N > 1 bench lines run a different workload than the N = 1 line (BASELINE configs[3]: cart-pole 151^4 x 31,
    axis-0 slabs; configs[1] is 44 us of work per sweep and cannot amortise a halo exchange).  Its 1-GPU rate is the
    denominator of the strong-scaling speed-up, so it is measured here as well (and again by rank 0 of every N > 1 run:
    `value_1gpu_same_workload`)."""
    from pyro_amd import configs
    from pyro_amd.planning import dynamicprogramming
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build(name)
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"], dtype=cfg["dtype"])
    p, g = dp._p, cfg["grid_sys"]
    p.sweep(2, 1.0, -1.0)
    p.synchronize()
    t0 = time.perf_counter()
    p.sweep(sweeps, 1.0, -1.0)
    p.synchronize()
    dt = time.perf_counter() - t0
    out = {"workload": "%s: %s" % (cfg["name"], cfg["description"]), "value": g.nodes_n * g.actions_n * sweeps / dt,
           "unit": "cells/s", "ms_per_step": dt / sweeps * 1e3, "steps": sweeps}
    p.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default=None)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1 or os.environ.get("PVI_FORCE_PARALLEL"):
        from pyro_amd import parallel_bench
        return parallel_bench.run(args)
    args.with_sharded_reference = args.workload is None and not args.no_cpu   # the default driver run
    args.workload = args.workload or "c2"
    big = args.workload in ("c3", "c4", "c5")
    args.steps = args.steps if args.steps is not None else (20 if big else 2000)
    args.warmup = args.warmup if args.warmup is not None else (2 if big else 200)
    run_single(args)


if __name__ == "__main__":
    main()
