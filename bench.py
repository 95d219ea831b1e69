#!/usr/bin/env python3
"""
bench.py -- value-iteration sweep throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c1|c2|c2p|c3|c4|c5|<custom>]

A "step" is one VI sweep (one Bellman backup of every state-action cell of the grid); J is resident in HBM before the
timed region.  W untimed sweeps, then batches of exactly K sweeps, each batch bracketed by a device synchronisation,
repeated until the timed region is at least 0.3 s long (`batches`, `timed_steps` in the line; a 1 ms region measures
the clock ramp, not the kernel).  For N > 1: barrier + max over ranks (pyro_amd/parallel_bench.py).  Rank 0 prints ONE
JSON line: metric = state-action cell updates per second (whole job), sweeps/s, set-up time, the roofline of the sweep
kernel (algorithmic bytes / HIP-event kernel time on the kernel's own stream) and -- at N = 1 -- a CPU baseline (the
oracle's C/OpenMP twin on this box's host cores, bounded sample) with the relative error of J against it.

Default N = 1 headline: C3 = BASELINE.json configs[2], cart-pole 101^4 x 21, f32 -- the largest single-GPU float32
configuration.  The default run carries the other BASELINE configurations as `secondary` objects, each with its own
roofline: C2 (configs[1]), C2' (the north-star 201x201x201 pendulum grid), C5 (configs[4], f64), C4 on one GPU (the
workload of the N > 1 lines), C1 (configs[0]) and the table tier.  `--workload X` prints the line of X alone.
"""
import argparse
import contextlib
import io
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the CPU baseline's OpenMP team: one thread per core, pinned, sleeping (not spinning) between parallel regions --
# must be in the environment before libgomp is loaded
os.environ.setdefault("OMP_PROC_BIND", "spread")
os.environ.setdefault("OMP_PLACES", "cores")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
HBM_STREAM_GBS = 6300.0      # measured on the GPU box: pure 16-byte read stream, tools/hbmbench.hip (6.0-6.7 TB/s)
VALU_PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6}      # vector peaks (MI355X_MICROARCH.md / public spec)
MIN_REGION_S = 0.3
# reference NumPy solver timed in the BUILD container (it cannot travel to the GPU box)
REFERENCE_NUMPY_SWEEPS_PER_SEC = {"c1": 163.0, "c2": 0.0518, "c2p": 0.887}


def git_head():
    try:
        return subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True,
                              timeout=5).stdout.strip() or None
    except Exception:
        return None


def oracle_problem(cfg):
    """Oracle-side description of the same workload (checker / CPU baseline only)."""
    from oracle import vi_oracle as O
    s, g, cf = cfg["sys"], cfg["grid_sys"], cfg["cf"]
    dyn_id, params = s.device_dynamics()
    return O.Problem(g.x_level, g.u_level, g.dt, dyn_id, np.array(params), cf.Q, cf.R, cf.S, cf.xbar, cf.ubar,
                     float(cf.INF), float(cf.EPS), x_lb=s.x_lb, x_ub=s.x_ub, u_lb=s.u_lb, u_ub=s.u_ub)


def usable_cpus():
    """CPUs this process may really use: scheduler affinity, cut by a cgroup CPU quota if there is one."""
    from oracle import c_oracle as CO
    n = len(CO.ALLOWED_CPUS)            # (taken before libgomp pinned this thread)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return n


def cpu_baseline(cfg, budget_s=20.0):
    """Oracle C/OpenMP twin (oracle/vi_oracle.c) on the host cores, bounded sample.  One thread per physical core;
    whole sweeps inside ONE persistent parallel region (vio_sweeps: preallocated ping-pong buffers, no thread start-up
    between sweeps) while they fit the budget, otherwise a leading slice of the nodes of one sweep.  The single-core
    rate is measured on a slice beside it, so the line shows how the all-cores number scales.
    Returns (record, J after `n` sweeps | None, n, slice | None)."""
    from oracle import c_oracle as CO
    p = oracle_problem(cfg)
    c = CO.CProblem(p)
    cores = max(1, min(CO.physical_cores(), usable_cpus(), CO.max_threads()))
    J0 = c.terminal_cost()
    A = p.actions_n
    cells_per_sweep = p.nodes_n * A
    # a slice from the middle of the grid (in- and out-of-box cells in their global proportion) for the probes
    mid = p.nodes_n // 2

    def probe(threads, seconds):
        n = 2048 * threads
        while True:
            n = min(n, p.nodes_n - mid)
            t0 = time.perf_counter()
            c.sweep(J0, 1.0, mid, mid + n, threads=threads)
            el = time.perf_counter() - t0
            if el >= seconds or mid + n >= p.nodes_n:
                return n * A / max(el, 1e-9)
            n *= 4
    one = probe(1, 0.5)
    rate = probe(cores, 0.5)
    sweep_s = cells_per_sweep / rate
    rec = dict(unit="cells/s", cores=cores, kind="port", per_core_value=one,
               host_cpus_visible=os.cpu_count(), host_cpus_usable=usable_cpus())
    if 2.5 * sweep_s <= budget_s:
        nsweeps = int(max(1, min(50, (budget_s - sweep_s) // sweep_s)))
        _, _, work = c.sweeps(J0, 1, threads=cores)                 # untimed: first touch of the buffers
        work[0][:] = J0
        t0 = time.perf_counter()
        J, _, _ = c.sweeps(None, nsweeps, threads=cores, work=work)
        dt = time.perf_counter() - t0
        v = nsweeps * cells_per_sweep / dt
        rec.update(value=v, sweeps_per_sec=nsweeps / dt, scaling_vs_one_core=v / one, efficiency=v / one / cores,
                   sample="%d whole sweeps of the workload in %.1f s, one persistent OpenMP region of %d threads "
                          "(oracle/vi_oracle.c vio_sweeps)" % (nsweeps, dt, cores))
        return rec, J, nsweeps, None
    nodes = int(min(p.nodes_n - mid, max(4096, rate * budget_s / A)))
    t0 = time.perf_counter()
    Js, _ = c.sweep(J0, 1.0, mid, mid + nodes, threads=cores)
    dt = time.perf_counter() - t0
    v = nodes * A / dt
    rec.update(value=v, sweeps_per_sec=v / cells_per_sweep, scaling_vs_one_core=v / one, efficiency=v / one / cores,
               sample="nodes [%d, %d) of %d, one sweep from J0, in %.1f s on %d threads (oracle/vi_oracle.c vio_sweep)"
                      % (mid, mid + nodes, p.nodes_n, dt, cores))
    return rec, Js, 1, (mid, mid + nodes)


N_SIMD, CLOCK_HZ = 256 * 4, 2.4e9          # 256 CUs x 4 SIMDs, 2.4 GHz
VALU_CLK, LDS_CLK = 2.33, 15.0             # cheapest issue cost per wave64 instruction, measured by tools/valubench.hip


def issue_roofline(ctr, kern_ms, cells):
    """What actually bounds the fused sweeps: SIMD instruction issue.  On gfx950 the vector-ALU, LDS and scalar streams
    of a SIMD add up instead of overlapping (tools/valubench.hip: 64 v_fma + 8 ds_read2_b32 take 153 + 120 clk,
    not max); cheapest costs per wave64 instruction: VALU 2.33 clk (v_fma; v_cmp / v_cndmask / v_cvt 4, v_pk 4.5),
    ds_read2_b32 15 clk.  `frac` = instruction counts (rocprofv3 PMC, profiles/counters.json) priced at those
    cheapest costs / SIMD cycles the kernel was resident: the share of the kernel time that is issue at the
    hardware's best rate; the rest is dearer instruction kinds, scalar work and exposed latency.  This is the
    builder's additive model, not a vendor roofline; the instruction counts are the committed PMC passes named in
    `counters_source`, the kernel time is measured in this run."""
    simd_clk = kern_ms * 1e-3 * CLOCK_HZ
    acc = (ctr["valu_insts_per_launch"] * VALU_CLK + ctr.get("lds_insts_per_launch", 0.0) * LDS_CLK) / N_SIMD
    return {"bound": "simd-issue", "achieved": acc, "peak": simd_clk, "unit": "clk per SIMD", "frac": acc / simd_clk,
            "model": "VALU %.2f clk + LDS %.0f clk per wave64 instruction, additive" % (VALU_CLK, LDS_CLK),
            "valu_insts_per_cell": ctr["valu_insts_per_launch"] * 64.0 / cells,
            "lds_insts_per_cell": ctr.get("lds_insts_per_launch", 0.0) * 64.0 / cells,
            "salu_insts_per_cell": ctr.get("salu_insts_per_launch", 0.0) * 64.0 / cells,
            "lds_bank_conflict_share": (ctr["lds_bank_conflict_cycles"] / ctr["lds_idx_active_cycles"]
                                        if ctr.get("lds_idx_active_cycles") else None),
            "counters_source": ctr.get("source")}


def lds_roofline(ctr, kern_ms, cells, n, w, lean):
    """The second on-chip ceiling of the float32 sweeps: LDS bandwidth.  Every cell gathers 2^n values of w bytes from the
    LDS window (N*A*2^n*w bytes per sweep, C3 140 GB -- not HBM traffic); the narrow-read class they use
    (ds_read2_b32, ds_read_b32) moves 128 B per clock per CU (MI355X_MICROARCH.md, LDS), 256 CUs at 2.4 GHz = 78.6 TB/s.
    `lds_busy_frac` = LDS-array cycles per CU (SQ_LDS_IDX_ACTIVE, committed PMC pass) / kernel cycles measured in this run;
    `bank_conflict_share` of those cycles are conflict cycles."""
    if not lean or not ctr.get("lds_idx_active_cycles"):
        return None
    alg = cells * (1 << n) * w
    peak = 256 * 128 * CLOCK_HZ / 1e12
    ach = alg / (kern_ms * 1e-3) / 1e12
    return {"bound": "lds", "achieved": ach, "peak": peak, "unit": "TB/s", "frac": ach / peak,
            "algorithmic_bytes_per_launch": alg,
            "lds_busy_frac": ctr["lds_idx_active_cycles"] / 256.0 / (kern_ms * 1e-3 * CLOCK_HZ),
            "bank_conflict_share": ctr["lds_bank_conflict_cycles"] / ctr["lds_idx_active_cycles"],
            "counters_source": ctr.get("source")}


def load_counters(workload):
    """Per-launch hardware counters of the sweep kernel from the committed rocprofv3 PMC passes
    (profiles/counters.json, produced by tools/tools_counters.sh + tools/make_counters_json.py).  They are NOT measured
    by this run: every use carries the `source` recorded with them (profile file, commit, kernel)."""
    path = os.path.join(ROOT, "profiles", "counters.json")
    try:
        return json.load(open(path)).get(workload, {})
    except Exception:
        return {}


def measure(name, steps, warmup, keep_handle=False):
    """Create the problem through the class surface (timed: set-up), W warm-up sweeps, then batches of K sweeps until
    the timed region reaches MIN_REGION_S.  Returns (line fragment, cfg, handle | None)."""
    from pyro_amd import configs
    from pyro_amd.planning import dynamicprogramming
    cfg = configs.build(name)
    g = cfg["grid_sys"]
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, cfg["cf"], dtype=cfg["dtype"])
    p = dp._p
    p.synchronize()
    setup_ms = (time.perf_counter() - t0) * 1e3       # pvi_create (lean set-up, tile tuning) + J0 = h(x)
    dp.save_time_history = False
    dp.verbose = False
    N, A = g.nodes_n, g.actions_n
    w = 4 if cfg["dtype"] == "float32" else 8
    dt_name = "f32" if w == 4 else "f64"
    pbytes = 1 if A <= 256 else 2

    if warmup:
        p.sweep(warmup, 1.0, -1.0)
    p.synchronize()
    batches, elapsed, kern_ms_sum, stats = 0, 0.0, 0.0, None
    while batches == 0 or elapsed < MIN_REGION_S:
        t0 = time.perf_counter()
        stats, done = p.sweep(steps, 1.0, -1.0)
        p.synchronize()
        elapsed += time.perf_counter() - t0
        assert done == steps
        kern_ms_sum += p.last_sweep_ms()               # HIP events on the kernel's own stream
        batches += 1
    timed = steps * batches
    kern_ms = kern_ms_sum / timed

    alg_bytes = N * (2 * w + pbytes)
    ctr = load_counters(cfg["name"])
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    flops_cell = {2: 30, 4: 90}[g.sys.n] if g.sys.m == 1 else 120
    cells_per_s_kernel = N * A / (kern_ms * 1e-3)
    out = {
        "value": N * A * timed / elapsed, "unit": "cells/s", "steps": steps, "warmup": warmup,
        "batches": batches, "timed_steps": timed, "timed_region_s": elapsed,
        "ms_per_step": elapsed / timed * 1e3, "sweeps_per_sec": timed / elapsed, "setup_ms": setup_ms,
        "dtype": dt_name,
        "config": {"workload": "%s: %s" % (cfg["name"], cfg["description"]), "nodes": N, "actions": A,
                   "cells_per_sweep": N * A, "dt": g.dt, "alpha": 1.0, "parallelism": "1 GPU"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": ctr.get("hbm_bytes_per_launch"),
                     "traffic_source": ctr.get("source") if ctr.get("hbm_bytes_per_launch") else None,
                     "measured_read_stream_GBps": HBM_STREAM_GBS,
                     "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kern_ms,
                     "note": "nominal: the fused sweep is instruction-issue bound (A actions per node on %d B of "
                             "compulsory traffic), see roofline_issue" % (2 * w + pbytes)},
        "roofline_issue": None if not ctr.get("valu_insts_per_launch") else issue_roofline(ctr, kern_ms, N * A),
        "roofline_lds": lds_roofline(ctr, kern_ms, N * A, g.sys.n, w, w == 4 and "k_sweep_lean" in str(ctr.get("kernel", ""))),
        "flops_frac_vector_peak": {"dtype": dt_name, "peak_tflops": VALU_PEAK_TFLOPS[dt_name],
                                   "algorithmic_flops_per_cell": flops_cell,
                                   "frac": cells_per_s_kernel * flops_cell / (VALU_PEAK_TFLOPS[dt_name] * 1e12)},
        "last_stats": [float(v) for v in stats[-1]],
        "kernel_path": p.describe(),
    }
    ref = REFERENCE_NUMPY_SWEEPS_PER_SEC.get(cfg["name"])
    if ref:
        out["reference_numpy_build_container"] = {
            "sweeps_per_sec": ref, "x_faster": out["sweeps_per_sec"] / ref,
            "note": "reference NumPy solver timed in the build container (8 vCPU, BASELINE.md); it cannot travel to "
                    "the GPU box -- context, not a same-box measurement"}
    if not keep_handle:
        p.close()
        p = None
    return out, cfg, p


def accuracy_vs_cpu(p, J_cpu, n_cmp, rows):
    """max |J_gpu - J_cpu| / max |J_cpu| after n_cmp sweeps from J0 (whole grid, or the CPU sample's node range)."""
    p.terminal_cost()
    p.sweep(n_cmp, 1.0, -1.0)
    Jg = p.get_J()
    if rows is not None:
        Jg = Jg[rows[0]:rows[1]]
    return float(np.abs(Jg - J_cpu).max() / max(np.abs(J_cpu).max(), 1e-300))


def run_single(args):
    head, cfg, p = measure(args.workload, args.steps, args.warmup, keep_handle=True)
    out = {"metric": "vi_state_action_cell_updates_per_sec", "value": head.pop("value"), "unit": head.pop("unit"),
           "n_gpus": 1, "steps": head.pop("steps"), "warmup": head.pop("warmup"),
           "ms_per_step": head.pop("ms_per_step"), "higher_is_better": True,
           # N = 1 of the weak-scaling family of the N > 1 lines (C3 per GPU, pyro_amd/parallel_bench.py)
           "scaling": "weak",
           "vs_baseline": None, "dtype": head.pop("dtype"), "data": "synthetic", "config": head.pop("config")}
    out.update(head)
    out["head"] = git_head()
    out["jstar_rel_err_vs_cpu"], out["jstar_rel_err_after_sweeps"] = None, 0
    if not args.no_cpu:
        # (after the timed region: a descheduled launch thread would show up as GPU idle time inside it)
        cpu, J_cpu, n_cmp, rows = cpu_baseline(cfg, args.cpu_budget)
        out["cpu_baseline"] = cpu
        out["speedup_vs_cpu_baseline"] = out["value"] / cpu["value"]
        out["jstar_rel_err_vs_cpu"] = accuracy_vs_cpu(p, J_cpu, n_cmp, rows)
        out["jstar_rel_err_after_sweeps"] = n_cmp
        out["jstar_rel_err_nodes"] = "whole grid" if rows is None else "nodes [%d, %d)" % rows
    p.close()
    if args.with_secondary:
        sec = {}
        for name, st, wu in (("c2", 2000, 200), ("c2p", 2000, 200), ("c5", 10, 2), ("c4", 5, 2), ("c1", 2000, 200)):
            try:
                frag, scfg, sp = measure(name, st, wu, keep_handle=(name == "c2p"))
                if sp is not None:                       # north-star grid: small enough for its own CPU leg
                    cpu, J_cpu, n_cmp, rows = cpu_baseline(scfg, 5.0)
                    frag["cpu_baseline"] = cpu
                    frag["speedup_vs_cpu_baseline"] = frag["value"] / cpu["value"]
                    frag["jstar_rel_err_vs_cpu"] = accuracy_vs_cpu(sp, J_cpu, n_cmp, rows)
                    sp.close()
                sec[name] = frag
            except Exception as e:                       # a secondary line must not take the headline down
                sec[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            from pyro_amd import configs
            with contextlib.redirect_stdout(io.StringIO()):
                c2 = configs.build("c2")
            sec["table_tier_c2"] = table_tier_reference(c2)
        except Exception as e:
            sec["table_tier_c2"] = {"error": "%s: %s" % (type(e).__name__, e)}
        out["secondary"] = sec
        out["sharded_workload_1gpu"] = sec.get("c4")      # the workload of the N > 1 lines, on one GPU
    print(json.dumps(out))


def table_tier_reference(cfg, sweeps=10):
    """The same workload through tier B (x_next / G tables of the reference's LUT class, packed once and streamed once
    per sweep by k_sweep_tablep): the one HBM-bound sweep of the library, so its HBM fraction is a roofline in the
    usual sense.  Tables are built on the GPU by the fused handle (pvi_build_tables) and handed back through
    pvi_set_tables."""
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
    ms, n = 0.0, 0
    while n == 0 or ms < MIN_REGION_S * 1e3:
        h.sweep(sweeps, 1.0, -1.0)
        ms += h.last_sweep_ms()
        n += sweeps
    ms /= n
    packed = "table-packed" in h.describe()
    h.close()
    # bytes the sweep streams per cell: the packed records (int32 offset, n fractions, G; float32 or float64)
    n_ = xn.shape[2]
    rec = (4 + 4 * n_ + 4) if cfg["dtype"] == "float32" else (8 + 8 * n_ + 8)
    byt = G.size * (rec if packed else (n_ * 8 + 8))
    w = 4 if cfg["dtype"] == "float32" else 8
    alg = G.size * (n_ * w + w + 0.125)                   # N*A*(n*w + w + 1/8)
    return {"kernel": "k_sweep_tablep (packed records)" if packed else "k_sweep_table", "ms_per_step": ms,
            "timed_steps": n, "table_bytes_per_step": byt, "cells_per_sec": G.size / (ms * 1e-3),
            "roofline": {"bound": "hbm", "achieved": byt / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": byt / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "measured_read_stream_GBps": HBM_STREAM_GBS,
                         "frac_of_measured_stream": byt / (ms * 1e-3) / 1e9 / HBM_STREAM_GBS,
                         "algorithmic_bytes_per_launch": alg, "frac_algorithmic": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default=None)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-secondary", action="store_true", help="headline workload only")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1 or os.environ.get("PVI_FORCE_PARALLEL"):
        from pyro_amd import parallel_bench
        return parallel_bench.run(args)
    args.with_secondary = args.workload is None and not args.no_cpu and not args.no_secondary   # the default driver run
    args.workload = args.workload or "c3"
    big = args.workload in ("c3", "c4", "c5")
    args.steps = args.steps if args.steps is not None else (20 if big else 2000)
    args.warmup = args.warmup if args.warmup is not None else (2 if big else 200)
    run_single(args)


if __name__ == "__main__":
    main()
