#!/usr/bin/env python3
"""
bench.py -- value-iteration sweep throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c1|c2|c2p|c3|c4|c5|c5d|h3|<custom>]

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


def cpu_baseline(cfg, budget_s=20.0, J_start=None, depth=0):
    """Oracle C/OpenMP twin (oracle/vi_oracle.c) on the host cores, bounded sample.  One thread per physical core;
    whole sweeps inside ONE persistent parallel region (vio_sweeps: preallocated ping-pong buffers, no thread start-up
    between sweeps) while they fit the budget, otherwise a slice of the nodes of one sweep from the middle of the grid.
    The single-core rate is measured on a slice beside it, so the line shows how the all-cores number scales.
    `J_start`: the cost-to-go the sweeps start from -- the GPU's own J after `depth` sweeps, so that the same pass is the
    accuracy check at depth (a sweep from J0 = h(x) checks nothing where S = 0: J0 is identically zero there).
    Returns (record, J after `n` sweeps | None, n, slice | None, the twin)."""
    from oracle import c_oracle as CO
    p = oracle_problem(cfg)
    c = CO.CProblem(p)
    cores = max(1, min(CO.physical_cores(), usable_cpus(), CO.max_threads()))
    J0 = c.terminal_cost() if J_start is None else np.ascontiguousarray(J_start, dtype=np.float64)
    start = "J0 = h(x)" if J_start is None else "the GPU's J after %d sweeps" % depth
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
                          "from %s (oracle/vi_oracle.c vio_sweeps)" % (nsweeps, dt, cores, start))
        return rec, J, nsweeps, None, c
    nodes = int(min(p.nodes_n - mid, max(4096, rate * budget_s / A)))
    t0 = time.perf_counter()
    Js, _ = c.sweep(J0, 1.0, mid, mid + nodes, threads=cores)
    dt = time.perf_counter() - t0
    v = nodes * A / dt
    rec.update(value=v, sweeps_per_sec=v / cells_per_sweep, scaling_vs_one_core=v / one, efficiency=v / one / cores,
               sample="nodes [%d, %d) of %d, one sweep from %s, in %.1f s on %d threads (oracle/vi_oracle.c vio_sweep)"
                      % (mid, mid + nodes, p.nodes_n, start, dt, cores))
    return rec, Js, 1, (mid, mid + nodes), c


N_SIMD, CLOCK_HZ = 256 * 4, 2.4e9          # 256 CUs x 4 SIMDs, 2.4 GHz
VALU_CLK = 2.0                             # architectural floor per wave64 vector instruction (157.3 TF f32 = 32 FMA lanes/clk/SIMD)


def issue_roofline(ctr, kern_ms, cells):
    """What actually bounds the fused sweeps: SIMD instruction issue.  A SIMD issues one vector instruction of a wave64 in
    2 clk at best (32 lanes per clock: the 157.3 TF f32 vector peak); tools/valubench.hip measures 2.3 clk for v_fma_f32,
    ~4 for v_cmp / v_cndmask / v_cvt and float64, 4.5 for v_pk_fma_f32.  `frac` = vector instructions of one launch
    (rocprofv3 PMC, profiles/counters.json) priced at that 2 clk floor / SIMD cycles the kernel was resident: the share of
    the kernel that is vector issue at the hardware's best rate (always < 1); the rest is dearer instruction kinds, LDS and
    scalar instructions taking issue slots (tools/valubench.hip: the streams of a SIMD add up rather than overlap) and
    exposed latency.  The LDS side is its own object (`roofline_lds`: array-busy share from SQ_LDS_IDX_ACTIVE).  The
    instruction counts are the committed PMC passes named in `counters_source`, the kernel time is measured in this run."""
    simd_clk = kern_ms * 1e-3 * CLOCK_HZ
    acc = ctr["valu_insts_per_launch"] * VALU_CLK / N_SIMD
    return {"bound": "simd-issue", "achieved": acc, "peak": simd_clk, "unit": "clk per SIMD", "frac": acc / simd_clk,
            "model": "vector instructions x %.0f clk per wave64 instruction (the f32 vector peak); LDS: see roofline_lds" % VALU_CLK,
            "valu_insts_per_cell": ctr["valu_insts_per_launch"] * 64.0 / cells,
            "lds_insts_per_cell": ctr.get("lds_insts_per_launch", 0.0) * 64.0 / cells,
            "salu_insts_per_cell": ctr.get("salu_insts_per_launch", 0.0) * 64.0 / cells,
            "lds_bank_conflict_share": (ctr["lds_bank_conflict_cycles"] / ctr["lds_idx_active_cycles"]
                                        if ctr.get("lds_idx_active_cycles") else None),
            "counters_source": ctr.get("source")}


def lds_roofline(ctr, kern_ms, cells, n, w, lean, pairs=False):
    """The second on-chip ceiling of the float32 sweeps: LDS bandwidth.  Every cell gathers 2^n values of w bytes from the
    LDS window (N*A*2^n*w bytes per sweep, C3 140 GB -- not HBM traffic).  Peak by read class
    (MI355X_MICROARCH.md, LDS): ds_read_b32 / ds_read2_b32 (the 2-D kernel) move 128 B per clock per CU = 78.6 TB/s over
    256 CUs at 2.4 GHz; ds_read_b64 (the 4-D kernel's position pairs, `pairs`) 256 B per clock = 157 TB/s.
    `lds_busy_frac` = LDS-array cycles per CU (SQ_LDS_IDX_ACTIVE, committed PMC pass) / kernel cycles measured in this run;
    `bank_conflict_share` of those cycles are conflict cycles."""
    if not lean or not ctr.get("lds_idx_active_cycles"):
        return None
    alg = cells * (1 << n) * w
    peak = 256 * (256 if pairs else 128) * CLOCK_HZ / 1e12
    ach = alg / (kern_ms * 1e-3) / 1e12
    return {"bound": "lds", "achieved": ach, "peak": peak, "unit": "TB/s", "frac": ach / peak,
            "read_class": "ds_read_b64" if pairs else "ds_read_b32 / ds_read2_b32",
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


def norm_kernel(name):
    """A kernel-trace name ("void k_sweep64<3, unsigned char, true, true, true>(DevP, ...)") and pvi_describe's `kernel=`
    token in one form: no return type, no argument list, no spaces."""
    n = str(name or "").strip()
    if n.startswith("void "):
        n = n[5:]
    depth = 0
    for i, ch in enumerate(n):                     # cut the argument list: the first '(' outside the template brackets
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            n = n[:i]
            break
    return n.replace(" ", "")


def kernel_manifest():
    """{kernel: {"exact": ISA hash, ...}} of the library this process loads: pyro_amd/kernel_manifest.json, written by
    pyro_amd/_build.py next to libpyrovi.so from the device assembly of the same compilation (it travels to the GPU box with the
    .so).  {} when it is missing or names another library file (its "_library" sha256)."""
    from pyro_amd import _build, kernel_manifest as KM
    global _MANIFEST_CACHE
    try:
        key = (_build.MANIFEST, _build.OUT, os.path.getmtime(_build.MANIFEST), os.path.getmtime(_build.OUT))
        if _MANIFEST_CACHE is None or _MANIFEST_CACHE[0] != key:
            _MANIFEST_CACHE = (key, KM.load_manifest(_build.MANIFEST, library=_build.OUT))      # ({} unless written for THIS .so)
        return _MANIFEST_CACHE[1]
    except (OSError, ValueError):
        return {}


_MANIFEST_CACHE = None


def kernel_isa_hash(kernel):
    """ISA hash of one kernel of this build ("k_sweep_lean4<2,unsignedchar,true,true>", any spelling norm_kernel accepts), or
    None.  This is what ties PMC counters and the verified-kernel list to a code object: a comment edit under csrc/ leaves it
    alone, a changed register assignment of the same source does not (rounds 4-5 hashed the source bytes)."""
    return kernel_manifest().get(norm_kernel(kernel), {}).get("exact")


def provenance(kernel=None):
    """What the bench line says about code-object provenance (pyro_amd/kernel_manifest.py, profiles/verified_kernels.json):
    how many kernels of this build are not code objects that have passed the GPU suite on an MI355X -- opt-in ones (they run
    only when a caller asks: profiles/optin_kernels.txt) counted apart -- and whether THIS run's kernel is verified."""
    from pyro_amd import kernel_manifest as KM
    man = kernel_manifest()
    if not man:
        return {"unverified_kernels": None, "note": "pyro_amd/kernel_manifest.json missing or written for another libpyrovi.so"}
    cl = KM.classify(man)
    pats = KM.optin_patterns()
    bad = [k for k in cl["layout_only"] + cl["unverified"] if not KM.is_optin(k, pats)]
    out = {"kernels": len(man), "unverified_kernels": len(bad), "unverified_opt_in": len(cl["layout_only"]) + len(cl["unverified"]) - len(bad)}
    if bad:
        out["unverified"] = bad[:8]
    if kernel:
        k = norm_kernel(kernel)
        out["kernel"] = k
        out["kernel_isa"] = man.get(k, {}).get("exact")
        out["kernel_verified"] = k in cl["verified"]
    return out


F32_FEEDBACK = False          # --f32-feedback: the float32 accuracy mode of 4-D grids (PVI_FLAG_F32_FEEDBACK, k_sweep_lean4fb)


def check_counters(ctr, desc):
    """The committed PMC passes were taken with ONE kernel (template instantiation: recorded with them as `kernel`, the
    name the kernel trace printed) and one variant of its launch (tile shape, window layout ...: `kernel_path`);
    pvi_create picks both for THIS run by timing.  The counters describe this run only if the KERNEL NAME pvi_describe
    reports (`kernel=`) is the one they were taken from, the variant tokens agree AND the kernel's ISA hash in this build
    (kernel_isa_hash) is the one recorded with the counters; otherwise they are dropped and the line says why, loudly."""
    if not ctr:
        return {}, None
    keys = ("path", "tile", "choice", "win", "tables", "stage", "vmask", "mapping", "sparse", "npt", "lsplit")
    a = dict(t.split("=", 1) for t in str(ctr.get("kernel_path", "")).split() if "=" in t)
    b = dict(t.split("=", 1) for t in desc.split() if "=" in t)
    diff = []
    want, have = norm_kernel(ctr.get("kernel")), norm_kernel(b.get("kernel"))
    if not want or want != have:
        diff.append("kernel: counters %s, this run %s" % (want or "(not recorded)", have or "(not reported)"))
    diff += ["%s: counters %s, this run %s" % (k, a.get(k), b.get(k)) for k in keys if k in a and a.get(k) != b.get(k)]
    if not a:
        diff.append("the committed counters do not record the kernel variant they were taken with")
    isa_now = kernel_isa_hash(have) if have else None
    if not ctr.get("isa_hash") or ctr.get("isa_hash") != isa_now:
        diff.append("code object: counters taken on ISA %s of %s, this build's is %s (the kernel was recompiled to other instructions "
                    "since the PMC passes: refresh profiles/ with tools/tools_counters.sh + tools/make_counters_json.py)"
                    % (ctr.get("isa_hash") or "(not recorded)", want or "?", isa_now or "(no manifest)"))
    if diff:
        msg = "PMC counters of %s do not describe this run's kernel (%s): traffic / issue / LDS objects dropped" % (
            ctr.get("source"), "; ".join(diff))
        print("bench.py: " + msg, file=sys.stderr)
        return {}, msg
    return ctr, None


def measure(name, steps, warmup, keep_handle=False):
    """Create the problem through the class surface (timed: set-up), W warm-up sweeps, then batches of K sweeps until
    the timed region reaches MIN_REGION_S.  Returns (line fragment, cfg, handle | None)."""
    from pyro_amd import configs
    from pyro_amd.planning import dynamicprogramming
    cfg = configs.build(name)
    g = cfg["grid_sys"]
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, cfg["cf"], dtype=cfg["dtype"], f32_feedback=F32_FEEDBACK)
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

    alg_bytes = N * (2 * w + pbytes + (8 if F32_FEEDBACK else 0))      # error-feedback storage: one residual per node, read and written
    desc = p.describe()
    tok = dict(t.split("=", 1) for t in desc.split() if "=" in t)
    ctr, ctr_err = check_counters(load_counters(cfg["name"]), desc)
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    # pendulum 30, cart-pole 90, two-link 120; the helicopter's dynamics are three multiplies, its trilinear
    # interpolation 14 lerp flops + cost and minimum
    flops_cell = ({2: 30, 3: 40, 4: 90}[g.sys.n] if g.sys.m == 1 else 120)
    # Cells whose x_next leaves the grid box cost exactly INF (Q = INF + alpha*0): the sweeps that walk set-up's validity
    # masks (sparse=1) never evaluate them.  `value` counts every state-action cell of the grid (BASELINE's metric: each
    # one IS updated); flop rates must only count the cells that were computed.
    inbox = float(tok["inbox"]) if tok.get("inbox") not in (None, "-1.0000") else None
    walked = inbox if (inbox is not None and tok.get("sparse") == "1") else 1.0
    cells_per_s_kernel = N * A * walked / (kern_ms * 1e-3)
    out = {
        "value": N * A * timed / elapsed, "unit": "cells/s", "steps": steps, "warmup": warmup,
        "batches": batches, "timed_steps": timed, "timed_region_s": elapsed,
        "ms_per_step": elapsed / timed * 1e3, "sweeps_per_sec": timed / elapsed, "setup_ms": setup_ms,
        "dtype": dt_name,
        "config": {"workload": "%s: %s" % (cfg["name"], cfg["description"]), "nodes": N, "actions": A,
                   "cells_per_sweep": N * A, "dt": g.dt, "alpha": 1.0, "parallelism": "1 GPU",
                   **({"f32_feedback": True} if F32_FEEDBACK else {})},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": ctr.get("hbm_bytes_per_launch"),
                     "traffic_source": ctr.get("source") if ctr.get("hbm_bytes_per_launch") else None,
                     "measured_read_stream_GBps": HBM_STREAM_GBS,
                     "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kern_ms,
                     "kernel": norm_kernel(tok.get("kernel")),
                     "note": "nominal: the fused sweep is instruction-issue bound (A actions per node on %d B of "
                             "compulsory traffic), see roofline_issue" % (2 * w + pbytes)},
        "inbox_fraction": inbox, "cells_evaluated_fraction": walked,
        "cells_evaluated_per_sec": N * A * walked * timed / elapsed,
        "counters_error": ctr_err,
        "provenance": provenance(tok.get("kernel")),
        "roofline_issue": None if not ctr.get("valu_insts_per_launch") else issue_roofline(ctr, kern_ms, N * A),
        "roofline_lds": lds_roofline(ctr, kern_ms, N * A, g.sys.n, w, w == 4 and "k_sweep_lean" in str(ctr.get("kernel", "")),
                                     pairs=tok.get("win") == "1"),
        "flops_frac_vector_peak": {"dtype": dt_name, "peak_tflops": VALU_PEAK_TFLOPS[dt_name],
                                   "algorithmic_flops_per_cell": flops_cell, "cells": "evaluated cells only",
                                   "frac": cells_per_s_kernel * flops_cell / (VALU_PEAK_TFLOPS[dt_name] * 1e12)},
        "last_stats": [float(v) for v in stats[-1]],
        "kernel_path": desc,
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


def cpu_leg(p, cfg, budget_s, depth):
    """CPU baseline + accuracy at depth in one pass.  J_K = the GPU's cost-to-go after `depth` sweeps (downloaded; its
    float32 values are exact in float64).  CPU: n sweeps (whole grid) or one sweep (a slice) of the float64 oracle twin
    from J_K.  GPU: the same n sweeps from the same J_K.  Reported: max |J_gpu - J_cpu| / max |J_cpu| over the compared
    nodes, and the float64 Q-regret of the GPU's actions on the GPU's own previous cost-to-go (how much worse than the
    best action, relative to max |J|: the measure of a policy that is insensitive to ties)."""
    J_K = p.get_J()
    cpu, J_cpu, n_cmp, rows, c = cpu_baseline(cfg, budget_s, J_start=J_K, depth=depth)
    p.set_J(J_K)
    p.sweep(n_cmp, 1.0, -1.0)
    Jg, pig, Jprev = p.get_J(), p.get_pi(), p.get_J(prev=True)
    lo, hi = (0, Jg.size) if rows is None else rows
    scale = max(float(np.abs(J_cpu).max()), 1e-300)
    err = float(np.abs(Jg[lo:hi] - J_cpu).max() / scale)
    rng = np.random.default_rng(0)
    nodes = np.sort(rng.choice(np.arange(lo, hi), size=min(hi - lo, 200000), replace=False))
    q, qmin = c.q_at(Jprev, nodes, pig[nodes])
    ok = np.isfinite(q) & np.isfinite(qmin)
    regret = float((q[ok] - qmin[ok]).max() / scale) if ok.any() else 0.0
    acc = {"step_rel_err_vs_cpu": err, "step_rel_err_after_sweeps": n_cmp, "step_rel_err_from_depth": depth,
           "step_rel_err_nodes": "whole grid" if rows is None else "nodes [%d, %d)" % rows,
           "pi_q_regret_vs_cpu": regret, "pi_q_regret_nodes": int(nodes.size),
           "step_check": "GPU and float64 CPU twin both advance the GPU's own J after %d sweeps by %d sweep(s): a "
                         "consistency check of the kernel at depth, NOT accumulated drift (see jstar_rel_err_vs_cpu)" % (depth, n_cmp)}
    return cpu, acc


def converged_check(cfg, tol=0.1, every=100, max_sweeps=20000, feedback=True):
    """north_star: "J* match within 1e-5".  The workload solved to `tol` on the GPU -- float32 production path, float64
    path (oracle-pinned one step at a time by the tests) and, on 4-D grids, the float32 path with error-feedback storage
    (PVI_FLAG_F32_FEEDBACK: the accuracy mode, k_sweep_lean4fb) -- in lockstep batches of `every` sweeps; the drift
    max |J32 - J64| / max |J64| is recorded after every batch.  Returns the final figures, the sweep counts and the curves."""
    from pyro_amd.planning import dynamicprogramming
    g = cfg["grid_sys"]
    hs = {}
    t0 = time.perf_counter()
    kinds = [("float32", "float32", False), ("float64", "float64", False)]
    if feedback and g.sys.n == 4:
        kinds.append(("float32fb", "float32", True))
    for key, dt, fb in kinds:
        with contextlib.redirect_stdout(io.StringIO()):
            dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(g, cfg["cf"], dtype=dt, f32_feedback=fb)
        hs[key] = dp._p
    done = {k: 0 for k in hs}
    stop = {k: False for k in hs}
    secs = {k: 0.0 for k in hs}
    curve, curve_fb = [], []
    while not all(stop.values()) and max(done.values()) < max_sweeps:
        for key, h in hs.items():
            if not stop[key]:
                h.synchronize()
                t1 = time.perf_counter()
                st, n = h.sweep(every, 1.0, tol)
                h.synchronize()
                secs[key] += time.perf_counter() - t1
                done[key] += n
                stop[key] = n < every or (n and st[-1][3] <= tol)
        a, b = hs["float32"].get_J(), hs["float64"].get_J()
        m = np.abs(b).max()
        curve.append([max(done["float32"], done["float64"]), float(np.abs(a - b).max() / m)])
        if "float32fb" in hs:
            curve_fb.append([max(done["float32fb"], done["float64"]), float(np.abs(hs["float32fb"].get_J() - b).max() / m)])
    out = {"tol": tol, "sweeps_f32": done["float32"], "sweeps_f64": done["float64"], "rel_err": curve[-1][1],
           "max_transient_rel_err": max(e for _, e in curve),
           "max_J": float(np.abs(b).max()), "drift_curve": curve, "seconds": time.perf_counter() - t0,
           "paths": {k: h.describe() for k, h in hs.items()}}
    if curve_fb:
        out["feedback"] = {"sweeps": done["float32fb"], "rel_err": curve_fb[-1][1], "max_transient_rel_err": max(e for _, e in curve_fb),
                           "ms_per_sweep": secs["float32fb"] / max(1, done["float32fb"]) * 1e3,
                           "ms_per_sweep_plain": secs["float32"] / max(1, done["float32"]) * 1e3,
                           "ms_per_sweep_f64": secs["float64"] / max(1, done["float64"]) * 1e3, "drift_curve": curve_fb}
    for h in hs.values():
        h.close()
    return out


def accumulated_check(name, sweeps=None, tol=None):
    """north_star: "J* match to the CPU reference within 1e-5 relative" -- ACCUMULATED from J0 = h(x): the GPU's production
    path (the workload's own dtype) and the float64 CPU twin of the oracle both start from the terminal cost and run the
    same number of sweeps (`sweeps`, or as many as the GPU needs to reach `tol` -- a whole solve); reported: max |J_gpu -
    J_cpu| / max |J_cpu| over the whole grid and the float64 Q-regret of the GPU's policy.  Runs on the workloads whose CPU
    side fits a few seconds (the headline grids need 11 s of 16 cores per sweep: their J* is checked float32 against
    float64 on the GPU, `converged`, the float64 path being oracle-pinned step by step by the tests)."""
    from oracle import c_oracle as CO
    from pyro_amd import configs
    from pyro_amd.planning import dynamicprogramming
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build(name)
        dp = dynamicprogramming.DynamicProgrammingWithLookUpTable(cfg["grid_sys"], cfg["cf"], dtype=cfg["dtype"])
    p = dp._p
    t0 = time.perf_counter()
    if tol is not None:
        n = 0
        while True:
            st, done = p.sweep(1024, 1.0, float(tol))
            n += done
            if done < 1024 or n >= 20000:
                break
    else:
        n = int(sweeps)
        p.sweep(n, 1.0, -1.0)
    Jg, pig, Jprev = p.get_J(), p.get_pi(), p.get_J(prev=True)
    gpu_s = time.perf_counter() - t0
    p.close()
    c = CO.CProblem(oracle_problem(cfg))
    cores = max(1, min(CO.physical_cores(), usable_cpus(), CO.max_threads()))
    t0 = time.perf_counter()
    Jc = np.array(c.sweeps(c.terminal_cost(), n, threads=cores)[0])
    cpu_s = time.perf_counter() - t0
    scale = max(float(np.abs(Jc).max()), 1e-300)
    rng = np.random.default_rng(0)
    nodes = np.sort(rng.choice(Jg.size, size=min(Jg.size, 100000), replace=False))
    q, qmin = c.q_at(Jprev, nodes, pig[nodes])
    ok = np.isfinite(q) & np.isfinite(qmin)
    return {"workload": "%s: %s" % (cfg["name"], cfg["description"]), "dtype": cfg["dtype"], "sweeps_from_J0": n,
            "solved_to_tol": tol, "rel_err": float(np.abs(Jg - Jc).max() / scale), "max_J": scale,
            "pi_q_regret": float((q[ok] - qmin[ok]).max() / scale) if ok.any() else 0.0,
            "gpu_s": gpu_s, "cpu_s": cpu_s, "cpu_threads": cores}


TOL_JSTAR = 1e-5        # BASELINE.json north_star


from pyro_amd.benchline import compact_line, emit  # noqa: E402,F401  (the one driver-parsed line; shared with the N > 1 harness)


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
    depth = out["warmup"] + out["timed_steps"]
    if not args.no_cpu:
        # (after the timed region: a descheduled launch thread would show up as GPU idle time inside it)
        try:
            cpu, acc = cpu_leg(p, cfg, args.cpu_budget, depth)
            out["cpu_baseline"] = cpu
            out["speedup_vs_cpu_baseline"] = out["value"] / cpu["value"]
            out.update(acc)
        except KeyError as e:               # the oracle's C twin covers the mechanical closed forms only
            out["cpu_baseline"] = {"error": "no C twin for this system (%s): see the NumPy oracle tests" % e}
    p.close()
    failures = []
    if args.converged:
        try:
            cv = converged_check(cfg)
            out["jstar_rel_err_converged_f32_vs_f64"] = cv["rel_err"]
            out["converged"] = cv
            if not cv["rel_err"] <= TOL_JSTAR:
                failures.append("converged float32 J* differs from float64 by %.3g (> %g)" % (cv["rel_err"], TOL_JSTAR))
        except Exception as e:                           # reported LOUDLY: the line carries the error and jstar_ok = false
            out["jstar_rel_err_converged_f32_vs_f64"] = None
            out["converged"] = {"error": "%s: %s" % (type(e).__name__, e)}
            failures.append("converged check failed: %s" % out["converged"]["error"])
    if args.with_secondary:
        # accumulated J against the CPU twin from J0: a whole solve of configs[0], and fixed depths of the
        # north-star grid and of a small cart-pole (the headline's kernel family) in float32
        acc = {}
        for name, kw in (("c1", dict(tol=0.1)), ("c2p", dict(sweeps=200)), ("c3s", dict(sweeps=30))):
            try:
                acc[name] = accumulated_check(name, **kw)
            except Exception as e:
                acc[name] = {"error": "%s: %s" % (type(e).__name__, e)}
                failures.append("accumulated check %s failed: %s" % (name, acc[name]["error"]))
        good = {k: v for k, v in acc.items() if "rel_err" in v}
        out["jstar_checks"] = acc
        if good:
            worst = max(good, key=lambda k: good[k]["rel_err"])
            out["jstar_rel_err_vs_cpu"] = good[worst]["rel_err"]
            out["jstar_rel_err_vs_cpu_on"] = "; ".join(
                "%s %s %d sweeps from J0%s: %.2e" % (k, v["dtype"], v["sweeps_from_J0"],
                                                      " (solved to tol %g)" % v["solved_to_tol"] if v["solved_to_tol"] else "",
                                                      v["rel_err"]) for k, v in good.items())
            if not out["jstar_rel_err_vs_cpu"] <= TOL_JSTAR:
                failures.append("accumulated J differs from the CPU twin by %.3g on %s (> %g)" % (out["jstar_rel_err_vs_cpu"], worst, TOL_JSTAR))
        sec = {}
        for name, st, wu in (("c2", 2000, 200), ("c2p", 2000, 200), ("c5", 10, 2), ("c5d", 5, 2), ("c4", 5, 2),
                             ("c1", 2000, 200), ("h3", 200, 20)):
            try:
                frag, scfg, sp = measure(name, st, wu, keep_handle=(name == "c2p"))
                if sp is not None:                       # north-star grid: small enough for its own CPU leg
                    cpu, acc1 = cpu_leg(sp, scfg, 5.0, frag["warmup"] + frag["timed_steps"])
                    frag["cpu_baseline"] = cpu
                    frag["speedup_vs_cpu_baseline"] = frag["value"] / cpu["value"]
                    frag.update(acc1)
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
        out["sharded_workload_1gpu"] = "secondary.c4"     # the workload of the N > 1 lines, on one GPU
    if args.converged or args.with_secondary:
        out["jstar_ok"] = not failures
    if failures:
        out["jstar_failures"] = failures
        print("bench.py: ACCURACY CHECK FAILED: " + "; ".join(failures), file=sys.stderr)
    from pyro_amd import _native, _build
    if os.path.realpath(_native.LIB_PATH) != os.path.realpath(_build.OUT):
        # PYROVI_LIB points somewhere else (a sanitizer or experiment build, a test build): the figures of this line are not
        # the product's
        out["invalid"] = "PYROVI_LIB=%s: not the product library pyro_amd/libpyrovi.so" % _native.LIB_PATH
    emit(out)


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
    ap.add_argument("--no-selftest", action="store_true", help="N > 1: skip the sharded-against-one-rank check of a small grid")
    ap.add_argument("--selftest-grid", default=None, help="N > 1: another grid for that check (default cartpole:41,13,17,15:7:float32)")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--converged", action="store_true", default=None,
                    help="solve the workload to tol 0.1 in float32 and float64 and report the J* difference "
                         "(default: on for the default run, off with --workload)")
    ap.add_argument("--no-converged", dest="converged", action="store_false")
    ap.add_argument("--f32-feedback", action="store_true",
                    help="with --workload <4-D float32 workload>: error-feedback storage of J (k_sweep_lean4fb)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1 or os.environ.get("PVI_FORCE_PARALLEL"):
        from pyro_amd import parallel_bench
        return parallel_bench.run(args)
    args.with_secondary = args.workload is None and not args.no_cpu and not args.no_secondary   # the default driver run
    if args.converged is None:
        args.converged = args.with_secondary
    if args.f32_feedback:
        if args.workload is None:
            ap.error("--f32-feedback goes with --workload (the default line is the plain float32 production path)")
        global F32_FEEDBACK
        F32_FEEDBACK = True
    args.workload = args.workload or "c3"
    big = args.workload in ("c3", "c4", "c5", "c5d")
    args.steps = args.steps if args.steps is not None else (20 if big else 2000)
    args.warmup = args.warmup if args.warmup is not None else (2 if big else 200)
    run_single(args)


if __name__ == "__main__":
    main()
