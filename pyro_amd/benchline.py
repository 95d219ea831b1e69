"""The ONE line of stdout the driver parses, for bench.py (N = 1) and pyro_amd/parallel_bench.py (N > 1).

Round 3's line had grown to 26.5 KB (eight secondary workloads with three roofline objects each, twenty timed candidate
tilings inside every `kernel_path`, a twenty-point drift curve) and the driver could not parse it.  The full record now goes
to gpurun_out/bench_full.json and to stderr; stdout carries a compact headline of < 4 KB.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _r(v, digits=4):
    """Compact float for the headline line."""
    if isinstance(v, float):
        return float("%.*g" % (digits + 2, v))
    return v


def compact_line(full, full_path=None):
    """The ONE line the driver parses (< 4 KB): BASELINE's metric and contract keys, the headline workload's roofline and
    CPU baseline, the accuracy figures, and one short object per secondary workload.  Everything else -- the candidate
    timings inside `kernel_path`, the drift curve, every secondary's own roofline objects -- is in the full record
    (stderr and `full_path`)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data")
    out = {k: _r(full.get(k)) for k in keep}
    cfg = full.get("config", {})
    out["config"] = {k: cfg.get(k) for k in ("workload", "nodes", "actions", "cells_per_sweep", "parallelism") if k in cfg}
    for k in ("sweeps_per_sec", "batches", "timed_steps", "timed_region_s", "setup_ms", "rccl_ranks", "kernel_ms_max_rank",
              "exposed_exchange_ms_max_rank", "overlap_efficiency", "value_1gpu_c3", "value_1gpu_same_workload",
              "strong_scaling_speedup", "in_library_rccl_error", "selftest", "invalid", "value_unverified"):
        if k in full:
            out[k] = _r(full[k])
    pr = full.get("per_rank")
    if pr:                                  # N > 1: per-rank kernel / exchange times (a few floats per rank)
        out["per_rank"] = {k: v for k, v in pr.items() if k != "source"}
    rf = full.get("roofline") or {}
    out["roofline"] = {k: _r(rf.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms",
                                                  "algorithmic_bytes_per_launch", "kernel")}
    if rf.get("traffic_source"):
        out["roofline"]["traffic_source"] = str(rf["traffic_source"]).split(" ")[0]        # the profile file
    ri = full.get("roofline_issue")
    if ri:
        out["roofline_issue"] = {k: _r(ri.get(k)) for k in ("frac", "valu_insts_per_cell", "lds_insts_per_cell",
                                                             "salu_insts_per_cell", "lds_bank_conflict_share")}
    rl = full.get("roofline_lds")
    if rl:
        out["roofline_lds"] = {k: _r(rl.get(k)) for k in ("frac", "lds_busy_frac", "bank_conflict_share")}
    fv = full.get("flops_frac_vector_peak")
    if fv:                                  # algorithmic flops of the evaluated cells / the dtype's vector peak: what the loop reaches
        out["vector_peak_frac"] = _r(fv.get("frac"))
    if full.get("counters_error"):
        out["counters_error"] = str(full["counters_error"])[:200]
    pv = full.get("provenance")
    if pv:       # code objects of this build that have not passed the GPU suite on hardware (pyro_amd/kernel_manifest.py)
        out["unverified_kernels"] = pv.get("unverified_kernels")
        out["kernel_verified"] = pv.get("kernel_verified")
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: _r(cb.get(k)) for k in ("value", "unit", "cores", "kind", "sweeps_per_sec", "per_core_value",
                                                          "error") if k in cb}
        if "sample" in cb:
            out["cpu_baseline"]["sample"] = str(cb["sample"])[:160]
    for k in ("speedup_vs_cpu_baseline", "jstar_rel_err_vs_cpu", "jstar_rel_err_vs_cpu_on", "jstar_ok",
              "jstar_rel_err_converged_f32_vs_f64", "step_rel_err_vs_cpu", "pi_q_regret_vs_cpu"):
        if k in full:
            out[k] = _r(full[k])
    cv = full.get("converged")
    if cv:
        out["converged"] = {k: _r(cv.get(k)) for k in ("tol", "sweeps_f32", "sweeps_f64", "rel_err", "max_transient_rel_err",
                                                       "seconds", "error") if k in cv}
        fb = cv.get("feedback")
        if fb:      # the float32 accuracy mode (PVI_FLAG_F32_FEEDBACK) solved beside the two
            out["converged"]["feedback"] = {k: _r(fb.get(k)) for k in ("sweeps", "rel_err", "max_transient_rel_err", "ms_per_sweep",
                                                                       "ms_per_sweep_plain", "ms_per_sweep_f64") if k in fb}
    tok = dict(t.split("=", 1) for t in str(full.get("kernel_path", "")).split() if "=" in t)
    out["kernel_path"] = " ".join("%s=%s" % (k, tok[k]) for k in ("path", "tile", "block", "lds_bytes", "mapping", "sparse")
                                  if k in tok)
    sec = {}
    for name, f in (full.get("secondary") or {}).items():
        if "error" in f:
            sec[name] = {"error": str(f["error"])[:120]}
            continue
        r = f.get("roofline") or {}
        sec[name] = {"ms_per_step": _r(f.get("ms_per_step")), "kernel_ms": _r(r.get("kernel_ms")),
                     "cells_per_s": _r(f.get("value", f.get("cells_per_sec"))), "hbm_frac": _r(r.get("frac")),
                     "traffic_x": _r(r["traffic"] / r["algorithmic_bytes_per_launch"]) if r.get("traffic") else None,
                     "dtype": f.get("dtype")}
        for k in ("scaling", "strong_scaling_speedup", "value_1gpu_same_workload", "kernel_ms_max_rank",
                  "exposed_exchange_ms_max_rank", "overlap_efficiency"):
            if f.get(k) is not None:
                sec[name][k] = _r(f[k])
        if f.get("launches_per_step") is not None:
            sec[name]["launches_per_step"] = _r(f["launches_per_step"])
        if f.get("reference_numpy_build_container"):
            sec[name]["x_reference_numpy"] = _r(f["reference_numpy_build_container"]["x_faster"])
        if f.get("counters_error"):
            sec[name]["counters_error"] = True
    if sec:
        out["secondary"] = sec
    out["head"] = full.get("head")
    out["full_record"] = full_path
    return out


def emit(full, fd=None):
    """Full record -> gpurun_out/bench_full.json (+ stderr); compact headline -> the LAST (and only) line of stdout."""
    path = None
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "bench_full.json")
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        path = os.path.relpath(path, ROOT)
    except OSError:
        path = None
    print("bench.py full record: " + json.dumps(full), file=sys.stderr)
    sys.stderr.flush()
    line = json.dumps(compact_line(full, path))
    if fd is None:
        print(line)
        sys.stdout.flush()
    else:                                 # (N > 1: the real stdout, saved before gloo's and RCCL's banners were redirected)
        os.write(fd, (line + "\n").encode())
    return line
