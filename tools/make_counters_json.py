"""Merge gpurun_out/counters_<w>.json (tools_counters.sh) into profiles/counters.json (read by bench.py)."""
import json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROUND = os.environ.get("PVI_ROUND", "r04")
HEAD = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or os.environ.get("PVI_HEAD", "?")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles", "counters.json")
out = json.load(open(dst)) if os.path.exists(dst) else {}
for w in sys.argv[1:]:
    raw = json.load(open(os.path.join(ROOT, "gpurun_out", "counters_%s.json" % w)))
    src = raw["kernels"]
    # The production sweep kernel: the one pvi_describe names (`kernel=` in the kernel path the counter pass logged: the
    # template instantiation of the handle's last launch).  pvi_create also launches slower candidates of the same family
    # once or twice each (mappings, dense / sparse walks, tile shapes): "the k_sweep* with the most vector work" picked one
    # of THOSE for C5 / C5d in round 3.  Without a `kernel=` token: the sweep kernel with the most calls.
    from bench import norm_kernel, kernel_isa_hash
    want = norm_kernel(dict(t.split("=", 1) for t in raw.get("kernel_path", "").split() if "=" in t).get("kernel"))
    sweeps = {k: v for k, v in src.items() if "k_sweep" in k and "finish" not in k and "_probe" not in k}
    named = [(k, v) for k, v in sweeps.items() if want and want != "-" and norm_kernel(k) == want]
    if want and want != "-" and not named:
        # (the counter script truncates kernel names at 64 characters)
        named = [(k, v) for k, v in sweeps.items() if want.startswith(norm_kernel(k)) or norm_kernel(k).startswith(want[:40])]
    if len(named) == 1:
        kname, sweep = named[0]
    else:
        if want and want != "-":
            print("make_counters_json: %s: %d kernels match %s -- taking the one with the most calls" % (w, len(named), want))
        kname, sweep = max((named or list(sweeps.items())), key=lambda kv: kv[1].get("calls", 0))
    cal = next((v for k, v in src.items() if "to_f64" in k), {})
    out[w] = {
        "hbm_bytes_per_launch": (2.0 * sweep["FETCH_SIZE"] + sweep["WRITE_SIZE"]) * 1024.0,
        "fetch_kib_raw": sweep["FETCH_SIZE"], "write_kib_raw": sweep["WRITE_SIZE"],
        "valu_insts_per_launch": sweep["SQ_INSTS_VALU"], "salu_insts_per_launch": sweep["SQ_INSTS_SALU"],
        "lds_insts_per_launch": sweep["SQ_INSTS_LDS"], "waves": sweep["SQ_WAVES"],
        "valu_busy_quadcycles": sweep["SQ_ACTIVE_INST_VALU"], "lds_idx_active_cycles": sweep["SQ_LDS_IDX_ACTIVE"],
        "lds_bank_conflict_cycles": sweep["SQ_LDS_BANK_CONFLICT"], "gui_active_cycles_8xcd": sweep["GRBM_GUI_ACTIVE"],
        "calibration_k_to_f64": {k: cal[k] for k in ("FETCH_SIZE", "WRITE_SIZE") if k in cal},
        "wave_cycles_quad": sweep.get("SQ_WAVE_CYCLES"), "wait_any_quad": sweep.get("SQ_WAIT_ANY"),
        "wait_inst_any_quad": sweep.get("SQ_WAIT_INST_ANY"), "active_inst_any_quad": sweep.get("SQ_ACTIVE_INST_ANY"),
        "kernel": kname, "kernel_calls": sweep.get("calls"), "kernel_path": raw.get("kernel_path", ""),
        # the code object the passes ran on (the counter pass ran on THIS tree's library: gpurun snapshots it with its manifest)
        "isa_hash": (raw.get("isa_hashes") or {}).get(norm_kernel(kname)) or kernel_isa_hash(kname),
        "source": "profiles/%s_counters_%s.json (rocprofv3 --pmc passes of tools/tools_counters.sh), kernel %s as of commit %s"
                  % (ROUND, w, kname.split("<")[0], HEAD),
    }
    json.dump({"workload": w, "kernel_path": raw.get("kernel_path", ""), "kernels": src}, open(os.path.join(ROOT, "profiles", "%s_counters_%s.json" % (ROUND, w)), "w"), indent=1)
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: v.get("hbm_bytes_per_launch") for k, v in out.items() if isinstance(v, dict)}))
