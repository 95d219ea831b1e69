"""Merge gpurun_out/counters_<w>.json (tools_counters.sh) into profiles/counters.json (read by bench.py)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles", "counters.json")
out = json.load(open(dst)) if os.path.exists(dst) else {}
for w in sys.argv[1:]:
    src = json.load(open(os.path.join(ROOT, "gpurun_out", "counters_%s.json" % w)))["kernels"]
    sweep = next(v for k, v in src.items() if "k_sweep" in k)
    cal = next((v for k, v in src.items() if "to_f64" in k), {})
    out[w] = {
        "hbm_bytes_per_launch": (2.0 * sweep["FETCH_SIZE"] + sweep["WRITE_SIZE"]) * 1024.0,
        "fetch_kib_raw": sweep["FETCH_SIZE"], "write_kib_raw": sweep["WRITE_SIZE"],
        "valu_insts_per_launch": sweep["SQ_INSTS_VALU"], "salu_insts_per_launch": sweep["SQ_INSTS_SALU"],
        "lds_insts_per_launch": sweep["SQ_INSTS_LDS"], "waves": sweep["SQ_WAVES"],
        "valu_busy_quadcycles": sweep["SQ_ACTIVE_INST_VALU"], "lds_idx_active_cycles": sweep["SQ_LDS_IDX_ACTIVE"],
        "lds_bank_conflict_cycles": sweep["SQ_LDS_BANK_CONFLICT"], "gui_active_cycles_8xcd": sweep["GRBM_GUI_ACTIVE"],
        "calibration_k_to_f64": {k: cal[k] for k in ("FETCH_SIZE", "WRITE_SIZE") if k in cal},
    }
    json.dump({"workload": w, "kernels": src}, open(os.path.join(ROOT, "profiles", "r01_counters_%s.json" % w), "w"), indent=1)
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: v.get("hbm_bytes_per_launch") for k, v in out.items() if isinstance(v, dict)}))
