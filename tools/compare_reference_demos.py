#!/usr/bin/env python3
"""End-to-end parity on the reference's OWN workloads (build container only; reads /root/reference at run time): every script of
examples/demos_by_tool/dynamicprogramming/ (and the value-iteration scripts elsewhere, tools/run_reference_demos.py MORE) is run
TWICE, unmodified -- once with the reference itself (`import pyro` from /root/reference, NumPy / SciPy on the host) and once with
`pyro.*` resolved to `pyro_amd.*` (emulated library today) -- and what the solve left in `dp` is compared: the number of sweeps
(`dp.k`: same stop sweep), J (relative to max |J|) and pi (fraction of nodes with another action).

    PYROVI_LIB=tests/emu/_build/libpyrovi_emu.so python tools/compare_reference_demos.py [script ...]
"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import run_reference_demos as R  # noqa: E402

TAIL = r"""
import numpy as _np
_dp = ns.get("dp") or ns.get("evaluator")
if _dp is None:        # (scripts that call their solver something else: the last object that looks like one)
    _c = [v for v in ns.values() if all(hasattr(v, a) for a in ("J", "pi", "k", "grid_sys", "compute_steps"))]
    _dp = _c[-1] if _c else None
if _dp is not None:
    _np.savez(%(out)r, J=_np.asarray(_dp.J, dtype=float), pi=_np.asarray(_dp.pi).astype(int), k=int(_dp.k))
print("DEMO-OK")
"""
REF_BOOT = r"""
import runpy, sys, time
sys.path.insert(0, %(ref)r)
import matplotlib
matplotlib.use("Agg")
import matplotlib.pyplot as plt
plt.show = lambda *a, **k: None
plt.pause = lambda *a, **k: None
ns = runpy.run_path(%(script)r, run_name="__main__")
"""


def run(code, limit):
    t0 = time.time()
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=limit, cwd="/tmp",
                           env=dict(os.environ, MPLBACKEND="Agg", PYTHONDONTWRITEBYTECODE="1"))
        return r.returncode == 0 and "DEMO-OK" in r.stdout, (r.stdout + r.stderr)[-400:], time.time() - t0
    except subprocess.TimeoutExpired:
        return False, "TIMEOUT", time.time() - t0


def main():
    # (the two double-pendulum scripts: 4.4 M nodes x 25 actions of Python-loop table building in the reference -- hours; skipped)
    default = [f for f in sorted(os.listdir(R.DEMOS)) if f.endswith(".py") and "double_pendulum" not in f] + R.MORE
    names = sys.argv[1:] or default
    limit = int(os.environ.get("DEMO_TIMEOUT", "2400"))
    bad = 0
    for n in names:
        script = os.path.join(R.EXAMPLES, n) if "/" in n else os.path.join(R.DEMOS, n)
        with tempfile.TemporaryDirectory() as tmp:
            o_ref, o_new = os.path.join(tmp, "ref.npz"), os.path.join(tmp, "new.npz")
            ok_r, log_r, t_r = run(REF_BOOT % dict(ref=os.path.dirname(R.EXAMPLES), script=script) + TAIL % dict(out=o_ref), limit)
            boot = R.BOOT % dict(root=R.ROOT, script=script)
            boot = boot[:boot.index("dp = ns.get(\"dp\")")]
            ok_n, log_n, t_n = run(boot + TAIL % dict(out=o_new), limit)
            name = os.path.basename(n)
            if not (ok_r and ok_n and os.path.exists(o_ref) and os.path.exists(o_new)):
                bad += 1
                print("%-52s NOT COMPARED  reference: %s (%.0f s)  pyro_amd: %s (%.0f s)\n      %s" % (
                    name, "ok" if ok_r else "failed", t_r, "ok" if ok_n else "failed", t_n, (log_n if ok_r else log_r).strip().replace("\n", "\n      ")[-300:]), flush=True)
                continue
            a, b = np.load(o_ref), np.load(o_new)
            m = max(np.abs(a["J"]).max(), 1e-300)
            eJ = np.abs(a["J"] - b["J"]).max() / m
            dpi = float((a["pi"] != b["pi"]).mean())
            same_k = int(a["k"]) == int(b["k"])
            verdict = "ok" if (same_k and eJ <= 1e-9 and dpi <= 1e-3) else "DIFFERS"
            bad += verdict != "ok"
            print("%-52s %-7s sweeps %5d / %5d   max |dJ| / max |J| %.2e   pi differs on %.4f %% of %d nodes   (reference %.0f s, pyro_amd %.0f s)" % (
                name, verdict, int(a["k"]), int(b["k"]), eJ, 100.0 * dpi, a["pi"].size, t_r, t_n), flush=True)
    print("scripts whose solve differs from the reference's (or could not be compared): %d / %d" % (bad, len(names)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
