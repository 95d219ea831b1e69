#!/usr/bin/env python3
"""Drop-in check in the BUILD CONTAINER only (it reads /root/reference at run time; nothing of it is copied or shipped): the
reference's own dynamic-programming demo scripts, examples/demos_by_tool/dynamicprogramming/*.py, executed UNMODIFIED with
`import pyro...` resolved to this package -- `pyro.X` -> `pyro_amd.X` by an import hook -- matplotlib on the Agg backend.
A script passes when it runs to its last line.  Under emulation today (PYROVI_LIB=tests/emu/_build/libpyrovi_emu.so), on a GPU box
it would need the reference tree, which does not travel: this is a builder's tool, its log goes to profiles/.

    PYROVI_LIB=tests/emu/_build/libpyrovi_emu.so python tools/run_reference_demos.py [script.py ... | more]
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXAMPLES = os.path.join(os.environ.get("PYRO_REFERENCE", "/root/reference"), "examples")
DEMOS = os.path.join(EXAMPLES, "demos_by_tool", "dynamicprogramming")
# the value-iteration scripts elsewhere in the reference's examples that import nothing outside the mirrored modules (`more`);
# lqr_vs_valueiteration_for_a_simple_pendulum.py and optimal_control_demo.py also need pyro.control.lqr /
# pyro.planning.trajectoryoptimisation, the rl_* scripts stable_baselines3: other tools of the reference, out of scope
MORE = ["courses/udes_gro860/dp_mass_min_time_policy_evaluation.py", "courses/udes_gro860/dp_mass_min_time_optimal.py",
        "courses/udes_gro860/dp_demo_swingup.py", "demos_by_system/pendulum_simple/simple_pendulum_with_valueiteration_quadratic.py",
        "demos_by_system/pendulum_simple/simple_pendulum_with_valueiteration_minimum_time.py",
        "demos_by_system/holonomic_mobile_robot/holonomic_mobile_robot_with_valueiteration.py",
        "demos_by_system/mountain_car/mountain_car_with_valueiteration_quadratic.py",
        "demos_by_system/car_steering/car_with_valueiteration_quadratic_cost.py",
        "demos_by_system/car_steering/car_with_valueiteration_minimum_time.py",
        "demos_by_system/car_propulsion/longitudinal_car_braking_value_iteration.py"]

BOOT = r"""
import importlib, importlib.abc, importlib.util, runpy, sys, time
sys.path.insert(0, %(root)r)
import matplotlib
matplotlib.use("Agg")
import matplotlib.pyplot as plt
plt.show = lambda *a, **k: None


class _Alias(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    # `pyro` and `pyro.anything` ARE `pyro_amd` and `pyro_amd.anything`
    def find_spec(self, name, path=None, target=None):
        if name == "pyro" or name.startswith("pyro."):
            return importlib.util.spec_from_loader(name, self)
        return None

    def create_module(self, spec):
        return importlib.import_module("pyro_amd" + spec.name[4:])

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _Alias())
t0 = time.time()
ns = runpy.run_path(%(script)r, run_name="__main__")
dp = ns.get("dp")
if dp is not None and hasattr(dp, "_p"):
    print("DEMO-ENGINE tier=%%s k=%%s %%s" %% (getattr(dp, "tier", "?"), getattr(dp, "k", "?"), dp._p.describe()[:140]))
import os
if dp is not None and os.environ.get("DEMO_SAVE"):       # (the policy file the `..._load.py` script of the same demo reads)
    dp.save_latest(os.environ["DEMO_SAVE"])
print("DEMO-OK %%.1f s" %% (time.time() - t0))
"""
# double_pendulum_optimal_swingup_load.py reads a policy that neither the reference ships nor any of its scripts writes: the solve
# of double_pendulum_optimal_swingup.py is saved under that name (dp.save_latest) so that the pair runs
SAVE_AS = {"double_pendulum_optimal_swingup.py": "double_pendulum_51_41_51_41_5_5"}


def main():
    names = sys.argv[1:] or sorted(f for f in os.listdir(DEMOS) if f.endswith(".py"))
    if names == ["more"]:
        names = MORE
    limit = int(os.environ.get("DEMO_TIMEOUT", "1500"))
    ok = skipped = 0
    for n in names:
        script = os.path.join(EXAMPLES, n) if "/" in n else os.path.join(DEMOS, n)
        t0 = time.time()
        try:
            r = subprocess.run([sys.executable, "-c", BOOT % dict(root=ROOT, script=script)], capture_output=True, text=True, timeout=limit,
                               cwd="/tmp", env=dict(os.environ, MPLBACKEND="Agg", PYTHONDONTWRITEBYTECODE="1", DEMO_SAVE=SAVE_AS.get(os.path.basename(n), "")))
            out, rc = r.stdout + r.stderr, r.returncode
        except subprocess.TimeoutExpired as e:
            out, rc = ((e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or "")) + "\nTIMEOUT", -9
        good = rc == 0 and "DEMO-OK" in out
        if not good and "FileNotFoundError" in out and ".npy" in out:
            # (double_pendulum_optimal_swingup_load.py reads a policy file that neither the reference ships nor any of its scripts writes)
            print("%-52s n/a   %6.1f s  needs a data file the reference does not ship" % (n, time.time() - t0), flush=True)
            skipped += 1
            continue
        ok += good
        eng = [l for l in out.splitlines() if l.startswith("DEMO-ENGINE")]
        last = [l for l in out.strip().splitlines() if l.strip()][-3:]
        print("%-52s %s  %6.1f s  %s" % (os.path.basename(n), "ok  " if good else "FAIL", time.time() - t0, eng[0][12:] if eng else ""), flush=True)
        if not good:
            print("      " + "\n      ".join(l[:220] for l in last), flush=True)
    print("reference demo scripts that run unmodified: %d / %d (%d more need a data file the reference does not ship)" % (ok, len(names) - skipped, skipped))
    return 0 if ok == len(names) - skipped else 1


if __name__ == "__main__":
    sys.exit(main())
