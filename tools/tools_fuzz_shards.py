"""Randomised run of the in-library multi-GPU path (pvi_shard_*: axis-0 slabs, halo exchange, all-gathered statistics, the
boundary-first overlap schedule) with 2 ... 6 rank PROCESSES -- under emulation today (PYROVI_LIB=tests/emu/_build/libpyrovi_emu.so,
PVI_RCCL_LIB=tests/emu/_build/librccl_emu.so: the shared-memory stand-in for librccl), on an 8-GPU node once one is reachable (drop the
two variables; one rank per GPU needs `device=rank`, which the rank script of tests/test_gpu_parity.py does not do: single box = emulation).

Random problem (pendulum 2-D / cart-pole 4-D / two-link 4-D, float64 or float32, random dims with few rows per rank, random
action counts), random world size, overlap on / off, error-feedback storage on / off for the 4-D float32 cases, a stop tolerance taken from the whole-grid solve so that the stop falls
inside the batch: the concatenated slabs must equal the whole-grid handle BIT FOR BIT (J, pi), the statistics to 1e-12 and the stop
sweep exactly.  A slab thinner than its halo is refused by pvi_shard_create on every rank alike ("refused": counted, not a failure).

usage: tools_fuzz_shards.py [n_cases] [seed]"""
import contextlib
import io
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from pyro_amd import configs

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)

import test_gpu_parity as T  # noqa: E402  (the rank script the GPU test runs)

# the GPU test's rank script with one more argument: error-feedback storage of a float32 J on the pieces of every slab
RANK = T._RCCL_RANK.replace("overlap=bool(int(sys.argv[6])))", "overlap=bool(int(sys.argv[6])), f32_feedback=bool(int(sys.argv[8])))")
assert RANK != T._RCCL_RANK

fails = refused = 0
for case in range(n_cases):
    world = int(rng.integers(2, 7))
    kind = str(rng.choice(["pendulum", "cartpole", "cartpole", "twolink"]))
    dtype = str(rng.choice(["float32", "float64"]))
    rows = world * int(rng.integers(2, 7)) + int(rng.integers(0, world))
    if kind == "pendulum":
        rows = world * int(rng.integers(2, 40)) + int(rng.integers(0, world))      # (more rows: halos of several rows)
        dims, ud = [max(rows, 8), int(rng.integers(8, 90))], [int(rng.integers(2, 14))]
    elif kind == "cartpole":
        dims, ud = [max(rows, 5)] + [int(rng.integers(5, 12)) for _ in range(3)], [int(rng.integers(2, 8))]
    else:
        dims, ud = [max(rows, 5)] + [int(rng.integers(5, 10)) for _ in range(3)], [int(rng.integers(2, 4)), int(rng.integers(2, 4))]
    overlap = bool(rng.random() < 0.6)
    fb = bool(kind != "pendulum" and dtype == "float32" and rng.random() < 0.5)      # (4-D float32: k_sweep_lean4fb, verified on hardware)
    spec = "%s:%s:%s:%s" % (kind, ",".join(map(str, dims)), ",".join(map(str, ud)), dtype)
    tag = "%3d world %d overlap %d fb %d %-46s " % (case, world, overlap, fb, spec)
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = configs.build(spec)
        from pyro_amd import _native
        try:
            h = cfg["grid_sys"]._device_problem(cost=cfg["cf"].device_cost(), dtype=dtype, flags=_native.FLAG_F32_FEEDBACK if fb else 0)
        except _native.NativeError as e:        # (a grid too small for the 4-D window sweep does not take the flag)
            if "PVI_FLAG_F32_FEEDBACK" not in str(e):
                raise
            fb = False
            h = cfg["grid_sys"]._device_problem(cost=cfg["cf"].device_cost(), dtype=dtype)
    h.terminal_cost()
    h.sweep(5, 1.0, -1.0)
    probe, _ = h.sweep(int(rng.integers(3, 40)), 1.0, -1.0)       # a scout run picks the tolerance ...
    tol = float(np.array(probe)[-1, 3]) * 1.0000001
    h.terminal_cost()                                              # ... and the run proper starts over (J and the feedback residuals)
    st5, _ = h.sweep(5, 1.0, -1.0)
    st, n = h.sweep(400, 1.0, tol)
    env = dict(os.environ, PVI_EMU_THREADS="2")
    with tempfile.TemporaryDirectory() as tmp:
        script = os.path.join(tmp, "rccl_rank.py")
        open(script, "w").write(RANK % dict(root=ROOT))
        idfile = os.path.join(tmp, "comm.id")
        procs = [subprocess.Popen([sys.executable, script, str(r), str(world), spec, idfile, os.path.join(tmp, "r%d.npz" % r), str(int(overlap)), repr(tol), str(int(fb))],
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
        outs = []
        for p in procs:
            try:
                outs.append((p.wait(timeout=900), p.stdout.read()))
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                outs.append((-9, "timeout"))
        ok = all(rc == 0 and ("RCCL-RANK-OK %d" % r) in o for r, (rc, o) in enumerate(outs))
        if not ok:
            text = "\n".join(o[-600:] for _, o in outs)
            if all(rc != 0 for rc, _ in outs) and ("halo" in text and ("rows" in text or "slab" in text)):
                refused += 1
                why = [l for l in text.splitlines() if "halo" in l]
                print(tag + "refused on every rank: " + (why[-1][:140] if why else ""), flush=True)
            else:
                fails += 1
                print(tag + "FAIL (ranks did not finish)\n" + text[-1500:], flush=True)
            h.close()
            continue
        parts = [np.load(os.path.join(tmp, "r%d.npz" % r)) for r in range(world)]
    bad = []
    if not all(int(p["n"]) == n for p in parts):
        bad.append("stop sweep %s against %d" % ([int(p["n"]) for p in parts], n))
    if not np.array_equal(np.concatenate([p["J"] for p in parts]), h.get_J()):
        bad.append("J differs")
    if not np.array_equal(np.concatenate([p["pi"] for p in parts]), h.get_pi()):
        bad.append("pi differs")
    if not all(np.allclose(p["st5"], st5[-1], rtol=1e-12) and np.allclose(p["st"], st[-1], rtol=1e-12) for p in parts):
        bad.append("statistics differ")
    desc = str(parts[0]["desc"])
    tok = dict(t.split("=", 1) for t in desc.split() if "=" in t)
    print(tag + "stop after %3d sweeps, halo %s pieces %s%s %s" % (n, tok.get("halo", "?"), tok.get("pieces", "?"), " feedback" if fb else "", "FAIL: " + "; ".join(bad) + " :: " + desc[:200] if bad else ""),
          flush=True)
    fails += bool(bad)
    h.close()
print("shards: failures %d / %d (%d refused: slab thinner than its halo)" % (fails, n_cases, refused))
sys.exit(1 if fails else 0)
