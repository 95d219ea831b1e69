"""The three-dimensional golden cases (helicopter tunnel, car parking, active suspension) as oracle problems.
Shared by the CPU oracle tests and the GPU parity tests; parameters as tests/golden/make_goldens.py set them."""
import os

import numpy as np

from conftest import GOLDEN
from oracle import vi_oracle as O


def _ground(a, w, phi, x, slope):
    """z(x) / dz(x) of QuarterCarOnRoughTerrain (suspension.py:72-92): sum over the sine terms, left to right."""
    z = 0
    for i in range(a.size):
        z = z + (a[i] * w[i] * np.cos(w[i] * (x - phi[i])) if slope else a[i] * np.sin(w[i] * (x - phi[i])))
    return z


def case3d(name):
    """-> (golden npz, oracle Problem, alpha)."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    lv = O.make_levels(g["x_lb"], g["x_ub"], g["dims"])
    ul = O.make_levels(g["u_lb"], g["u_ub"], g["udims"])
    kw = dict(x_lb=g["x_lb"], x_ub=g["x_ub"], u_lb=g["u_lb"], u_ub=g["u_ub"])
    if name.startswith("helicopter"):
        mass, vx, width = 0.1, 5.0, 1.0
        dyn, c, alpha = O.DYN_HELICOPTER, np.array([1.0 / mass, vx]), 0.999
        kw.update(obstacles=dict(axes=(2, 1), half=(width, width),
                                 boxes=[[2, 2, 4, 4], [8, 5, 10, 10], [14, 0, 16, 4]]), domain_check=True)
    elif name.startswith("car"):
        dyn, c, alpha = O.DYN_KINCAR, np.array([1. / float(g["lenght"])]), 0.99
        U = np.stack(np.meshgrid(*ul, indexing="ij"), axis=-1).reshape(-1, 2)
        kw.update(obstacles=dict(axes=(0, 1), half=(float(g["lenght"]) * 0.5, float(g["width"]) * 0.5),
                                 boxes=g["obstacles"].tolist()),
                  act_aux=np.array([U[a, 0] * np.tan(U[a, 1]) * (1. / float(g["lenght"])) for a in range(len(U))]))
    else:
        mass, k, b, vx = g["params"]
        dyn, c, alpha = O.DYN_QUARTERCAR, np.array([1. / mass, k, b, vx]), 0.99
        kw.update(level_tables=dict(
            z=np.array([_ground(g["a"], g["w"], g["phi"], x, False) for x in lv[2]], dtype=float),
            dz=np.array([_ground(g["a"], g["w"], g["phi"], x, True) for x in lv[2]], dtype=float)))
    p = O.Problem(lv, ul, float(g["dt"]), dyn, c, g["Q"], g["R"], g["S"], g["xbar"], g["ubar"], float(g["INF"]),
                  float(g["EPS"]), **kw)
    return g, p, alpha


CASES3D = ("helicopter_11x11x11x5", "car_21x21x11x3x3", "suspension_13x11x21x5")
