"""The golden cases of the explicit (non-mechanical) systems -- helicopter tunnel, car parking, active suspension (n = 3),
point robot with obstacles, longitudinal car (n = 2) -- as oracle problems.
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
    elif name.startswith("obstacles"):
        dyn, c, alpha = O.DYN_HOLONOMIC, np.zeros(1), 1.0
        kw.update(obstacles=dict(axes=(0, 1), half=(0.0, 0.0), boxes=[[2, 2, 4, 10], [6, -8, 8, 8], [-8, -8, -1, 8]]))
    elif name.startswith("longcar"):
        lenght, xc, yc, m, grav, rho, cdA, mu_max, mu_slope = (float(v) for v in g["params"])
        ry, rr = yc / lenght, xc / lenght
        rf = 1 - rr
        dyn, c, alpha = O.DYN_LONGCAR, np.array([m, ry, m * grav * rr, m * grav * rf]), 1.0
        rcda = rho * cdA
        aux = []
        for a in range(len(ul[0])):
            mu = mu_max * (2 / (1 + np.exp(-mu_slope * ul[0][a])) - 1)        # slip2force, vehicle_propulsion.py:100
            aux.append((mu * m * grav * rr, m * (1 + mu * ry)))
        kw.update(act_aux=np.array(aux), level_tables=dict(
            fd=np.array([0.5 * rcda * v * np.abs(v) for v in lv[1]], dtype=float)))
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


CASES3D = ("helicopter_11x11x11x5", "car_21x21x11x3x3", "suspension_13x11x21x5", "obstacles_21x21x3x3", "longcar_31x31x7")
