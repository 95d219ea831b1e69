"""The C/OpenMP twin of the oracle must be bit-identical to the NumPy oracle (CPU only)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import c_oracle as CO
from oracle import vi_oracle as O
from test_oracle_golden import DOUBLEP, TWOLINK, load, problem_from

CASES = {
    "pendulum_21x21x5": (O.DYN_PENDULUM, O.pendulum_consts()),
    "pendulum_demo_51x51x9": (O.DYN_PENDULUM, O.pendulum_consts()),
    "cartpole_11p4x5": (O.DYN_CARTPOLE, O.cartpole_consts()),
    "twolink_11p4x3x3": (O.DYN_TWOLINK, O.twolink_consts(**TWOLINK)),
    "doublependulum_13x11x13x11x3x3": (O.DYN_TWOLINK, O.twolink_consts(**DOUBLEP)),
}


@pytest.mark.parametrize("name", list(CASES))
def test_c_twin_bitwise(name):
    g = load(name)
    p = problem_from(g, *CASES[name])
    alpha = float(g["alpha"]) if "alpha" in g.files else 1.0
    c = CO.CProblem(p)
    J = O.terminal_cost(p)
    assert np.array_equal(c.terminal_cost(), J)
    for _ in range(3):
        Jn, pi = O.sweep(p, J, alpha)
        Jc, pic = c.sweep(J, alpha, threads=2)
        assert np.array_equal(Jc, Jn) and np.array_equal(pic, pi)
        J = Jn
    # partial range + f32 storage rounding
    lo, hi = p.nodes_n // 3, p.nodes_n // 2
    Jc, _ = c.sweep(J, alpha, lo, hi, f32=True)
    Jn, _ = O.sweep(p, J, alpha, ids=np.arange(lo, hi), dtype=np.float32)
    assert np.array_equal(Jc, Jn)


def test_c_twin_config1_full_solve():
    g = load("config1_pendulum_101x101x11")
    p = problem_from(g, O.DYN_PENDULUM, O.pendulum_consts())
    c = CO.CProblem(p)
    J, k, delta = c.terminal_cost(), 0, p.INF
    while delta > 0.1:
        Jn, pi = c.sweep(J)
        _, delta = O.sweep_stats(Jn, J)
        J, k = Jn, k + 1
    assert k == 618
    np.testing.assert_allclose(J, g["J"], rtol=1e-12, atol=1e-12)
