"""
Internal axis order of a value-iteration problem (round 5; opt-in: DynamicProgramming(..., internal_order="swapped")).

The float32 window sweep of 4-D grids (pyro_amd/csrc/sweep_lean4.inc) runs the lanes of a wave along the LAST state axis.  The
gather of every lane is shifted by the displacement of ITS node; lanes with different displacements read a window row that no
longer maps one lane to one LDS bank, and 42 % of the sweep's LDS cycles are bank conflicts.  For the cart-pole
the displacement depends on (theta, dtheta) only (reference pyro/dynamic/cartpole.py:369-437: H, C, g are functions of q[1], dq[1]).
In the reference's state order (x, theta, dx, dtheta) the last axis is dtheta -- every lane has its own shift.  With the
generalised coordinates swapped, q = (theta, x), the state is (theta, x, dtheta, dx), the last axis is dx and the lanes of a tile
row all shift together.

It is the same problem: the same grid levels per physical axis, the same dynamics (Dyn<PVI_DYN_CARTPOLE_SW>, its own dynamics id:
it reads the angle and its rate from the other slots and returns (ddtheta, ddx)), the same cost (Q, S, xbar permuted).  The value
function is the reference's with its axes transposed: `SwappedProblem` keeps the device arrays in the internal order and
transposes J and pi where they cross the class surface (get / set), so the caller sees the reference's node order.

What differs from the reference-order engine: float32 roundings (the corner sum runs over the permuted axes) -- inside the float32
tolerance, like every other float32 kernel variant; float64 handles are never swapped (their sums follow the reference's order
operation for operation).  Not offered on a swapped engine: look-up-table builds, rollouts, sharding (NotImplementedError).
"""
import numpy as np

from pyro_amd import _native

SWAP = (1, 0, 3, 2)          # q = (theta, x): state axes (x, theta, dx, dtheta) -> (theta, x, dtheta, dx); its own inverse


def swap_applies(sys, dd, dtype):
    """The swapped order is defined for the stock cart-pole's closed form in float32."""
    return dd is not None and dd[0] == _native.DYN_CARTPOLE and np.dtype(dtype) == np.float32 and getattr(sys, "n", 0) == 4


def swap_problem_kwargs(kw):
    """The keyword arguments of _native.Problem (GridDynamicSystem._problem_kwargs) for the same problem in the swapped order."""
    if kw["dynamics_id"] != _native.DYN_CARTPOLE or len(kw["x_levels"]) != 4:
        raise NotImplementedError("internal_order='swapped': the cart-pole's closed form only")
    p = list(SWAP)
    out = dict(kw)
    out["x_levels"] = [kw["x_levels"][i] for i in p]
    out["x_lb"] = np.asarray(kw["x_lb"], dtype=float)[p]
    out["x_ub"] = np.asarray(kw["x_ub"], dtype=float)[p]
    out["dynamics_id"] = _native.DYN_CARTPOLE_SW       # core.h DynCartPole<true>: the same five constants
    out["dyn_params"] = [float(v) for v in kw["dyn_params"]]
    # (trig: cos / sin over the ANGLE's levels -- the same two arrays; the library reads them along axis 0 in this order)
    cost = kw.get("cost")
    if cost is not None:
        c = dict(cost)
        for name in ("Q", "S"):
            if name in c:
                c[name] = np.asarray(c[name], dtype=float)[np.ix_(p, p)]
        c["xbar"] = np.asarray(c["xbar"], dtype=float)[p]
        out["cost"] = c
    for key in ("rows", "obstacles", "ext_J", "ext_pi"):
        if kw.get(key) is not None:
            raise NotImplementedError("internal_order='swapped' with %s" % key)
    return out


class SwappedProblem:
    """A _native.Problem of the swapped order behind the reference's node order.  Node-ordered arrays are transposed on the way
    in and out; everything else (sweeps, statistics, describe, close ...) is the inner handle's."""

    swapped = True

    def __init__(self, inner, dims):
        self._inner = inner
        self.dims = tuple(int(v) for v in dims)                      # the reference's order
        self._idims = tuple(self.dims[i] for i in SWAP)              # the inner handle's order
        if tuple(inner.dims) != self._idims:
            raise ValueError("inner handle has dims %s, expected %s" % (inner.dims, self._idims))

    # ---- node-ordered arrays ----------------------------------------------------------------
    def _out(self, a):
        return np.ascontiguousarray(np.asarray(a).reshape(self._idims).transpose(SWAP)).reshape(-1)

    def _in(self, a):
        a = np.asarray(a)
        if a.size != int(np.prod(self.dims)):
            raise ValueError("Grid size does not match data")
        return np.ascontiguousarray(a.reshape(self.dims).transpose(SWAP)).reshape(-1)

    def get_J(self, row0=None, nrows=None, prev=False):
        if row0 is not None or nrows is not None:
            raise NotImplementedError("row ranges of a swapped engine")
        return self._out(self._inner.get_J(prev=prev))

    def set_J(self, J, row0=None, nrows=None):
        if row0 is not None or nrows is not None:
            raise NotImplementedError("row ranges of a swapped engine")
        self._inner.set_J(self._in(J))

    def get_pi(self, row0=None, nrows=None):
        if row0 is not None or nrows is not None:
            raise NotImplementedError("row ranges of a swapped engine")
        return self._out(self._inner.get_pi())

    def set_pi(self, pi, row0=None, nrows=None):
        if row0 is not None or nrows is not None:
            raise NotImplementedError("row ranges of a swapped engine")
        self._inner.set_pi(self._in(pi))

    # ---- what a swapped engine does not offer -----------------------------------------------------
    def _refuse(self, what):
        raise NotImplementedError("%s on an engine with internal_order='swapped': build the problem in the reference's order" % what)

    def build_tables(self, *a, **k):
        self._refuse("look-up tables")

    def set_tables(self, *a, **k):
        self._refuse("look-up tables")

    def policy_tables(self, *a, **k):
        self._refuse("policy tables")

    def set_interpolation(self, *a, **k):
        self._refuse("another interpolant")

    def rollout(self, *a, **k):
        self._refuse("rollouts")

    def set_rollout_params(self, *a, **k):
        self._refuse("rollouts")

    def device_J(self, *a, **k):
        self._refuse("device pointers")

    def device_pi(self, *a, **k):
        self._refuse("device pointers")

    def describe(self):
        return self._inner.describe() + " order=swapped"

    def __getattr__(self, name):            # sweep, terminal_cost, synchronize, last_sweep_ms, self_check, close, dtype, ...
        return getattr(self.__dict__["_inner"], name)
