// pyrovi.hip -- gfx950 (MI355X / CDNA4) kernels and C ABI for grid value-iteration sweeps.
//
// One fused kernel per Bellman backup (reference: pyro/planning/dynamicprogramming.py:557-570
// driven by the tables of pyro/planning/discretizer.py:342-376):
//     x_next = f(x,u)*dt + x  ->  box validity  ->  G = g(x,u)*dt | INF
//            ->  n-linear interpolation of J_k at x_next (fill 0 outside the grid)
//            ->  J_{k+1}[s] = min_a (G + alpha*J_interp),  pi[s] = first argmin
// plus the per-sweep reductions of finalize_backward_step (:240-261).
//
// Arithmetic contract (DESIGN.md "numerics"): dynamics, Euler step, validity and the
// interval/fraction of the interpolation are always float64 in the reference's operation order
// (this file is compiled with -ffp-contract=off; every fused multiply-add below is explicit).
// Only the storage of J and the interpolation/Bellman arithmetic follow `dtype`.
//
// Written for gfx950 only: 64-wide wavefronts, wave-level __shfl reductions, LDS staging.

#include "core.h"
#include "host.h"

// The fold of sweep_finish as its own one-wave launch (SweepCtl::split_finish).  With ~10^5..10^6 tiles and two workgroups
// per CU, the ticket protocol keeps every workgroup's slot occupied for two dependent atomic round trips after its last
// useful instruction (C3: 0.44 ms of 4.9 ms per sweep); a kernel boundary orders the statistics for free.
__global__ void k_sweep_finish(SweepCtl sc) {
    const int l = threadIdx.x;
    double v0 = dec_f64(atomicMax(&sc.slot[4 * l + 0], 0ull));
    double v1 = dec_f64(atomicMax(&sc.slot[4 * l + 1], 0ull));
    double v2 = dec_f64(atomicMax(&sc.slot[4 * l + 2], 0ull));
    v0 = wave_max(v0);
    v1 = wave_max(v1);
    v2 = wave_max(v2);
    if (l == 0 && !sc.ctrl->done) {  // (a batch that has stopped leaves its record alone)
        const double dmin = -v2, delta = fmax(fabs(v1), fabs(dmin));
        sc.result[0] = v0;
        sc.result[1] = v1;
        sc.result[2] = dmin;
        sc.result[3] = delta;
        sc.ctrl->k_done = sc.k + 1;
        if (sc.tol >= 0.0 && delta <= sc.tol) sc.ctrl->done = 1;
    }
}

__global__ void k_reset_stats(unsigned long long* slots, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) slots[i] = enc_f64(-INFINITY);
}

__global__ void k_begin_batch(Ctrl* ctrl) {
    ctrl->done = 0;
    ctrl->k_done = 0;
    ctrl->ticket = 0u;
    for (int i = 0; i < STAT_SHARDS; ++i) ctrl->shard_ticket[i] = 0u;
}

template <typename REAL, int N>
__global__ void k_terminal_cost(DevP P, REAL* __restrict__ J) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)(P.store_end - P.store_begin) * P.plane;
    if (s >= total) return;
    DevP Q = P;  // decode relative to the stored slab
    Q.row_begin = P.store_begin;
    int idx[N];
    decode_node<N>(Q, s, idx);
    double x[N], dx[N];
#pragma unroll
    for (int d = 0; d < N; ++d) {
        x[d] = P.lev[d][idx[d]];
        dx[d] = x[d] - P.xbar[d];
    }
    double h = quad_form<N>(P.S, dx);
    if (P.domain_check && !state_valid<N>(P, x)) h = P.INF;  // costfunction.py:385-387
    if (P.ontarget && l2norm<N>(dx) < P.EPS) h = 0.0;
    if (P.reach) h = (l2norm<N>(dx) < P.EPS) ? 0.0 : P.INF;   // Reachability.h (costfunction.py:454-466): target set or INF
    J[s] = (REAL)h;
}

// =================================================================================================
// the fused sweep, tier A (in-kernel dynamics), one thread per node, actions looped in registers.
// v0 gather path: J_k read straight through L1/L2.
// =================================================================================================
// SPARSE (4-D, state box = grid box, A <= 128): the actions whose cell lands in the box come from the 128-bit mask that
// k_valid_mask wrote at set-up (further down: the same float64 expressions), every lane walks the set bits of ITS mask with
// the cell arithmetic of the dense loop, and the cells outside -- Q = INF + alpha * 0 = INF -- enter the argmin as one
// candidate (INF, first clear bit).  This is the float32-storage path of systems whose float32 displacement cancels
// (the two-link arm): 88 % of its cells are outside the box.
template <int DYN, typename REAL, typename PI_T, bool LEVLDS, bool SPARSE = false>
__global__ __launch_bounds__(256) void k_sweep(DevP P, const REAL* __restrict__ Jin, REAL* __restrict__ Jout,
                                               PI_T* __restrict__ pi, double alpha, SweepCtl sc,
                                               const double* __restrict__ utab, const double* __restrict__ gutab,
                                               const int* __restrict__ aoktab, const uint4* __restrict__ vmask = nullptr) {
    using D = Dyn<DYN>;
    constexpr int DOF = D::DOF, N = 2 * DOF, M = D::M;
    static_assert(!SPARSE || DOF == 2, "validity masks are kept for 4-D grids");
    if (sc.ctrl->done) return;
    // grid levels: LDS copies when they fit (they are read several times per cell), else global memory
    extern __shared__ __attribute__((aligned(16))) double lev_lds[];
    const double* lev[N];
    {
        double* dst = lev_lds;
#pragma unroll
        for (int d = 0; d < N; ++d) {
            if constexpr (LEVLDS) {
                for (int i = threadIdx.x; i < P.dim[d]; i += blockDim.x) dst[i] = P.lev[d][i];
                lev[d] = dst;
                dst += P.dim[d];
            } else {
                lev[d] = P.lev[d];
            }
        }
        if constexpr (LEVLDS) __syncthreads();
    }
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long owned = (long long)(P.row_end - P.row_begin) * P.plane;
    double st_j = -INFINITY, st_dmax = -INFINITY, st_ndmin = -INFINITY;
    if (o < owned) {
        int idx[N];
        decode_node<N>(P, o, idx);
        double x[N], dx[N];
        long long self = (long long)(idx[0] - P.store_begin) * P.strd[0];
#pragma unroll
        for (int d = 0; d < N; ++d) {
            x[d] = lev[d][idx[d]];
            dx[d] = x[d] - P.xbar[d];
            if (d > 0) self += idx[d] * P.strd[d];
        }
        // state part of g (costfunction.py:195) and on-target zone (:199-202)
        const double gx = quad_form<N>(P.Q, dx);
        const bool on_target = P.ontarget && (l2norm<N>(dx) < P.EPS);

        // position rows of x_next = f*dt + x with f[0:dof] = dq: the same for every action
        bool pos_ok = true, pos_in = true, halo_bad = false;
        int ci[N];
        double y[N];
        long long base = 0;
#pragma unroll
        for (int i = 0; i < DOF; ++i) {
            const double xn = x[DOF + i] * P.dt + x[i];
            pos_ok = pos_ok && !(xn < P.lb[i]) && !(xn > P.ub[i]);
            pos_in = pos_in && !(xn < P.glo[i]) && !(xn > P.ghi[i]);
            ci[i] = find_interval(lev[i], P.dim[i], P.glo[i], P.inv_step[i], xn);
            y[i] = (xn - lev[i][ci[i]]) / (lev[i][ci[i] + 1] - lev[i][ci[i]]);
        }
        if (pos_in) {
            int r0 = ci[0];
            if (r0 < P.store_begin || r0 + 1 >= P.store_end) {
                halo_bad = true;
                r0 = min(max(r0, P.store_begin), P.store_end - 2);
            }
            base = (long long)(r0 - P.store_begin) * P.strd[0];
#pragma unroll
            for (int i = 1; i < DOF; ++i) base += ci[i] * P.strd[i];
        }
        if (halo_bad) atomicOr(&sc.ctrl->halo_err, 1);

        double tr[8];
        D::trig_from_tables(P, idx, tr);
        D dyn;
        dyn.init(P.c, x, tr);

        REAL best = (REAL)0;
        int arg = 0;
        const REAL alpha_r = (REAL)alpha;
        auto cell = [&](int a) -> REAL {
            double u[M], acc[DOF];
#pragma unroll
            for (int k = 0; k < M; ++k) u[k] = utab[a * M + k];
            dyn.accel(u, acc);
            const int aok_a = aoktab[a];  // unconditional: a wave-uniform scalar load
            bool ok = pos_ok && aok_a != 0, inb = pos_in;
            double xnv[DOF];
#pragma unroll
            for (int i = 0; i < DOF; ++i) {
                const int d = DOF + i;
                xnv[i] = acc[i] * P.dt + x[d];
                ok = ok && !(xnv[i] < P.lb[d]) && !(xnv[i] > P.ub[d]);
                inb = inb && !(xnv[i] < P.glo[d]) && !(xnv[i] > P.ghi[d]);
            }
            long long b = base;
            if (inb) {  // interval + fraction (a float64 division per axis) only where the value is used
#pragma unroll
                for (int i = 0; i < DOF; ++i) {
                    const int d = DOF + i;
                    double l0, l1;
                    ci[d] = find_interval_lv(lev[d], P.dim[d], P.glo[d], P.inv_step[d], xnv[i], l0, l1);
                    y[d] = (xnv[i] - l0) / (l1 - l0);
                    b += ci[d] * P.strd[d];
                }
            }
            // G (dynamicprogramming.py:534-549)
            const double g = on_target ? 0.0 : (gx + gutab[a]);
            const REAL G = ok ? (REAL)(g * P.dt) : (REAL)P.INF;
            const REAL Jn = inb ? Interp<REAL, N>::eval(Jin, P.strd, b, y) : (REAL)0;
            REAL q;
            if (sizeof(REAL) == 8)
                q = G + alpha_r * Jn;  // two roundings, as numpy (:567)
            else
                q = fmaf(alpha_r, Jn, G);
            return q;
        };
        if constexpr (SPARSE) {
            const uint4 mk = pos_in ? vmask[o] : make_uint4(0u, 0u, 0u, 0u);
            const unsigned w[4] = {mk.x, mk.y, mk.z, mk.w};
            int first_out = -1;
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                const unsigned z = ~w[k];
                const int i = 32 * k + __ffs((int)z) - 1;
                if (z && i < P.A) first_out = i;
            }
            bool have = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                unsigned m = w[k];
                while (m != 0u) {
                    const int a = 32 * k + __ffs((int)m) - 1;
                    m &= m - 1u;
                    const REAL q = cell(a);
                    if (!have || q < best) {
                        best = q;
                        arg = a;
                        have = true;
                    }
                }
            }
            const REAL inf_r = (REAL)P.INF;
            if (first_out >= 0 && (!have || inf_r < best || (inf_r == best && first_out < arg))) {
                best = inf_r;
                arg = first_out;
            }
        } else {
            for (int a = 0; a < P.A; ++a) {
                const REAL q = cell(a);
                if (a == 0 || q < best) {
                    best = q;
                    arg = a;
                }
            }
        }
        Jout[self] = best;
        pi[o] = (PI_T)arg;
        const double jn = (double)best, d = jn - (double)Jin[self];
        st_j = jn;
        st_dmax = d;
        st_ndmin = -d;
    }
    block_stats(st_j, st_dmax, st_ndmin, sc.slot);
    sweep_finish(sc);
}


// =================================================================================================
// Three-dimensional systems (n = 3; the reference's helicopter / car-parking / active-suspension demos).  They are not
// mechanical (no [q; dq] split), so they get their own closed forms and one thread-per-node sweep: x_next in float64 in
// the reference's operation order, validity = box + obstacle list, look-up-table semantics Q = G + alpha*J_interp with
// G = INF on invalid cells (dynamicprogramming.py:534-549, :567), scipy-order trilinear interpolation.
// Axes whose x_next does not depend on the action (UDEP bit clear) get their interval and fraction once per node.
// =================================================================================================
template <int DYN>
struct Dyn3;

// ConstantSpeedHelicopterTunnel (drone.py:613-636): dx = [1/mass * u, x0, vx].  c = [1/mass, vx]
template <>
struct Dyn3<PVI_DYN_HELICOPTER> {
    static constexpr int N = 3, M = 1, UDEP = 1;
    __device__ bool action_ok(const DevP&, const double*, int) const { return true; }
    __device__ void init(const DevP&, const int*, const double*) {}
    __device__ void f(const DevP& P, const double* x, const double* u, int, double* dx) const {
        dx[0] = P.c[0] * u[0];
        dx[1] = x[0];
        dx[2] = P.c[1];
    }
};

// KinematicBicyleModel (vehicle_steering.py:64-86): dx = [u0 cos x2, u0 sin x2, u0 tan(u1) (1/length)]
template <>
struct Dyn3<PVI_DYN_KINCAR> {
    static constexpr int N = 3, M = 2, UDEP = 7;
    double c2, s2;
    __device__ bool action_ok(const DevP&, const double*, int) const { return true; }
    __device__ void init(const DevP& P, const int* idx, const double*) {
        c2 = P.trig[0][idx[2]];
        s2 = P.trig[1][idx[2]];
    }
    __device__ void f(const DevP& P, const double*, const double* u, int a, double* dx) const {
        dx[0] = u[0] * c2;
        dx[1] = u[0] * s2;
        dx[2] = P.aux[a];
    }
};

// QuarterCarOnRoughTerrain (suspension.py:100-124): dx = [1/mass (u - k (x1 - z) - b (x0 - dz)), x0, vx]
// c = [1/mass, k, b, vx]; z, dz = ground height / slope at the node's x2 (host tables)
template <>
struct Dyn3<PVI_DYN_QUARTERCAR> {
    static constexpr int N = 3, M = 1, UDEP = 1;
    double ks, bs;  // k (x1 - z), b (x0 - dz)
    __device__ bool action_ok(const DevP&, const double*, int) const { return true; }
    __device__ void init(const DevP& P, const int* idx, const double* x) {
        ks = P.c[1] * (x[1] - P.trig[0][idx[2]]);
        bs = P.c[2] * (x[0] - P.trig[1][idx[2]]);
    }
    __device__ void f(const DevP& P, const double* x, const double* u, int, double* dx) const {
        dx[0] = P.c[0] * ((u[0] - ks) - bs);
        dx[1] = x[0];
        dx[2] = P.c[3];
    }
};

// HolonomicMobileRobot (vehicle_steering.py:238-259; :336-382 adds obstacle boxes around the point robot): dx = [u0, u1]
template <>
struct Dyn3<PVI_DYN_HOLONOMIC> {
    static constexpr int N = 2, M = 2, UDEP = 3;
    __device__ bool action_ok(const DevP&, const double*, int) const { return true; }
    __device__ void init(const DevP&, const int*, const double*) {}
    __device__ void f(const DevP&, const double*, const double* u, int, double* dx) const {
        dx[0] = u[0];
        dx[1] = u[1];
    }
};

// LongitudinalFrontWheelDriveCarWithWheelSlipInput (vehicle_propulsion.py:130-223), x = [x, v], u = [slip]:
//   mu = mu_max (2 / (1 + exp(-mu_slope slip)) - 1);  fd = 0.5 rho cdA v |v|;  a = (mu m g rr - fd) / (m (1 + mu ry))
//   dx = [v, a];  isavalidinput also rejects negative normal forces:  m g rr - m a ry < 0  or  m g rf + m a ry < 0.
// Per action (host NumPy, the reference's own expressions): aux[2a] = mu m g rr, aux[2a+1] = m (1 + mu ry); per level of
// axis 1: trig[0] = fd.  c = [m, ry, m g rr, m g rf]
template <>
struct Dyn3<PVI_DYN_LONGCAR> {
    static constexpr int N = 2, M = 1, UDEP = 2;
    double fd;
    __device__ void init(const DevP& P, const int* idx, const double*) { fd = P.trig[0][idx[1]]; }
    __device__ double acc(const DevP& P, int a) const { return (P.aux[2 * a] - fd) / P.aux[2 * a + 1]; }
    __device__ bool action_ok(const DevP& P, const double*, int a) const {
        const double ma_ry = (P.c[0] * acc(P, a)) * P.c[1];
        return !((P.c[2] - ma_ry) < 0.0) && !((P.c[3] + ma_ry) < 0.0);
    }
    __device__ void f(const DevP& P, const double* x, const double*, int a, double* dx) const {
        dx[0] = x[1];
        dx[1] = acc(P, a);
    }
};

template <int DYN, typename REAL, typename PI_T>
__global__ __launch_bounds__(256) void k_sweep3(DevP P, const REAL* __restrict__ Jin, REAL* __restrict__ Jout,
                                                PI_T* __restrict__ pi, double alpha, SweepCtl sc,
                                                const double* __restrict__ utab, const double* __restrict__ gutab,
                                                const int* __restrict__ aoktab) {
    using D = Dyn3<DYN>;
    constexpr int N = D::N, M = D::M;
    if (sc.ctrl->done) return;
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long owned = (long long)(P.row_end - P.row_begin) * P.plane;
    double st_j = -INFINITY, st_dmax = -INFINITY, st_ndmin = -INFINITY;
    if (o < owned) {
        int idx[N];
        decode_node<N>(P, o, idx);
        double x[N], dx[N];
        long long self = (long long)(idx[0] - P.store_begin) * P.strd[0];
#pragma unroll
        for (int d = 0; d < N; ++d) {
            x[d] = P.lev[d][idx[d]];
            dx[d] = x[d] - P.xbar[d];
            if (d > 0) self += idx[d] * P.strd[d];
        }
        // node part of g (costfunction.py:195-202; :403-414 with the domain check): INF on a rejected node, 0 on target
        const double gx = quad_form<N>(P.Q, dx);
        const bool on_target = P.ontarget && (l2norm<N>(dx) < P.EPS);
        const bool node_bad = P.domain_check && !state_valid<N>(P, x);
        D dyn;
        dyn.init(P, idx, x);
        // axes whose x_next is the same for every action: evaluated with a placeholder action
        double xn[N], y[N], u0[M];
        int ci[N];
        bool inb_fix = true;
#pragma unroll
        for (int k = 0; k < M; ++k) u0[k] = 0.0;
        {
            double f0[N];
            dyn.f(P, x, u0, 0, f0);
#pragma unroll
            for (int d = 0; d < N; ++d) {
                if (!((D::UDEP >> d) & 1)) {
                    xn[d] = f0[d] * P.dt + x[d];
                    inb_fix = inb_fix && !(xn[d] < P.glo[d]) && !(xn[d] > P.ghi[d]);
                    double l0, l1;
                    ci[d] = find_interval_lv(P.lev[d], P.dim[d], P.glo[d], P.inv_step[d], xn[d], l0, l1);
                    y[d] = (xn[d] - l0) / (l1 - l0);
                }
            }
        }
        REAL best = (REAL)0;
        int arg = 0;
        const REAL alpha_r = (REAL)alpha;
        for (int a = 0; a < P.A; ++a) {
            double u[M], fa[N];
#pragma unroll
            for (int k = 0; k < M; ++k) u[k] = utab[a * M + k];
            dyn.f(P, x, u, a, fa);
            bool inb = inb_fix;
#pragma unroll
            for (int d = 0; d < N; ++d) {
                if ((D::UDEP >> d) & 1) {
                    xn[d] = fa[d] * P.dt + x[d];  // discretizer.py:363
                    inb = inb && !(xn[d] < P.glo[d]) && !(xn[d] > P.ghi[d]);
                }
            }
            const bool ok = aoktab[a] != 0 && dyn.action_ok(P, x, a) && state_valid<N>(P, xn);
            REAL Jn = (REAL)0;
            if (inb) {
                long long b = 0;
#pragma unroll
                for (int d = 0; d < N; ++d) {
                    if ((D::UDEP >> d) & 1) {
                        double l0, l1;
                        ci[d] = find_interval_lv(P.lev[d], P.dim[d], P.glo[d], P.inv_step[d], xn[d], l0, l1);
                        y[d] = (xn[d] - l0) / (l1 - l0);
                    }
                    int c = ci[d];
                    if (d == 0) {
                        if (c < P.store_begin || c + 1 >= P.store_end) {
                            atomicOr(&sc.ctrl->halo_err, 1);
                            c = min(max(c, P.store_begin), P.store_end - 2);
                        }
                        c -= P.store_begin;
                    }
                    b += c * P.strd[d];
                }
                Jn = Interp<REAL, N>::eval(Jin, P.strd, b, y);
            }
            const double g = on_target ? 0.0 : (node_bad ? P.INF : (gx + gutab[a]));
            const REAL G = ok ? (REAL)(g * P.dt) : (REAL)P.INF;
            REAL q;
            if (sizeof(REAL) == 8)
                q = G + alpha_r * Jn;
            else
                q = fmaf(alpha_r, Jn, G);
            if (P.hard_inf && !ok) q = (REAL)P.INF;  // base class: exactly INF (dynamicprogramming.py:225-233)
            if (a == 0 || q < best) {
                best = q;
                arg = a;
            }
        }
        Jout[self] = best;
        pi[o] = (PI_T)arg;
        const double jn = (double)best, d = jn - (double)Jin[self];
        st_j = jn;
        st_dmax = d;
        st_ndmin = -d;
    }
    block_stats(st_j, st_dmax, st_ndmin, sc.slot);
    sweep_finish(sc);
}

// -------------------------------------------------------------------------------------------------
// float32 production form of the explicit systems ("fast3").  What k_sweep3 spends its time on does not change from sweep
// to sweep or does not need float64 levels:
//   * ok(node, a) = isavalidinput and isavalidstate(x_next) -- the obstacle boxes are a loop of float64 compares per cell
//     -- is decided ONCE at set-up by k_mask3 (the same float64 expressions) into one 64-bit mask per node (A <= 64);
//   * the interval of x_next on a linspace axis and its fraction come from t = (x_next - lo) * (1 / step) in float64
//     (floor, then ONE rounding of t - floor(t) to float32) instead of a level search and a float64 division: the same cell
//     and fraction up to 1e-16 of a cell (float32 handles carry 1e-5);
//   * axes whose x_next does not depend on the action keep their interval from the node prologue, as in k_sweep3.
// Gathers stay in global memory: the lanes of a wave are consecutive nodes of the LAST axis and every explicit system moves
// along it by a node-uniform amount, so each of the 2^n gathers of a wave is one coalesced row segment.
// Measured on the helicopter tunnel 201 x 201 x 401 x 11 (float32): see DESIGN.md section 4.6.
// -------------------------------------------------------------------------------------------------
template <int DYN>
__global__ __launch_bounds__(256) void k_mask3(DevP P, unsigned long long* __restrict__ okmask) {
    using D = Dyn3<DYN>;
    constexpr int N = D::N, M = D::M;
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long owned = (long long)(P.row_end - P.row_begin) * P.plane;
    if (o >= owned) return;
    int idx[N];
    decode_node<N>(P, o, idx);
    double x[N];
#pragma unroll
    for (int d = 0; d < N; ++d) x[d] = P.lev[d][idx[d]];
    D dyn;
    dyn.init(P, idx, x);
    unsigned long long m = 0ull;
    for (int a = 0; a < P.A; ++a) {
        double u[M], fa[N], xn[N];
#pragma unroll
        for (int k = 0; k < M; ++k) u[k] = P.utab[a * M + k];
        dyn.f(P, x, u, a, fa);
#pragma unroll
        for (int d = 0; d < N; ++d) xn[d] = fa[d] * P.dt + x[d];  // discretizer.py:363
        const bool ok = P.aok[a] != 0 && dyn.action_ok(P, x, a) && state_valid<N>(P, xn);
        if (ok) m |= 1ull << a;
    }
    okmask[o] = m;
}

// FBARGS: empty, or (float* jlo, double alpha64) = error-feedback storage (PVI_FLAG_F32_FEEDBACK): the residual of every node's
// stored J and the discount factor unrounded.  (A parameter pack, so that the plain instantiation keeps the argument list and the
// body it had when it ran on hardware: tools/kernel_manifest.py.)
__device__ __forceinline__ float* fb_jlo() { return nullptr; }
__device__ __forceinline__ float* fb_jlo(float* jlo, double) { return jlo; }
__device__ __forceinline__ double fb_alpha() { return 0.0; }
__device__ __forceinline__ double fb_alpha(float*, double alpha64) { return alpha64; }
template <int DYN, typename PI_T, typename... FBARGS>
__global__ __launch_bounds__(256) void k_sweep3_fast(DevP P, const float* __restrict__ Jin, float* __restrict__ Jout, PI_T* __restrict__ pi,
                                                     float alpha, SweepCtl sc, const double* __restrict__ utab,
                                                     const double* __restrict__ gutab, const unsigned long long* __restrict__ okmask,
                                                     FBARGS... fbargs) {
    constexpr bool FB = sizeof...(FBARGS) == 2;
    static_assert(FB || sizeof...(FBARGS) == 0, "k_sweep3_fast: no extra arguments, or (float* jlo, double alpha64)");
    using D = Dyn3<DYN>;
    constexpr int N = D::N, M = D::M;
    if (sc.ctrl->done) return;
    // (Contiguous block ranges per XCD along axis 0, as in k_sweep64, were measured in round 4: 0.478 against 0.458 ms on the
    //  201 x 201 x 401 helicopter grid.  The split along axis 1 below takes the bytes this sweep moves through the fabric from
    //  1.15 to 0.13 GB of reads per launch at the same 0.46 ms: the sweep does not wait for them, but they are no longer moved.)
    const unsigned lb = blockIdx.x;
    long long o = (long long)lb * blockDim.x + threadIdx.x;
    const long long owned = (long long)(P.row_end - P.row_begin) * P.plane;
    int idx[N];
    bool live = o < owned;
    bool decoded = false;
    if constexpr (N == 3) {
      if (sc.xcd_remap == 3) {
        decoded = true;
        // XCD x (= block b % 8) sweeps ITS eighth of axis 1 for every row of axis 0 in turn.  The gathers of a node go to the
        // planes i0 + d0(a) of ALL actions -- twenty-odd planes of axes (1, 2) -- at its own (i1, i2) plus a small shift: with
        // an eighth of axis 1 per XCD those planes' strips (22 x 27 rows x 1.6 KB on the 201 x 201 x 401 helicopter grid:
        // 1 MB) stay in the XCD's 4 MB L2 while axis 0 advances, instead of 22 whole planes (7 MB) per XCD.
        const int x8 = (int)(lb & 7u), c0 = (int)((long long)P.dim[1] * x8 / 8), c1 = (int)((long long)P.dim[1] * (x8 + 1) / 8);
        const long long per0 = (long long)(c1 - c0) * P.dim[2], n = (long long)(lb >> 3) * blockDim.x + threadIdx.x;
        const int r = (int)(n / per0);
        const int rem = (int)(n - (long long)r * per0);
        live = r < P.row_end - P.row_begin;
        idx[0] = P.row_begin + r;
        idx[1] = c0 + rem / P.dim[2];
        idx[2] = rem - (rem / P.dim[2]) * P.dim[2];
        o = ((long long)r * P.dim[1] + idx[1]) * P.dim[2] + idx[2];
      }
    }
    if (!decoded && live) decode_node<N>(P, o, idx);
    double st_j = -INFINITY, st_dmax = -INFINITY, st_ndmin = -INFINITY;
    if (live) {
        double x[N], dx[N];
        long long self = (long long)(idx[0] - P.store_begin) * P.strd[0];
#pragma unroll
        for (int d = 0; d < N; ++d) {
            x[d] = P.lev[d][idx[d]];
            dx[d] = x[d] - P.xbar[d];
            if (d > 0) self += idx[d] * P.strd[d];
        }
        const double gx = quad_form<N>(P.Q, dx);
        const bool on_target = P.ontarget && (l2norm<N>(dx) < P.EPS);
        const bool node_bad = P.domain_check && !state_valid<N>(P, x);
        const unsigned long long okm = okmask[o];
        D dyn;
        dyn.init(P, idx, x);
        double xn[N], u0[M], y64[N];
        float y[N];
        int ci[N];
        bool inb_fix = true;
#pragma unroll
        for (int k = 0; k < M; ++k) u0[k] = 0.0;
        {
            double f0[N];
            dyn.f(P, x, u0, 0, f0);
#pragma unroll
            for (int d = 0; d < N; ++d) {
                if (!((D::UDEP >> d) & 1)) {
                    xn[d] = f0[d] * P.dt + x[d];
                    inb_fix = inb_fix && !(xn[d] < P.glo[d]) && !(xn[d] > P.ghi[d]);
                    double l0, l1;
                    ci[d] = find_interval_lv(P.lev[d], P.dim[d], P.glo[d], P.inv_step[d], xn[d], l0, l1);
                    y64[d] = (xn[d] - l0) / (l1 - l0);
                    y[d] = (float)y64[d];
                }
            }
        }
        float best = 0.f;
        int arg = 0;
        const float INF_F = (float)P.INF;
        for (int a = 0; a < P.A; ++a) {
            double u[M], fa[N];
#pragma unroll
            for (int k = 0; k < M; ++k) u[k] = utab[a * M + k];  // (wave-uniform: scalar loads)
            dyn.f(P, x, u, a, fa);
            bool inb = inb_fix;
#pragma unroll
            for (int d = 0; d < N; ++d) {
                if ((D::UDEP >> d) & 1) {
                    xn[d] = fa[d] * P.dt + x[d];
                    inb = inb && !(xn[d] < P.glo[d]) && !(xn[d] > P.ghi[d]);
                    const double t = (xn[d] - P.glo[d]) * P.inv_step[d];
                    double fl = floor(t);
                    fl = fl < 0.0 ? 0.0 : (fl > (double)(P.dim[d] - 2) ? (double)(P.dim[d] - 2) : fl);
                    ci[d] = (int)fl;
                    y[d] = (float)(t - fl);
                }
            }
            const bool ok = (okm >> a) & 1ull;
            float Jn = 0.f;
            if (inb) {
                long long b = 0;
#pragma unroll
                for (int d = 0; d < N; ++d) {
                    int c = ci[d];
                    if (d == 0) {
                        if (c < P.store_begin || c + 1 >= P.store_end) {
                            atomicOr(&sc.ctrl->halo_err, 1);
                            c = min(max(c, P.store_begin), P.store_end - 2);
                        }
                        c -= P.store_begin;
                    }
                    b += c * P.strd[d];
                }
                Jn = interp_f32<N>(Jin, P.strd, b, y);
            }
            const double g = on_target ? 0.0 : (node_bad ? P.INF : (gx + gutab[a]));
            const float G = ok ? (float)(g * P.dt) : INF_F;
            float q = fmaf(alpha, Jn, G);
            if (P.hard_inf && !ok) q = INF_F;
            if (a == 0 || q < best) {
                best = q;
                arg = a;
            }
        }
        if constexpr (FB) {
            // Error-feedback storage (PVI_FLAG_F32_FEEDBACK): the backup of the chosen action once more, the
            // float32 gathers combined in float64 with the unrounded fractions and cost, plus the residual of the node's last
            // store; what this store drops is the next residual.  (An action the base class rejects costs INF exactly.)
            float* const jlo = fb_jlo(fbargs...);
            const double alpha64 = fb_alpha(fbargs...);
            const float lo_old = jlo[o];
            float lo_new = 0.f;
            const bool ok = (okm >> arg) & 1ull;
            if (!(P.hard_inf && !ok)) {
                double u[M], fa[N];
#pragma unroll
                for (int k = 0; k < M; ++k) u[k] = utab[arg * M + k];
                dyn.f(P, x, u, arg, fa);
                bool inb = inb_fix;
#pragma unroll
                for (int d = 0; d < N; ++d) {
                    if ((D::UDEP >> d) & 1) {
                        xn[d] = fa[d] * P.dt + x[d];
                        inb = inb && !(xn[d] < P.glo[d]) && !(xn[d] > P.ghi[d]);
                        const double t = (xn[d] - P.glo[d]) * P.inv_step[d];
                        double fl = floor(t);
                        fl = fl < 0.0 ? 0.0 : (fl > (double)(P.dim[d] - 2) ? (double)(P.dim[d] - 2) : fl);
                        ci[d] = (int)fl;
                        y64[d] = t - fl;
                    }
                }
                double Jn = 0.0;
                if (inb) {
                    long long b = 0;
#pragma unroll
                    for (int d = 0; d < N; ++d) {
                        int c = ci[d];
                        if (d == 0) c = min(max(c, P.store_begin), P.store_end - 2) - P.store_begin;
                        b += c * P.strd[d];
                    }
                    Jn = interp_f32_in_f64<N>(Jin, P.strd, b, y64);
                }
                const double g = on_target ? 0.0 : (node_bad ? P.INF : (gx + gutab[arg]));
                const double t = __builtin_fma(alpha64, Jn, ok ? g * P.dt : P.INF) + (double)lo_old;
                best = (float)t;
                if (best < INFINITY && best > -INFINITY) lo_new = (float)(t - (double)best);   // (an INF beyond float32 leaves no residual)
            }
            jlo[o] = lo_new;
        }
        Jout[self] = best;
        pi[o] = (PI_T)arg;
        const double jn = (double)best, d = jn - (double)Jin[self];
        st_j = jn;
        st_dmax = d;
        st_ndmin = -d;
    }
    block_stats(st_j, st_dmax, st_ndmin, sc.slot);
    sweep_finish(sc);
}

// reference tables of the n = 3 systems: x_next_table, x_next_isok, action_isok, G (one thread per cell)
template <int DYN>
__global__ void k_build_tables3(DevP P, long long node0, long long nnodes, double* __restrict__ xnext,
                                unsigned char* __restrict__ xok, unsigned char* __restrict__ aok,
                                double* __restrict__ G) {
    using D = Dyn3<DYN>;
    constexpr int N = D::N, M = D::M;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nnodes * P.A) return;
    const long long ln = t / P.A;
    const int a = (int)(t - ln * P.A);
    DevP Q = P;
    Q.row_begin = 0;
    int idx[N];
    decode_node<N>(Q, node0 + ln, idx);
    double x[N], dx[N], u[M], fa[N], xn[N];
#pragma unroll
    for (int d = 0; d < N; ++d) {
        x[d] = P.lev[d][idx[d]];
        dx[d] = x[d] - P.xbar[d];
    }
    D dyn;
    dyn.init(P, idx, x);
#pragma unroll
    for (int k = 0; k < M; ++k) u[k] = P.utab[a * M + k];
    dyn.f(P, x, u, a, fa);
#pragma unroll
    for (int d = 0; d < N; ++d) xn[d] = fa[d] * P.dt + x[d];
    const bool ok = state_valid<N>(P, xn);
    const bool a_ok = P.aok[a] && dyn.action_ok(P, x, a);
    if (xnext) {
#pragma unroll
        for (int d = 0; d < N; ++d) xnext[t * N + d] = xn[d];
    }
    if (xok) xok[t] = ok;
    if (aok) aok[t] = a_ok;
    if (G) {
        const bool on_target = P.ontarget && (l2norm<N>(dx) < P.EPS);
        const bool node_bad = P.domain_check && !state_valid<N>(P, x);
        const double g = on_target ? 0.0 : (node_bad ? P.INF : (quad_form<N>(P.Q, dx) + P.gu[a]));
        G[t] = (ok && a_ok) ? g * P.dt : P.INF;
    }
}

// =================================================================================================
// tier B: table-driven sweep for arbitrary sys.f / cf.g (dynamicprogramming.py:564-570 verbatim:
// Q = G + alpha * J_interp(x_next_table)).  x_next [node][A][N] f64, G [node][A] f64: N*8 + 8 bytes per cell, read
// exactly once per sweep -- this tier is HBM-bound, so the tables are streamed with fully coalesced loads:
// a workgroup owns `npb` consecutive nodes, its lanes walk the nodes' cells in memory order (lane = cell), write
// Q to LDS, and one thread per node then scans its A values for the first minimum (np.argmin).  Action counts too
// large for the LDS row are processed in chunks of `achunk` actions.
// =================================================================================================
template <int N, typename REAL, typename PI_T, bool LEVLDS>
__global__ __launch_bounds__(256) void k_sweep_table(DevP P, const double* __restrict__ xnext,
                                                     const double* __restrict__ Gt,
                                                     const unsigned char* __restrict__ okt,
                                                     const REAL* __restrict__ Jin,
                                                     REAL* __restrict__ Jout, PI_T* __restrict__ pi, double alpha,
                                                     SweepCtl sc, int npb, int achunk, int qs_doubles, int lpn_log2) {
    extern __shared__ __attribute__((aligned(16))) double qs_raw[];
    REAL* qs = (REAL*)qs_raw;
    if (sc.ctrl->done) return;
    // grid levels: LDS copies behind the Q rows when they fit (lev_lds), else read from global memory
    const double* lev[N];
    {
        double* dst = qs_raw + qs_doubles;
#pragma unroll
        for (int d = 0; d < N; ++d) {
            if constexpr (LEVLDS) {
                for (int i = threadIdx.x; i < P.dim[d]; i += blockDim.x) dst[i] = P.lev[d][i];
                lev[d] = dst;
                dst += P.dim[d];
            } else {
                lev[d] = P.lev[d];
            }
        }
        if constexpr (LEVLDS) __syncthreads();
    }
    const long long owned = (long long)(P.row_end - P.row_begin) * P.plane;
    const long long n0 = (long long)blockIdx.x * npb;
    const int nn = (int)min((long long)npb, owned - n0);
    const REAL alpha_r = (REAL)alpha;
    const int lpn = 1 << lpn_log2, sn = threadIdx.x >> lpn_log2, sj = threadIdx.x & (lpn - 1);
    REAL best = (REAL)0;
    int arg = 0;
    for (int a0 = 0; a0 < P.A; a0 += achunk) {
        const int ac = min(achunk, P.A - a0);
        const int ncell = nn * ac;
        // the next cell's table entries are requested before the current cell is evaluated (one memory round trip
        // per cell would otherwise sit in front of ~150 dependent instructions)
        struct CellIn {
            double x[N], g;
            unsigned char ok;
        };
        auto fetch = [&](int lc, CellIn& c) {
            const int ln = lc / ac, a = a0 + (lc - ln * ac);
            const long long cell = (n0 + ln) * P.A + a;  // one chunk: consecutive lanes = consecutive cells in memory
            if constexpr (N == 2) {
                const double2 t = *(const double2*)(xnext + cell * 2);
                c.x[0] = t.x;
                c.x[1] = t.y;
            } else if constexpr (N == 4) {
                const double4 t = *(const double4*)(xnext + cell * 4);
                c.x[0] = t.x;
                c.x[1] = t.y;
                c.x[2] = t.z;
                c.x[3] = t.w;
            } else {
#pragma unroll
                for (int d = 0; d < N; ++d) c.x[d] = xnext[cell * N + d];
            }
            c.g = Gt[cell];
            c.ok = okt ? okt[cell] : (unsigned char)1;
        };
        CellIn cur, nxt;
        if ((int)threadIdx.x < ncell) fetch(threadIdx.x, cur);
        for (int lc = threadIdx.x; lc < ncell; lc += blockDim.x) {
            if (lc + (int)blockDim.x < ncell) fetch(lc + blockDim.x, nxt);
            bool inb = true;
            int ci[N];
            double y[N];
            long long b = 0;
#pragma unroll
            for (int d = 0; d < N; ++d) {
                const double v = cur.x[d];
                inb = inb && !(v < P.glo[d]) && !(v > P.ghi[d]);
                double l0, l1;
                ci[d] = find_interval_lv(lev[d], P.dim[d], P.glo[d], P.inv_step[d], v, l0, l1);
                y[d] = (v - l0) / (l1 - l0);
                if (P.nearest) y[d] = y[d] <= 0.5 ? 0.0 : 1.0;  // method='nearest' (see k_table_pack)
                int c = ci[d];
                if (d == 0) {
                    if (inb && (c < P.store_begin || c + 1 >= P.store_end)) atomicOr(&sc.ctrl->halo_err, 1);
                    c = min(max(c, P.store_begin), P.store_end - 2) - P.store_begin;
                }
                b += c * P.strd[d];
            }
            const REAL G = (REAL)cur.g;
            const REAL Jn = inb ? Interp<REAL, N>::eval(Jin, P.strd, b, y) : (REAL)0;
            REAL q;
            if (sizeof(REAL) == 8)
                q = G + alpha_r * Jn;
            else
                q = fmaf(alpha_r, Jn, G);
            // base-class semantics (dynamicprogramming.py:195-236): an invalid action / next state costs
            // exactly INF, not INF + alpha*J as in the look-up-table class (:567)
            if (!cur.ok) q = (REAL)P.INF;
            qs[lc] = q;
            cur = nxt;
        }
        __syncthreads();
        if (sn < nn) {  // first-minimum scan of the node's Q row, shared by 2^lpn_log2 neighbouring lanes
            const int per = (ac + lpn - 1) >> lpn_log2, k0 = sj * per, k1 = min(ac, k0 + per);
            const REAL* row = qs + sn * ac;
            REAL m = (REAL)0;
            int mi = 0x7fffffff;
            for (int k = k0; k < k1; ++k) {
                const REAL q = row[k];
                if (mi == 0x7fffffff || q < m) {
                    m = q;
                    mi = a0 + k;
                }
            }
            for (int off = lpn >> 1; off > 0; off >>= 1) {
                const REAL m2 = __shfl_xor(m, off, 64);
                const int i2 = __shfl_xor(mi, off, 64);
                if (i2 != 0x7fffffff && (mi == 0x7fffffff || m2 < m || (m2 == m && i2 < mi))) {
                    m = m2;
                    mi = i2;
                }
            }
            if (mi != 0x7fffffff && (a0 == 0 || m < best)) {
                best = m;
                arg = mi;
            }
        }
        __syncthreads();
    }
    double st_j = -INFINITY, st_dmax = -INFINITY, st_ndmin = -INFINITY;
    if (sn < nn && sj == 0) {
        const long long o = n0 + sn;
        const long long self = o + (long long)(P.row_begin - P.store_begin) * P.plane;
        Jout[self] = best;
        pi[o] = (PI_T)arg;
        const double jn = (double)best, d = jn - (double)Jin[self];
        st_j = jn;
        st_dmax = d;
        st_ndmin = -d;
    }
    block_stats(st_j, st_dmax, st_ndmin, sc.slot);
    sweep_finish(sc);
}

// =================================================================================================
// tier B, packed: the reference-layout tables are packed once (pvi_set_tables) into one record per cell,
//   { int32 offset of corner 0 in the stored J buffer (-1: x_next outside the grid -> J_interp = 0),
//     one fraction per axis, G }        float32 handles: 16 / 20 / 24 bytes for n = 2 / 3 / 4 (tables: 24 / 32 / 40),
//                                       float64 handles: 32 / 40 / 48 bytes, fractions and G kept in float64,
// with interval search, division and validity done in float64 exactly as k_sweep_table does them per sweep -- so the
// float64 records reproduce that kernel bit for bit while the sweep becomes a pure stream: record, 2^(n-1) corner-pair
// gathers, the interpolation sum, Q, running first minimum.  Records are stored action-major inside blocks of TAB_NB
// nodes ([block][action][node]): lane = node, consecutive lanes read consecutive records, no LDS, no scan.
// =================================================================================================
template <int N, typename REAL>
struct TabRec {
    int base;
    REAL y[N];
    REAL G;
};
#define TAB_NB 1024  // nodes per block of the packed layout (four workgroups of 256 lanes read one block)

template <int N, typename REAL>
__global__ __launch_bounds__(256) void k_table_pack(DevP P, const double* __restrict__ xnext, const double* __restrict__ Gt,
                                                    const unsigned char* __restrict__ okt, TabRec<N, REAL>* __restrict__ out,
                                                    long long cell0, long long cells, int* __restrict__ halo_err) {
    // the tables arrive in chunks: `xnext`, `Gt`, `okt` hold cells [cell0, cell0 + cells) of the reference layout
    const long long lc = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (lc >= cells) return;
    const long long cell = cell0 + lc;
    bool inb = true;
    long long b = 0;
    TabRec<N, REAL> r;
#pragma unroll
    for (int d = 0; d < N; ++d) {
        const double v = xnext[lc * N + d];
        inb = inb && !(v < P.glo[d]) && !(v > P.ghi[d]);
        double l0, l1;
        const int ci = find_interval_lv(P.lev[d], P.dim[d], P.glo[d], P.inv_step[d], v, l0, l1);
        double yd = (v - l0) / (l1 - l0);
        // scipy _evaluate_nearest: idx = where(y <= .5, i, i + 1) per axis -- as weights 1 / 0 of the linear form (exact)
        if (P.nearest) yd = yd <= 0.5 ? 0.0 : 1.0;
        r.y[d] = (REAL)yd;
        int c = ci;
        if (d == 0) {
            if (inb && (c < P.store_begin || c + 1 >= P.store_end)) atomicOr(halo_err, 1);
            c = min(max(c, P.store_begin), P.store_end - 2) - P.store_begin;
        }
        b += c * P.strd[d];
    }
    REAL G = (REAL)Gt[lc];
    // base-class semantics (an invalid cell costs exactly INF, dynamicprogramming.py:225-233) = INF + alpha * 0
    if (okt && !okt[lc]) {
        inb = false;
        G = (REAL)P.INF;
    }
    r.base = inb ? (int)b : -1;
    r.G = G;
    const long long o = cell / P.A;
    const int a = (int)(cell - o * P.A);
    out[((o / TAB_NB) * P.A + a) * TAB_NB + (o % TAB_NB)] = r;
}

template <int N, typename REAL, typename PI_T>
__global__ __launch_bounds__(256) void k_sweep_tablep(DevP P, const TabRec<N, REAL>* __restrict__ rec,
                                                      const REAL* __restrict__ Jin, REAL* __restrict__ Jout,
                                                      PI_T* __restrict__ pi, double alpha, SweepCtl sc) {
    if (sc.ctrl->done) return;
    const long long owned = (long long)(P.row_end - P.row_begin) * P.plane;
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    const TabRec<N, REAL>* __restrict__ r = rec + (o / TAB_NB) * P.A * TAB_NB + (o % TAB_NB);
    const REAL alpha_r = (REAL)alpha;
    double st_j = -INFINITY, st_dmax = -INFINITY, st_ndmin = -INFINITY;
    if (o < owned) {
        // The stream is bound by (bytes in flight per wave) / latency: actions are taken U at a time, the next U records
        // requested before the current ones are gathered and evaluated.
        constexpr int U = sizeof(REAL) == 4 ? 4 : 2;
        REAL best = (REAL)0;
        int arg = -1;
        TabRec<N, REAL> cur[U], nxt[U];
        const int A = P.A;
#pragma unroll
        for (int k = 0; k < U; ++k)
            if (k < A) cur[k] = r[(long long)k * TAB_NB];
        for (int a0 = 0; a0 < A; a0 += U) {
#pragma unroll
            for (int k = 0; k < U; ++k)
                if (a0 + U + k < A) nxt[k] = r[(long long)(a0 + U + k) * TAB_NB];
            REAL q[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                q[k] = cur[k].G;
                if (a0 + k < A && cur[k].base >= 0) {
                    if constexpr (sizeof(REAL) == 8) {
                        double yd[N];
#pragma unroll
                        for (int d = 0; d < N; ++d) yd[d] = cur[k].y[d];
                        q[k] = cur[k].G + alpha_r * interp_f64<N>((const double*)Jin, P.strd, (long long)cur[k].base, yd);
                    } else {
                        float yf[N];
#pragma unroll
                        for (int d = 0; d < N; ++d) yf[d] = cur[k].y[d];
                        q[k] = fmaf(alpha_r, interp_f32<N>((const float*)Jin, P.strd, (long long)cur[k].base, yf), cur[k].G);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < U; ++k)
                if (a0 + k < A && (arg < 0 || q[k] < best)) {
                    best = q[k];
                    arg = a0 + k;
                }
#pragma unroll
            for (int k = 0; k < U; ++k) cur[k] = nxt[k];
        }
        const long long self = o + (long long)(P.row_begin - P.store_begin) * P.plane;
        Jout[self] = best;
        pi[o] = (PI_T)arg;
        const double jn = (double)best, d = jn - (double)Jin[self];
        st_j = jn;
        st_dmax = d;
        st_ndmin = -d;
    }
    block_stats(st_j, st_dmax, st_ndmin, sc.slot);
    sweep_finish(sc);
}

#include "sweep_spline.inc"

// =================================================================================================
// reference tables for a block of rows: x_next_table, x_next_isok, action_isok, G
// (discretizer.py:342-376, :314-338; dynamicprogramming.py:517-553).  One thread per (node, action).
// =================================================================================================
template <int DYN>
__global__ void k_build_tables(DevP P, long long node0, long long nnodes, double* __restrict__ xnext,
                               unsigned char* __restrict__ xok, unsigned char* __restrict__ aok,
                               double* __restrict__ G) {
    using D = Dyn<DYN>;
    constexpr int DOF = D::DOF, N = 2 * DOF, M = D::M;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nnodes * P.A) return;
    const long long ln = t / P.A;
    const int a = (int)(t - ln * P.A);
    DevP Q = P;
    Q.row_begin = 0;
    int idx[N];
    decode_node<N>(Q, node0 + ln, idx);
    double x[N], dx[N], tr[8], u[M], acc[DOF];
#pragma unroll
    for (int d = 0; d < N; ++d) {
        x[d] = P.lev[d][idx[d]];
        dx[d] = x[d] - P.xbar[d];
    }
    D::trig_from_tables(P, idx, tr);
    D dyn;
    dyn.init(P.c, x, tr);
#pragma unroll
    for (int k = 0; k < M; ++k) u[k] = P.utab[a * M + k];
    dyn.accel(u, acc);
    bool ok = true;
    double xn[N];
#pragma unroll
    for (int i = 0; i < DOF; ++i) {
        xn[i] = x[DOF + i] * P.dt + x[i];
        xn[DOF + i] = acc[i] * P.dt + x[DOF + i];
    }
#pragma unroll
    for (int d = 0; d < N; ++d) ok = ok && !(xn[d] < P.lb[d]) && !(xn[d] > P.ub[d]);
    if (xnext) {
#pragma unroll
        for (int d = 0; d < N; ++d) xnext[t * N + d] = xn[d];
    }
    if (xok) xok[t] = ok;
    if (aok) aok[t] = P.aok[a];
    if (G) {
        const bool on_target = P.ontarget && (l2norm<N>(dx) < P.EPS);
        const double g = on_target ? 0.0 : (quad_form<N>(P.Q, dx) + P.gu[a]);
        G[t] = (ok && P.aok[a]) ? g * P.dt : P.INF;
    }
}

// =================================================================================================
// policy evaluation tables (dynamicprogramming.py:704-735): one control input per node,
//   u = ctl.c(x, rbar, t) ; x_next = f(x, u) dt + x ; ok = isavalidinput(x, u) and isavalidstate(x_next) ; G = g(x, u) dt | INF
// with the control law evaluated in-kernel where it is the reference's ComputedTorqueController on a fully actuated closed
// form (pyro/control/nonlinear.py:23-116; mechanical.py:186-214):
//   ddq_r = (0 - (2 zeta w0) dq_e) - (w0^2) q_e ;  forces = ((H ddq_r + C dq) + g) + d ;  u = inv(B) forces, B = I
// in that operation order (1-dof: the reference's bits; 2-dof: its 2x2 BLAS dots may use fused multiply-adds).
// =================================================================================================
template <int DYN>
struct CtForces;
template <>
struct CtForces<PVI_DYN_PENDULUM> {  // pendulum.py:80-150; c[3] = H = m1 lc1^2 + I1
    __device__ static void u(const double* c, const double* x, const double* tr, const double* ddq, double* u) {
        const double f = ((c[3] * ddq[0] + 0.0 * x[1]) + c[1] * tr[0]) + c[2] * x[1];
        u[0] = 1.0 * f;
    }
};
template <>
struct CtForces<PVI_DYN_TWOLINK> {  // manipulator.py:897-992 / pendulum.py:400-493, the terms of Dyn<PVI_DYN_TWOLINK>::init
    __device__ static void u(const double* c, const double* x, const double* tr, const double* ddq, double* u) {
        const double s1 = tr[0], c2 = tr[1], s2 = tr[2], s12 = tr[3];
        const double dq0 = x[2], dq1 = x[3];
        const double H00 = (c[0] + c[1] * (c[2] + c[3] * c2)) + c[4];
        const double H01 = (c[5] + c[6] * c2) + c[4];
        const double H11 = c[5] + c[4];
        const double h = c[6] * s2;
        const double C00 = -h * dq1, C10 = h * dq0, C01 = -h * (dq0 + dq1);
        const double G0 = -c[7] * s1 - c[8] * s12, G1 = -c[8] * s12;
        const double f0 = (((H00 * ddq[0] + H01 * ddq[1]) + (C00 * dq0 + C01 * dq1)) + G0) + c[9] * dq0;
        const double f1 = (((H01 * ddq[0] + H11 * ddq[1]) + (C10 * dq0 + 0.0 * dq1)) + G1) + c[10] * dq1;
        u[0] = 1.0 * f0 + 0.0 * f1;
        u[1] = 0.0 * f0 + 1.0 * f1;
    }
};
template <>
struct CtForces<PVI_DYN_CARTPOLE> {  // (under-actuated: the reference raises NotImplementedError; never launched)
    __device__ static void u(const double*, const double*, const double*, const double*, double* u) { u[0] = 0.0; }
};

template <int DYN>
__global__ void k_policy_tables(DevP P, int controller_id, const double* __restrict__ ctl, double* __restrict__ U,
                                double* __restrict__ xnext, unsigned char* __restrict__ okout, double* __restrict__ G, long long nodes) {
    using D = Dyn<DYN>;
    constexpr int DOF = D::DOF, N = 2 * DOF, M = D::M;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nodes) return;
    DevP Q = P;
    Q.row_begin = 0;
    int idx[N];
    decode_node<N>(Q, t, idx);
    double x[N], dx[N], tr[8], u[M], acc[DOF];
#pragma unroll
    for (int d = 0; d < N; ++d) {
        x[d] = P.lev[d][idx[d]];
        dx[d] = x[d] - P.xbar[d];
    }
    D::trig_from_tables(P, idx, tr);
    if (controller_id == 1) {
        double ddq[DOF];
#pragma unroll
        for (int i = 0; i < DOF; ++i) {
            const double q_e = x[i] - ctl[i], dq_e = x[DOF + i] - 0.0;
            ddq[i] = (0.0 - ctl[DOF] * dq_e) - ctl[DOF + 1] * q_e;
        }
        CtForces<DYN>::u(P.c, x, tr, ddq, u);
#pragma unroll
        for (int k = 0; k < M; ++k) U[t * M + k] = u[k];
    } else {
#pragma unroll
        for (int k = 0; k < M; ++k) u[k] = U[t * M + k];
    }
    D dyn;
    dyn.init(P.c, x, tr);
    dyn.accel(u, acc);
    bool ok = true;
    double xn[N], du[M];
#pragma unroll
    for (int i = 0; i < DOF; ++i) {
        xn[i] = x[DOF + i] * P.dt + x[i];
        xn[DOF + i] = acc[i] * P.dt + x[DOF + i];
    }
#pragma unroll
    for (int k = 0; k < M; ++k) {
        ok = ok && !(u[k] < P.ulb[k]) && !(u[k] > P.uub[k]);  // isavalidinput, system.py:208-215
        du[k] = u[k] - P.ubar[k];
    }
#pragma unroll
    for (int d = 0; d < N; ++d) ok = ok && !(xn[d] < P.lb[d]) && !(xn[d] > P.ub[d]);
#pragma unroll
    for (int d = 0; d < N; ++d) xnext[t * N + d] = xn[d];
    okout[t] = ok;
    const bool on_target = P.ontarget && (l2norm<N>(dx) < P.EPS);
    const double g = on_target ? 0.0 : (quad_form<N>(P.Q, dx) + quad_form<M>(P.R, du));
    G[t] = ok ? g * P.dt : P.INF;
}

// batched f(x,u) with in-kernel trig (mechanical.py:238-263)
template <int DYN>
__global__ void k_eval_f(const double* __restrict__ c16, long long B, const double* __restrict__ X,
                         const double* __restrict__ U, double* __restrict__ dX) {
    using D = Dyn<DYN>;
    constexpr int DOF = D::DOF, N = 2 * DOF, M = D::M;
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double c[16], x[N], u[M], tr[8], acc[DOF];
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = c16[i];
#pragma unroll
    for (int d = 0; d < N; ++d) x[d] = X[b * N + d];
#pragma unroll
    for (int k = 0; k < M; ++k) u[k] = U[b * M + k];
    D::trig_from_state(x, tr);
    D dyn;
    dyn.init(c, x, tr);
    dyn.accel(u, acc);
#pragma unroll
    for (int i = 0; i < DOF; ++i) {
        dX[b * N + i] = x[DOF + i];
        dX[b * N + DOF + i] = acc[i];
    }
}


// =================================================================================================
// batched closed-loop Euler rollouts of the look-up-table policy (one thread per trajectory, float64)
// =================================================================================================
// u_k = n-linear interpolation of input_from_action_id[pi[node], k] over the grid, 0 outside (LookUpTableController.c,
// dynamicprogramming.py:72-107; scipy's corner order)
template <int N, int M, typename PI_T>
__device__ inline void rollout_policy(const DevP& P, const PI_T* __restrict__ pi, const double* x, double* u) {
    double y[N];
    int ci[N];
    bool inb = true;
    long long base = 0;
#pragma unroll
    for (int d = 0; d < N; ++d) {
        inb = inb && !(x[d] < P.glo[d]) && !(x[d] > P.ghi[d]);
        ci[d] = find_interval(P.lev[d], P.dim[d], P.glo[d], P.inv_step[d], x[d]);
        y[d] = (x[d] - P.lev[d][ci[d]]) / (P.lev[d][ci[d] + 1] - P.lev[d][ci[d]]);
        base += ci[d] * P.strd[d];
    }
#pragma unroll
    for (int k = 0; k < M; ++k) {
        double val = 0.0;
        if (inb) {
            if (N == 2) {
                const double v00 = P.utab[(int)pi[base] * M + k], v01 = P.utab[(int)pi[base + P.strd[1]] * M + k];
                const double v10 = P.utab[(int)pi[base + P.strd[0]] * M + k];
                const double v11 = P.utab[(int)pi[base + P.strd[0] + P.strd[1]] * M + k];
                const double a0 = 1.0 - y[0], a1 = 1.0 - y[1];
                val = v00 * a0 * a1 + v01 * a0 * y[1] + v10 * y[0] * a1 + v11 * y[0] * y[1];
            } else {
#pragma unroll
                for (int corner = 0; corner < (1 << N); ++corner) {
                    double w = 1.0;
                    long long off = base;
#pragma unroll
                    for (int d = 0; d < N; ++d) {
                        const int bit = (corner >> (N - 1 - d)) & 1;
                        w = w * (bit ? y[d] : (1.0 - y[d]));
                        off += bit ? P.strd[d] : 0;
                    }
                    val = val + P.utab[(int)pi[off] * M + k] * w;
                }
            }
        }
        u[k] = val;
    }
}

// mechanical closed forms: f = [dq; ddq] anywhere in the state space
template <int DYN>
struct RollMech {
    static constexpr int N = 2 * Dyn<DYN>::DOF, M = Dyn<DYN>::M;
    __device__ static void f(const DevP& P, const double*, const double* x, const double* u, double* dx) {
        constexpr int DOF = Dyn<DYN>::DOF;
        double tr[8], acc[DOF];
        Dyn<DYN>::trig_from_state(x, tr);
        Dyn<DYN> dyn;
        dyn.init(P.c, x, tr);
        dyn.accel(u, acc);
#pragma unroll
        for (int j = 0; j < DOF; ++j) {
            dx[j] = x[DOF + j];
            dx[DOF + j] = acc[j];
        }
    }
};
// the explicit systems, as functions of a CONTINUOUS state and input (the sweep kernels only need them at grid nodes and
// grid actions and read host tables there).  rp = pvi_set_rollout_params: per-system constants, see include/pyrovi.h.
template <int DYN>
struct RollExpl;
template <>
struct RollExpl<PVI_DYN_HELICOPTER> {  // drone.py:613-636
    static constexpr int N = 3, M = 1;
    __device__ static void f(const DevP& P, const double*, const double* x, const double* u, double* dx) {
        dx[0] = P.c[0] * u[0];
        dx[1] = x[0];
        dx[2] = P.c[1];
    }
};
template <>
struct RollExpl<PVI_DYN_KINCAR> {  // vehicle_steering.py:64-86; c = [1 / length]
    static constexpr int N = 3, M = 2;
    __device__ static void f(const DevP& P, const double*, const double* x, const double* u, double* dx) {
        dx[0] = u[0] * cos(x[2]);
        dx[1] = u[0] * sin(x[2]);
        dx[2] = u[0] * tan(u[1]) * P.c[0];
    }
};
template <>
struct RollExpl<PVI_DYN_QUARTERCAR> {  // suspension.py:73-124; rp = [terms, a[terms], w[terms], phi[terms]]
    static constexpr int N = 3, M = 1;
    __device__ static void f(const DevP& P, const double* rp, const double* x, const double* u, double* dx) {
        const int nt = (int)rp[0];
        double z = 0.0, dz = 0.0;
        for (int i = 0; i < nt; ++i) {
            const double a = rp[1 + i], w = rp[1 + nt + i], ph = rp[1 + 2 * nt + i];
            z = z + a * sin(w * (x[2] - ph));
            dz = dz + a * w * cos(w * (x[2] - ph));
        }
        dx[0] = P.c[0] * ((u[0] - P.c[1] * (x[1] - z)) - P.c[2] * (x[0] - dz));
        dx[1] = x[0];
        dx[2] = P.c[3];
    }
};
template <>
struct RollExpl<PVI_DYN_HOLONOMIC> {  // vehicle_steering.py:238-259
    static constexpr int N = 2, M = 2;
    __device__ static void f(const DevP&, const double*, const double*, const double* u, double* dx) {
        dx[0] = u[0];
        dx[1] = u[1];
    }
};
template <>
struct RollExpl<PVI_DYN_LONGCAR> {  // vehicle_propulsion.py:96-184; rp = [mu_max, mu_slope, rho cdA, m, g, ry, rr]
    static constexpr int N = 2, M = 1;
    __device__ static void f(const DevP&, const double* rp, const double* x, const double* u, double* dx) {
        const double mu = rp[0] * (2.0 / (1.0 + exp(-rp[1] * u[0])) - 1.0);
        const double v = x[1], m = rp[3], g = rp[4];
        const double fd = 0.5 * rp[2] * v * fabs(v);
        dx[0] = v;
        dx[1] = (mu * m * g * rp[6] - fd) / (m * (1.0 + mu * rp[5]));
    }
};

template <typename F, typename PI_T>
__global__ void k_rollout(DevP P, const PI_T* __restrict__ pi, const double* __restrict__ rp, long long B, const double* __restrict__ X0,
                          int npts, double dt, double* __restrict__ Xt, double* __restrict__ Ut, double* __restrict__ Xe) {
    constexpr int N = F::N, M = F::M;
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double x[N];
#pragma unroll
    for (int d = 0; d < N; ++d) x[d] = X0[b * N + d];
    for (int i = 0; i < npts; ++i) {
        double u[M];
        rollout_policy<N, M, PI_T>(P, pi, x, u);  // controller.py:328-355: u = ctl.c(x)
        if (Xt) {
#pragma unroll
            for (int d = 0; d < N; ++d) Xt[(b * npts + i) * N + d] = x[d];
        }
        if (Ut) {
#pragma unroll
            for (int k = 0; k < M; ++k) Ut[(b * npts + i) * M + k] = u[k];
        }
        if (i + 1 < npts) {  // simulation.py:298-324: x <- f(x, u) dt + x
            double dx[N];
            F::f(P, rp, x, u, dx);
#pragma unroll
            for (int d = 0; d < N; ++d) x[d] = dx[d] * dt + x[d];
        }
    }
    if (Xe) {
#pragma unroll
        for (int d = 0; d < N; ++d) Xe[b * N + d] = x[d];
    }
}

template <typename PI_T>
__global__ void k_pi_from_i64(const long long* __restrict__ src, PI_T* __restrict__ dst, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (PI_T)src[i];
}

// dtype conversions for upload / download
template <typename REAL>
__global__ void k_from_f64(const double* __restrict__ src, REAL* __restrict__ dst, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (REAL)src[i];
}
template <typename SRC>
__global__ void k_to_f64(const SRC* __restrict__ src, double* __restrict__ dst, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (double)src[i];
}
template <typename PI_T>
__global__ void k_pi_to_i64(const PI_T* __restrict__ src, long long* __restrict__ dst, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (long long)src[i];
}

// =================================================================================================
// host side (the handle, helpers and cross-unit entry points: host.h)
// =================================================================================================
thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static const char* const OVERRIDE_KEYS[] = {
    "LSPLIT",      // log2 lanes per node of the float32 sweeps (small grids)
    "NPT",         // 2-D lean sweep: nodes per thread, 1 or 2
    "TV0", "TV1", "TV_EXACT",  // lean sweep tile shape (rows x columns; TV_EXACT: do not even out the column split)
    "TUNE",        // 0: no timed candidate sweeps at create (first candidate that fits)
    "LDS_KB",      // LDS budget of the lean window
    "DMA16",       // 0: 4-byte window DMA
    "NO_RS64", "NO_TBTILE", "NO_XCD", "XCD64", "NO_SPLIT_FINISH", "SPLIT_FINISH",   // layout / launch details of the lean and float64 sweeps
    "NO_LEAN",     // float32: skip the LDS-window kernel (plain-gather k_sweep_fast)
    "NO_FAST",     // float32: float64 dynamics with float32 storage (k_sweep<float>)
    "NO_SWEEP64",  // float64: the operation-for-operation kernel k_sweep instead of k_sweep64
    "SPARSE", "PATCH",         // 4-D float64 / exact-float32 sweeps: walk over validity masks, 8x8 patch mapping
    "NO_PACK",     // table tier: sweep the raw tables instead of the packed records
    "SPLINE_CHUNK",            // spline mode: rows per chunk of the substitution passes
    "WIN",         // 0: no LDS-window kernel for 4-D float32 sweeps (plain-gather k_sweep_fast)
    "XCD_CHUNK",   // 4-D float64 sweep: rows of axis 0 per chunk dealt round-robin to the XCDs (0: one contiguous eighth per XCD)
    "L4PIN",       // 4-D lean sweep tiling "cap/threads/widest" (the `choice=` token of pvi_describe): no timed candidates
    "BANDS",       // 4-D lean sweep launch order: bands of the tile list per XCD pass (default: sized for the L2)
    "TABLES",      // lean sweep per-node coefficient tables: 0 per-node arrays, 1 factorised where the dynamics allow
    "DEFER",       // 0: the 2-D float32 sweep keeps its in-kernel ticket / k_sweep_finish per sweep instead of the deferred fold
    "XCD3",        // 0: float32 3-D sweep in plain block order instead of an eighth of axis 1 per XCD
    "JWIN",        // 0: the register-table sweeps gather J from memory instead of the workgroup's LDS window
    "REGTAB",      // 0: the multi-sweep launch of 2-D grids recomputes the per-action cells every sweep (fence-based barrier)
    "VMASK",       // 0: the 4-D float32 window sweep clamps and compares cell indices instead of reading set-up's validity bits
    "RS4",         // 4-D float32 window sweep: the row pitch in 8-byte slots (even, at least the longest window row; bank experiments)
    "RS_CONG",     // 1: 4-D float32 window sweep, row pitch congruent to the widest tile's (even) width modulo 32 (bank experiments)
    "MULTI32",     // 1: batches of the 2-D float32 window sweep as one cooperative launch (k_sweep_leanm; opt-in until measured)
    "FBCHECK",     // 1: PVI_FLAG_F32_FEEDBACK on a 4-D grid runs the epilogue with the corruption detector (k_sweep_lean4fbc; opt-in until it has run)
    "UNPROVEN",    // 1: admit kernels that have not yet passed their tests on hardware (error-feedback storage outside 4-D grids)
    "MULTI",       // 0: one launch per sweep also where a batch could run as ONE multi-sweep launch (k_sweep64m)
};
static std::vector<std::pair<std::string, std::string>> g_overrides;
static std::mutex g_override_mu;

// value of an override or NULL.  The string is a thread-local COPY taken under the lock (one slot per key, so several
// values can be held at once): a concurrent pvi_override cannot pull it from under the reader.
const char* ovr(const char* key) {
    static thread_local std::vector<std::pair<std::string, std::string>> held;
    std::lock_guard<std::mutex> lk(g_override_mu);
    for (auto& kv : g_overrides)
        if (kv.first == key) {
            for (auto& h : held)
                if (h.first == key) {
                    h.second = kv.second;
                    return h.second.c_str();
                }
            held.reserve(64);          // (fewer keys than that exist: no reallocation moves the strings of other keys)
            held.emplace_back(kv.first, kv.second);
            return held.back().second.c_str();
        }
    return nullptr;
}
extern "C" int pvi_abi_version(void) { return PVI_ABI_VERSION; }

extern "C" int pvi_override(const char* key, const char* value) {
    if (!key) {  // clear everything
        std::lock_guard<std::mutex> lk(g_override_mu);
        g_overrides.clear();
        return PVI_OK;
    }
    bool known = false;
    for (const char* k : OVERRIDE_KEYS) known = known || !strcmp(k, key);
    if (!known) return fail(PVI_EINVAL, "pvi_override: unknown key '%s'", key);
    std::lock_guard<std::mutex> lk(g_override_mu);
    for (size_t i = 0; i < g_overrides.size(); ++i)
        if (g_overrides[i].first == key) {
            if (value) g_overrides[i].second = value;
            else g_overrides.erase(g_overrides.begin() + (long)i);
            return PVI_OK;
        }
    if (value) g_overrides.emplace_back(key, value);
    return PVI_OK;
}
extern "C" const char* pvi_last_error(void) { return g_err; }

extern "C" int pvi_device_count(int* count) {
    if (!count) return fail(PVI_EINVAL, "count is NULL");
    HIPCHK(hipGetDeviceCount(count));
    return PVI_OK;
}

static inline bool is_node_dyn(int dyn) { return dyn >= PVI_DYN_NODE_1x1 && dyn <= PVI_DYN_NODE_2x2; }
static inline bool is_dyn3(int dyn) { return dyn >= PVI_DYN_HELICOPTER && dyn <= PVI_DYN_LONGCAR; }  // the explicit (non-mechanical) systems
static inline bool is_cost_in_kernel(int c) {
    return c == PVI_COST_QUADRATIC || c == PVI_COST_TIME || c == PVI_COST_QUADRATIC_DOMAIN || c == PVI_COST_REACHABILITY;
}

static int dyn_shape(int dyn, int* n, int* m) {
    switch (dyn) {
        case PVI_DYN_PENDULUM: *n = 2; *m = 1; return 0;
        case PVI_DYN_CARTPOLE: *n = 4; *m = 1; return 0;
        case PVI_DYN_CARTPOLE_SW: *n = 4; *m = 1; return 0;
        case PVI_DYN_TWOLINK: *n = 4; *m = 2; return 0;
        case PVI_DYN_NODE_1x1: *n = 2; *m = 1; return 0;
        case PVI_DYN_NODE_2x1: *n = 4; *m = 1; return 0;
        case PVI_DYN_NODE_2x2: *n = 4; *m = 2; return 0;
        case PVI_DYN_HELICOPTER: *n = 3; *m = 1; return 0;
        case PVI_DYN_KINCAR: *n = 3; *m = 2; return 0;
        case PVI_DYN_QUARTERCAR: *n = 3; *m = 1; return 0;
        case PVI_DYN_HOLONOMIC: *n = 2; *m = 2; return 0;
        case PVI_DYN_LONGCAR: *n = 2; *m = 1; return 0;
    }
    return -1;
}

extern "C" void pvi_destroy(pvi_handle h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (void* p : h->dev_allocs) (void)hipFree(p);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int pvi_create(const pvi_desc* d, pvi_handle* out) {
    if (!d || !out) return fail(PVI_EINVAL, "NULL argument");
    if (d->struct_size != sizeof(pvi_desc))
        return fail(PVI_EINVAL, "pvi_desc size mismatch: caller %u, library %zu", d->struct_size, sizeof(pvi_desc));
    // reference: NotImplementedError for n not in {2,3,4}, m not in {1,2} (discretizer.py:245, :306)
    if (d->n < 2 || d->n > PVI_MAX_N) return fail(PVI_EINVAL, "state dimension n=%d not in {2,3,4}", d->n);
    if (d->m < 1 || d->m > PVI_MAX_M) return fail(PVI_EINVAL, "input dimension m=%d not in {1,2}", d->m);
    if (d->dtype != PVI_F32 && d->dtype != PVI_F64) return fail(PVI_EINVAL, "bad dtype %d", d->dtype);
    if (d->dynamics_id != PVI_DYN_TABLE) {
        int n, m;
        if (dyn_shape(d->dynamics_id, &n, &m)) return fail(PVI_EINVAL, "unknown dynamics_id %d", d->dynamics_id);
        if (n != d->n || m != d->m)
            return fail(PVI_EINVAL, "dynamics %d needs n=%d m=%d, got n=%d m=%d", d->dynamics_id, n, m, d->n, d->m);
        if (!is_cost_in_kernel(d->cost_id))
            return fail(PVI_EINVAL, "in-kernel dynamics need cost_id QUADRATIC, TIME or QUADRATIC_DOMAIN");
        if (d->dynamics_id == PVI_DYN_CARTPOLE_SW && d->dtype != PVI_F32)
            return fail(PVI_EINVAL, "PVI_DYN_CARTPOLE_SW is a float32 order (float64 sums follow the reference's axis order)");
        if (is_dyn3(d->dynamics_id)) {
            if (d->n_obs < 0 || d->n_obs > PVI_MAX_OBS) return fail(PVI_EINVAL, "n_obs=%d not in [0,%d]", d->n_obs, PVI_MAX_OBS);
            for (int k = 0; k < 2; ++k)
                if (d->n_obs && (d->obs_axis[k] < 0 || d->obs_axis[k] >= d->n))
                    return fail(PVI_EINVAL, "obs_axis[%d]=%d is not a state axis", k, d->obs_axis[k]);
            if (d->dynamics_id == PVI_DYN_QUARTERCAR && (!d->trig[0] || !d->trig[1]))
                return fail(PVI_EINVAL, "PVI_DYN_QUARTERCAR needs the ground tables z, dz in trig[0], trig[1]");
            if (d->dynamics_id == PVI_DYN_LONGCAR && (!d->trig[0] || !d->act_aux))
                return fail(PVI_EINVAL, "PVI_DYN_LONGCAR needs the drag table in trig[0] and act_aux = [A][2]");
        }
    }
    long long plane = 1, A = 1;
    for (int i = 0; i < d->n; ++i) {
        if (d->x_dim[i] < 2) return fail(PVI_EINVAL, "x_dim[%d]=%d < 2", i, d->x_dim[i]);
        if (!d->x_level[i]) return fail(PVI_EINVAL, "x_level[%d] is NULL", i);
        if (i > 0) plane *= d->x_dim[i];
    }
    for (int k = 0; k < d->m; ++k) {
        if (d->u_dim[k] < 1) return fail(PVI_EINVAL, "u_dim[%d]=%d < 1", k, d->u_dim[k]);
        if (!d->u_level[k]) return fail(PVI_EINVAL, "u_level[%d] is NULL", k);
        A *= d->u_dim[k];
    }
    if (plane > 0x7fffffffLL) return fail(PVI_EINVAL, "plane too large");
    if (A > 65536) return fail(PVI_EINVAL, "more than 65536 actions");
    if (d->row_begin < 0 || d->row_end > d->x_dim[0] || d->row_begin >= d->row_end)
        return fail(PVI_EINVAL, "bad slab rows [%d,%d) of %d", d->row_begin, d->row_end, d->x_dim[0]);
    if (d->halo_lo < 0 || d->halo_hi < 0) return fail(PVI_EINVAL, "negative halo");

    pvi_problem* h = new (std::nothrow) pvi_problem();
    if (!h) return fail(PVI_ENOMEM, "host allocation failed");
    h->d = *d;
    h->device = d->device;
    h->plane = plane;
    h->A = (int)A;
    h->pi_size = A <= 256 ? 1 : 2;
    int rc = PVI_OK;
    auto bail = [&](int code) {
        pvi_destroy(h);
        return code;
    };
#define HCHK(expr)                 \
    do {                           \
        rc = [&]() -> int {        \
            HIPCHK(expr);          \
            return PVI_OK;         \
        }();                       \
        if (rc) return bail(rc);   \
    } while (0)

    HCHK(hipSetDevice(h->device));
    HCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HCHK(hipEventCreate(&h->ev0));
    HCHK(hipEventCreate(&h->ev1));

    DevP& P = h->P;
    memset(&P, 0, sizeof(P));
    P.n = d->n;
    P.m = d->m;
    P.A = (int)A;
    P.dof = d->n / 2;
    P.udim[0] = d->u_dim[0];
    P.udim[1] = d->m > 1 ? d->u_dim[1] : 1;
    P.plane = plane;
    P.row_begin = d->row_begin;
    P.row_end = d->row_end;
    P.store_begin = d->row_begin - d->halo_lo < 0 ? 0 : d->row_begin - d->halo_lo;
    P.store_end = d->row_end + d->halo_hi > d->x_dim[0] ? d->x_dim[0] : d->row_end + d->halo_hi;
    // the interpolation reads rows ci and ci+1: a slab must hold at least two rows
    if (P.store_end - P.store_begin < 2) return bail(fail(PVI_EINVAL, "stored slab has fewer than 2 rows"));
    h->stored = (long long)(P.store_end - P.store_begin) * plane;
    h->owned = (long long)(P.row_end - P.row_begin) * plane;
    long long s = 1;
    for (int i = d->n - 1; i >= 0; --i) {
        P.dim[i] = d->x_dim[i];
        P.strd[i] = s;
        s *= d->x_dim[i];
    }
    for (int i = 0; i < d->n; ++i) {
        if ((rc = dev_upload(h, d->x_level[i], (size_t)d->x_dim[i], &P.lev[i]))) return bail(rc);
        P.lb[i] = d->x_lb[i];
        P.ub[i] = d->x_ub[i];
        P.glo[i] = d->x_level[i][0];
        P.ghi[i] = d->x_level[i][d->x_dim[i] - 1];
        P.inv_step[i] = (double)(d->x_dim[i] - 1) / (P.ghi[i] - P.glo[i]);
        P.xbar[i] = d->xbar[i];
    }
    P.dt = d->dt;
    memcpy(P.c, d->dyn_params, sizeof(P.c));
    // row-major n x n -> dense n x n at the front of the 16-slot arrays
    memcpy(P.Q, d->Q, sizeof(double) * d->n * d->n);
    memcpy(P.S, d->S, sizeof(double) * d->n * d->n);
    if (d->cost_id == PVI_COST_TIME || d->cost_id == PVI_COST_REACHABILITY) {
        memset(P.Q, 0, sizeof(P.Q));
        memset(P.S, 0, sizeof(P.S));
    }
    memcpy(P.R, d->R, sizeof(double) * d->m * d->m);
    for (int k = 0; k < d->m; ++k) {
        P.ubar[k] = d->ubar[k];
        P.ulb[k] = d->u_lb[k];
        P.uub[k] = d->u_ub[k];
    }
    P.EPS = d->EPS;
    P.INF = d->INF;
    P.ontarget = d->ontarget_check;
    P.domain_check = d->cost_id == PVI_COST_QUADRATIC_DOMAIN || d->cost_id == PVI_COST_REACHABILITY;
    if (d->cost_id == PVI_COST_REACHABILITY) {  // g = 0 on valid states else INF (no on-target zeroing); h: see k_terminal_cost
        P.ontarget = 0;
        P.reach = 1;
    }
    P.hard_inf = (d->flags & PVI_FLAG_HARD_INF) != 0;
    if (is_dyn3(d->dynamics_id)) {
        P.nobs = d->n_obs;
        P.obs_ax[0] = d->n_obs ? d->obs_axis[0] : 0;
        P.obs_ax[1] = d->n_obs ? d->obs_axis[1] : 0;
        P.obs_half[0] = d->obs_half[0];
        P.obs_half[1] = d->obs_half[1];
        memcpy(P.obs, d->obs_box, sizeof(double) * 4 * (size_t)d->n_obs);
    }

    // action tables, C order over u_dim (discretizer.py:253-302)
    std::vector<double> utab((size_t)A * d->m), gu((size_t)A);
    std::vector<unsigned char> aok((size_t)A);
    for (long long a = 0; a < A; ++a) {
        long long r = a;
        double du[PVI_MAX_M];
        bool ok = true;
        for (int k = d->m - 1; k >= 0; --k) {
            const int ik = (int)(r % d->u_dim[k]);
            r /= d->u_dim[k];
            const double u = d->u_level[k][ik];
            utab[a * d->m + k] = u;
            du[k] = u - d->ubar[k];
            ok = ok && !(u < d->u_lb[k]) && !(u > d->u_ub[k]);  // system.py:208-215
        }
        // TimeCostFunction (costfunction.py:318-334): g = 1 outside the target ball -- the constant rides in the
        // per-action term, the state term and the terminal cost are zero (Q = S = 0 below)
        gu[a] = d->cost_id == PVI_COST_TIME ? 1.0 : (d->cost_id == PVI_COST_REACHABILITY ? 0.0 : quad_form_host(d->R, du, d->m));
        aok[a] = ok;
    }
    if ((rc = dev_upload(h, utab.data(), utab.size(), &P.utab))) return bail(rc);
    if ((rc = dev_upload(h, gu.data(), gu.size(), &P.gu))) return bail(rc);
    if ((rc = dev_upload(h, aok.data(), aok.size(), &P.aok))) return bail(rc);
    P.nearest = 0;
    P.all_aok = 1;
    for (long long a = 0; a < A; ++a) P.all_aok = P.all_aok && aok[a];
    {
        std::vector<int> aok32(aok.begin(), aok.end());
        if ((rc = dev_upload(h, aok32.data(), aok32.size(), &h->aok32))) return bail(rc);
    }
    {   // f32 fast path: per-action {u0, u1, gu*dt, isavalidinput}
        std::vector<float4> act((size_t)A);
        for (long long a = 0; a < A; ++a)
            act[a] = make_float4((float)utab[a * d->m], d->m > 1 ? (float)utab[a * d->m + 1] : 0.f,
                                 (float)(gu[a] * d->dt), aok[a] ? 1.f : 0.f);
        if ((rc = dev_upload(h, act.data(), act.size(), &h->F.act))) return bail(rc);
        {   // the same constants packed for scalar loads (sweep_lean.inc lean_act_group): groups of 4 actions
            const int per = d->m == 1 ? 2 : 4;
            std::vector<float> actc((size_t)(((A + 3) & ~3ll) + 4) * per, 0.f);
            for (long long a = 0; a < A; ++a) {
                actc[a * per] = act[a].x;
                if (per == 2) actc[a * per + 1] = act[a].z;
                else { actc[a * per + 1] = act[a].y; actc[a * per + 2] = act[a].z; }
            }
            if ((rc = dev_upload(h, actc.data(), actc.size(), &h->LP.actc))) return bail(rc);
        }
        h->F.guard = 1e-3f;
        long long threads = h->owned;
        int ls = 0;
        while (threads < (1ll << 20) && (2 << ls) <= 64 && (2 << ls) <= A) {
            ++ls;
            threads <<= 1;
        }
        if (const char* e = ovr("LSPLIT")) ls = atoi(e);
        h->F.lsplit = ls;
        bool box_is_grid = true;
        for (int i = 0; i < d->n; ++i)
            box_is_grid = box_is_grid && d->x_lb[i] == P.glo[i] && d->x_ub[i] == P.ghi[i];
        if (d->dtype == PVI_F64 && d->dynamics_id != PVI_DYN_TABLE && !is_dyn3(d->dynamics_id) && box_is_grid &&
            !ovr("NO_SWEEP64")) {
            // float64 second form: {u0, u1, gu, isavalidinput} per action, {level, RN(1 / (next level - level))} per level
            std::vector<Act64> a64((size_t)A);
            for (long long a = 0; a < A; ++a)
                a64[a] = Act64{utab[a * d->m], d->m > 1 ? utab[a * d->m + 1] : 0.0, gu[a], aok[a] ? 1.0 : 0.0};
            size_t nlev = 0;
            for (int i = 0; i < d->n; ++i) nlev += (size_t)d->x_dim[i];
            std::vector<double2> lr(nlev);
            size_t at = 0;
            for (int i = 0; i < d->n; ++i)
                for (int k = 0; k < d->x_dim[i]; ++k, ++at) {
                    const double l0 = d->x_level[i][k];
                    const double dd = k + 1 < d->x_dim[i] ? d->x_level[i][k + 1] - l0 : 1.0;
                    lr[at] = make_double2(l0, 1.0 / dd);  // IEEE division: the correctly rounded reciprocal
                }
            size_t nvel = 0;  // the velocity axes' tables are the ones kept in LDS
            for (int i = d->n / 2; i < d->n; ++i) nvel += (size_t)d->x_dim[i];
            h->levr_bytes = nvel * sizeof(double2);
            if (h->levr_bytes <= 48 * 1024) {
                if ((rc = dev_upload(h, a64.data(), a64.size(), &h->act64))) return bail(rc);
                if ((rc = dev_upload(h, lr.data(), lr.size(), &h->levr))) return bail(rc);
                h->use64 = true;
            }
        }
        h->fast_ok = d->dtype == PVI_F32 && d->dynamics_id != PVI_DYN_TABLE && !is_dyn3(d->dynamics_id) && box_is_grid &&
                     h->stored < 0x7fffffffLL && !ovr("NO_FAST");
    }

    // trig tables over the angle levels: supplied by the host (numpy) or computed here with libm
    auto table = [&](int slot, int axis, double (*fn)(double)) -> int {
        std::vector<double> t((size_t)d->x_dim[axis]);
        if (d->trig[slot])
            memcpy(t.data(), d->trig[slot], t.size() * sizeof(double));
        else
            for (size_t i = 0; i < t.size(); ++i) t[i] = fn(d->x_level[axis][i]);
        return dev_upload(h, t.data(), t.size(), &P.trig[slot]);
    };
    double (*fsin)(double) = [](double v) { return std::sin(v); };
    double (*fcos)(double) = [](double v) { return std::cos(v); };
    if (d->dynamics_id == PVI_DYN_PENDULUM) {
        if ((rc = table(0, 0, fsin))) return bail(rc);
    } else if (d->dynamics_id == PVI_DYN_CARTPOLE) {
        if ((rc = table(0, 1, fcos)) || (rc = table(1, 1, fsin))) return bail(rc);
    } else if (d->dynamics_id == PVI_DYN_CARTPOLE_SW) {   // (the angle is axis 0: core.h DynCartPole<true>)
        if ((rc = table(0, 0, fcos)) || (rc = table(1, 0, fsin))) return bail(rc);
    } else if (d->dynamics_id == PVI_DYN_TWOLINK) {
        if ((rc = table(0, 0, fsin)) || (rc = table(1, 1, fcos)) || (rc = table(2, 1, fsin))) return bail(rc);
        std::vector<double> t((size_t)d->x_dim[0] * d->x_dim[1]);
        if (d->trig[3])
            memcpy(t.data(), d->trig[3], t.size() * sizeof(double));
        else
            for (int i = 0; i < d->x_dim[0]; ++i)
                for (int j = 0; j < d->x_dim[1]; ++j)
                    t[(size_t)i * d->x_dim[1] + j] = std::sin(d->x_level[0][i] + d->x_level[1][j]);
        if ((rc = dev_upload(h, t.data(), t.size(), &P.trig[3]))) return bail(rc);
    } else if (d->dynamics_id == PVI_DYN_KINCAR) {
        if ((rc = table(0, 2, fcos)) || (rc = table(1, 2, fsin))) return bail(rc);
        std::vector<double> aux((size_t)A);
        for (long long a = 0; a < A; ++a)  // u0 * tan(u1) * (1/length), vehicle_steering.py:84
            aux[a] = d->act_aux ? d->act_aux[a] : utab[a * 2] * std::tan(utab[a * 2 + 1]) * d->dyn_params[0];
        if ((rc = dev_upload(h, aux.data(), aux.size(), &P.aux))) return bail(rc);
    } else if (d->dynamics_id == PVI_DYN_QUARTERCAR) {
        if ((rc = table(0, 2, fsin)) || (rc = table(1, 2, fcos))) return bail(rc);  // (both supplied: checked above)
    } else if (d->dynamics_id == PVI_DYN_LONGCAR) {
        if ((rc = table(0, 1, fsin))) return bail(rc);  // drag force over the velocity levels (supplied: checked above)
        if ((rc = dev_upload(h, d->act_aux, (size_t)A * 2, &P.aux))) return bail(rc);
    } else if (is_node_dyn(d->dynamics_id)) {
        if (!d->trig[0] || !d->trig[1]) return bail(fail(PVI_EINVAL, "PVI_DYN_NODE_* needs the a0 / Bn tables in trig[0], trig[1]"));
        const int dof = d->n / 2;
        size_t nodes = 1, pos = 1;
        for (int i = 0; i < d->n; ++i) nodes *= (size_t)d->x_dim[i];
        for (int i = 0; i < dof; ++i) pos *= (size_t)d->x_dim[i];
        if ((rc = dev_upload(h, d->trig[0], nodes * dof, &P.trig[0]))) return bail(rc);
        if ((rc = dev_upload(h, d->trig[1], pos * dof * d->m, &P.trig[1]))) return bail(rc);
    }

    const size_t esz = d->dtype == PVI_F64 ? 8 : 4;
    for (int b = 0; b < 2; ++b) {
        if (d->ext_J[0] && d->ext_J[1]) {
            h->J[b] = d->ext_J[b];
            h->own_J = false;
        } else {
            HCHK(hipMalloc(&h->J[b], (size_t)h->stored * esz + 64));  // + slack: 16-byte window loads may run past a row
            h->dev_allocs.push_back(h->J[b]);
            HCHK(hipMemsetAsync(h->J[b], 0, (size_t)h->stored * esz, h->stream));
        }
    }
    if (d->ext_pi) {
        h->pi = d->ext_pi;
        h->own_pi = false;
    } else {
        HCHK(hipMalloc(&h->pi, (size_t)h->owned * h->pi_size));
        h->dev_allocs.push_back(h->pi);
        HCHK(hipMemsetAsync(h->pi, 0, (size_t)h->owned * h->pi_size, h->stream));
    }
    void* p = nullptr;
    HCHK(hipMalloc(&p, sizeof(Ctrl)));
    h->dev_allocs.push_back(p);
    h->ctrl = (Ctrl*)p;
    HCHK(hipMemsetAsync(h->ctrl, 0, sizeof(Ctrl), h->stream));
    HCHK(hipMalloc(&p, sizeof(unsigned long long) * STAT_WORDS * MAX_BATCH));
    h->dev_allocs.push_back(p);
    h->slots = (unsigned long long*)p;
    HCHK(hipMalloc(&p, sizeof(double) * 4 * MAX_BATCH));
    h->dev_allocs.push_back(p);
    h->results = (double*)p;
    HCHK(hipStreamSynchronize(h->stream));
#undef HCHK
    if ((rc = lean_setup(h))) return bail(rc);
    if (d->dynamics_id == PVI_DYN_CARTPOLE_SW && !h->lean4_ok)   // (its only production kernel: no other family is instantiated for it)
        return bail(fail(PVI_EINVAL, "PVI_DYN_CARTPOLE_SW needs the float32 window sweep of 4-D grids; this handle does not take it (%s)",
                         h->lean_why[0] ? h->lean_why : "no window set-up"));
    if (d->flags & PVI_FLAG_F32_FEEDBACK) {
        // error-feedback storage: one residual per owned node, private to the node (sweep_lean4.inc lean4_feedback for the 4-D
        // window sweep; sweep_lean.inc lean_feedback for the 2-D one: one-input systems, one node per thread)
        const bool lean2_fb = h->lean_ok && !h->lean4_ok && d->n == 2 && d->m == 1 && h->LP.npt == 1 &&
                              (d->dynamics_id == PVI_DYN_PENDULUM || d->dynamics_id == PVI_DYN_NODE_1x1);
        const bool fast3_fb = d->dtype == PVI_F32 && is_dyn3(d->dynamics_id) && A <= 64 && !ovr("NO_FAST");   // (k_sweep3_fast, set up below)
        // The 2-D and explicit-system forms were written while no MI355X was reachable (round 5) and their kernels are not yet
        // verified code objects (profiles/verified_kernels.json): they take pvi_override("UNPROVEN", "1") until the tests of
        // tests/test_gpu_zz_unproven.py have passed on hardware.  The 4-D form (k_sweep_lean4fb) has.
        if (!h->lean4_ok && (lean2_fb || fast3_fb) && !ovr_is("UNPROVEN", 1))
            return bail(fail(PVI_EINVAL, "PVI_FLAG_F32_FEEDBACK on a %s: its kernel has not yet run its tests on hardware; pvi_override(\"UNPROVEN\", \"1\") admits it",
                             lean2_fb ? "2-D grid" : "explicit system"));
        if (!h->lean4_ok && !lean2_fb && !fast3_fb)
            return bail(fail(PVI_EINVAL, "PVI_FLAG_F32_FEEDBACK needs a float32 production sweep (4-D grids, 2-D grids with one input, or an explicit system with at most 64 actions); this handle does not take one (%s)",
                             d->dtype != PVI_F32 ? "dtype is not float32" : h->lean_why[0] ? h->lean_why : "no window set-up"));
        float* lo = nullptr;
        if ((rc = dev_alloc(h, (size_t)h->owned, &lo))) return bail(rc);
        rc = [&]() -> int {
            HIPCHK(hipMemsetAsync(lo, 0, (size_t)h->owned * sizeof(float), h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            return PVI_OK;
        }();
        if (rc) return bail(rc);
        jlo_of(h) = lo;
        h->fbcheck = h->lean4_ok && ovr_is("FBCHECK", 1);
    }
    if (d->dtype == PVI_F32 && is_dyn3(d->dynamics_id) && A <= 64 && !ovr("NO_FAST")) {
        // fast3: the validity of every cell of an explicit system, once (sweep_lean.inc's idea applied to the obstacle tests)
        unsigned long long* m = nullptr;
        if ((rc = dev_alloc(h, (size_t)h->owned, &m))) return bail(rc);
        const unsigned gm = grid_for(h->owned);
        switch (d->dynamics_id) {
            case PVI_DYN_HELICOPTER: hipLaunchKernelGGL((k_mask3<PVI_DYN_HELICOPTER>), gm, 256, 0, h->stream, h->P, m); break;
            case PVI_DYN_KINCAR: hipLaunchKernelGGL((k_mask3<PVI_DYN_KINCAR>), gm, 256, 0, h->stream, h->P, m); break;
            case PVI_DYN_QUARTERCAR: hipLaunchKernelGGL((k_mask3<PVI_DYN_QUARTERCAR>), gm, 256, 0, h->stream, h->P, m); break;
            case PVI_DYN_HOLONOMIC: hipLaunchKernelGGL((k_mask3<PVI_DYN_HOLONOMIC>), gm, 256, 0, h->stream, h->P, m); break;
            default: hipLaunchKernelGGL((k_mask3<PVI_DYN_LONGCAR>), gm, 256, 0, h->stream, h->P, m); break;
        }
        rc = [&]() -> int {
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(h->stream));
            return PVI_OK;
        }();
        if (rc) return bail(rc);
        h->okmask3 = m;
    }
    // the float32-storage exact path (float64 dynamics: systems whose float32 displacement cancels) walks the same masks
    bool grid_is_box = true;
    for (int i = 0; i < d->n; ++i) grid_is_box = grid_is_box && d->x_lb[i] == P.glo[i] && d->x_ub[i] == P.ghi[i];
    const bool exact32 = d->dtype == PVI_F32 && d->dynamics_id != PVI_DYN_TABLE && !is_dyn3(d->dynamics_id) &&
                         !is_node_dyn(d->dynamics_id) && !h->lean_ok && !h->fast_ok && grid_is_box;
    if (exact32 && d->n == 4 && A <= 128 && !h->act64) {
        std::vector<Act64> a64((size_t)A);
        for (long long a = 0; a < A; ++a)
            a64[a] = Act64{utab[a * d->m], d->m > 1 ? utab[a * d->m + 1] : 0.0, gu[a], aok[a] ? 1.0 : 0.0};
        if ((rc = dev_upload(h, a64.data(), a64.size(), &h->act64))) return bail(rc);
    }
    if (((h->use64 && h->levr_bytes + (size_t)A * sizeof(Act64) <= 48 * 1024) || exact32) && d->n == 4 && A <= 128 &&
        !(ovr("SPARSE") && !atoi(ovr("SPARSE")))) {
        // SPARSE float64 sweep: validity of every (node, action) cell, once (it does not change between sweeps); kept
        // where fewer than half of the cells land in the box (PVI_SPARSE=1 keeps it regardless, =0 never builds it)
        rc = [&]() -> int {
            uint4* vm = nullptr;
            unsigned long long* cnt = nullptr;
            int r;
            if ((r = dev_alloc(h, (size_t)h->owned, &vm))) return r;
            if ((r = dev_alloc(h, 1, &cnt))) return r;
            HIPCHK(hipMemsetAsync(cnt, 0, sizeof(*cnt), h->stream));
            if ((r = launch_valid_mask(h, vm, cnt))) return r;  // (f64.hip)
            HIPCHK(hipGetLastError());
            unsigned long long inside = 0;
            HIPCHK(hipMemcpyAsync(&inside, cnt, sizeof(inside), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            dev_release(h, cnt);
            h->infrac64 = (double)inside / ((double)h->owned * (double)A);
            const bool forced = ovr("SPARSE") && atoi(ovr("SPARSE"));
            if (h->infrac64 < 0.5 || forced) {
                h->vmask = vm;
                h->sparse64 = 1;
            } else {
                dev_release(h, vm);
            }
            return PVI_OK;
        }();
        if (rc) return bail(rc);
    }
    if (h->use64 && d->n == 4) {
        // Wave mapping of the float64 sweep on 4-D grids, timed like the float32 tile shapes: patches win where few
        // cells land in the box (two-link 101^4 x 121: 33.0 -> 22.5 ms), lines where most do and the velocity plane does
        // not divide by 8 (cart-pole 51^4: 0.77 against 0.88 ms).  One warm-up and one timed sweep per mapping; the
        // results do not depend on it.  PVI_PATCH=0 / 1 pins it.
        // The SPARSE walk (validity masks) is timed the same way where the masks were built: it wins where the loop is
        // bound by instruction issue (two-link 41^4: 0.89 -> 0.48 ms) and loses where the gathers of the in-box cells
        // wait for HBM anyway (two-link 101^4: 22.7 -> 24.4 ms).  PVI_SPARSE=1 pins it on.
        const bool sparse_forced = ovr("SPARSE") && atoi(ovr("SPARSE"));
        if (ovr("PATCH") && (sparse_forced || !h->sparse64)) {
            h->patch64 = atoi(ovr("PATCH")) ? 1 : 0;
        } else {
            float best_ms = 1e30f;
            int best = 1, best_sp = h->sparse64;
            const bool have_mask = h->sparse64 != 0;
            for (int cand = 0; cand < 4; ++cand) {
                const int pm = cand & 1, sp = cand >> 1;
                if (sp && !have_mask) continue;
                if (sparse_forced && sp != best_sp) continue;
                if (ovr("PATCH") && pm != (atoi(ovr("PATCH")) ? 1 : 0)) continue;
                h->patch64 = pm;
                h->sparse64 = sp;
                float ms = 0.f;
                rc = [&]() -> int {
                    for (int rep = 0; rep < 2; ++rep) {
                        if (rep == 1) HIPCHK(hipEventRecord(h->ev0, h->stream));
                        hipLaunchKernelGGL(k_reset_stats, 1, STAT_WORDS, 0, h->stream, h->slots, STAT_WORDS);
                        hipLaunchKernelGGL(k_begin_batch, 1, 1, 0, h->stream, h->ctrl);
                        int r = launch_sweep(h, h->cur, 1.0, h->stream, 0, -1.0);
                        if (r) return r;
                    }
                    HIPCHK(hipEventRecord(h->ev1, h->stream));
                    HIPCHK(hipStreamSynchronize(h->stream));
                    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
                    return PVI_OK;
                }();
                if (rc) return bail(rc);
                if (ms < best_ms) {
                    best_ms = ms;
                    best = pm;
                    best_sp = sp;
                }
            }
            h->sparse64 = best_sp;
            if (have_mask && !h->sparse64) {  // the dense walk stays: the masks are not needed
                dev_release(h, (void*)h->vmask);
                h->vmask = nullptr;
            }
            h->patch64 = best;
            rc = [&]() -> int {  // the timed sweeps wrote into the second J buffer, pi and the control block
                HIPCHK(hipMemsetAsync(h->ctrl, 0, sizeof(Ctrl), h->stream));
                HIPCHK(hipMemsetAsync(h->pi, 0, (size_t)h->owned * h->pi_size, h->stream));
                HIPCHK(hipMemsetAsync(h->J[h->cur ^ 1], 0, (size_t)h->stored * 8, h->stream));
                HIPCHK(hipStreamSynchronize(h->stream));
                return PVI_OK;
            }();
            if (rc) return bail(rc);
        }
    }
    *out = h;
    return PVI_OK;
}

extern "C" int64_t pvi_plane_size(pvi_handle h) { return h ? h->plane : 0; }
extern "C" int64_t pvi_stored_nodes(pvi_handle h) { return h ? h->stored : 0; }
extern "C" int64_t pvi_owned_nodes(pvi_handle h) { return h ? h->owned : 0; }
extern "C" int pvi_pi_itemsize(pvi_handle h) { return h ? h->pi_size : 0; }

static int describe_impl(pvi_handle h, char* buf, int32_t n);
extern "C" int pvi_describe(pvi_handle h, char* buf, int32_t n) {
    if (!h || !buf || n <= 0) return fail(PVI_EINVAL, "bad argument");
    std::vector<char> tmp((size_t)n + 256);
    int rc = describe_impl(h, tmp.data(), (int32_t)tmp.size());
    if (rc) return rc;
    // `kernel=`: the sweep kernel of the last launch as a kernel trace names it, spaces removed ("-" before the first sweep);
    // the second token, right behind `path=`
    std::string d(tmp.data());
    const std::string k = std::string(" kernel=") + (h->kname[0] ? h->kname : "-");
    const size_t at = d.find(' ');
    if (at == std::string::npos) d += k; else d.insert(at, k);
    snprintf(buf, (size_t)n, "%s", d.c_str());
    return PVI_OK;
}
static int describe_impl(pvi_handle h, char* buf, int32_t n) {
    if (h->spline) {
        snprintf(buf, (size_t)n, "path=spline-%s chunk0=%d warm0=%d chunk1=%d warm1=%d",
                 h->d.dynamics_id == PVI_DYN_TABLE ? "table" : "fused", h->SP.chunk0, h->SP.warm0, h->SP.chunk1, h->SP.warm1);
        return PVI_OK;
    }
    const char* path = h->d.dynamics_id == PVI_DYN_TABLE ? (h->packed ? "table-packed" : "table")
                       : (h->d.dtype == PVI_F64 && h->use64) ? "exact-f64v2"
                       : h->d.dtype == PVI_F64 ? "exact-f64"
                       : (h->lean_ok || h->lean4_ok) ? "lean"
                       : (h->fast_ok && !is_node_dyn(h->d.dynamics_id)) ? "fast"
                       : h->okmask3 ? "fast3"
                       : (h->d.dynamics_id == PVI_DYN_TABLE ? (h->packed ? "table-packed" : "table") : "exact-f32");
    if (h->d.dtype == PVI_F64 && h->use64 && h->d.dynamics_id != PVI_DYN_TABLE) {
        snprintf(buf, (size_t)n, "path=exact-f64v2 mapping=%s off32=%d sparse=%d inbox=%.4f multi=%d regtab=%d note=%s",
                 h->P.n == 4 ? (h->patch64 ? "patch8x8" : "line64") : "line64",
                 (int)((unsigned long long)h->stored * 8ull < (1ull << 32)), h->sparse64, h->infrac64, h->multi64, h->regtab64, h->multi_why);
        return PVI_OK;
    }
    if (h->lean4_ok) {
        // win=1: position-paired window + ds_read_b64 (sweep_lean4.inc); tables: bit d set = the displacement table does
        // not span axis d; rowpieces / bands: the step-aligned row pieces of axis 2 and their grouping in the launch order
        snprintf(buf, (size_t)n, "path=lean tile=%dx%d grid=%ux1x1 block=%d pw1=%d lds_bytes=%zu lsplit=0 tb_tile=1 dma16=0 npt=1 "
                 "reach=0 opmag=0 sparse=0 win=1 tables=%d ptab=%d gx=%s stage=%d feedback=%d vmask=%d choice=%s tiles_per_plane=%d bands=%d cands=%s note=%s", h->L4.TV0, h->L4.TV1,
                 h->lean4_grid, h->lean4_block, h->L4.RS, h->lean4_lds, h->lean4_tables, h->lean4_ptab_inv, h->L4.gx ? "node" : "axes", h->lean4_stage,
                 h->jlo ? 1 : 0, h->L4.vmask ? 1 : 0, h->lean4_choice, h->L4.ntr, h->lean4_bands, h->lean4_cands[0] ? h->lean4_cands : "-", h->lean_why);
        return PVI_OK;
    }
    snprintf(buf, (size_t)n, "path=%s tile=%dx%d grid=%ux%ux%u block=%d pw1=%d lds_bytes=%zu lsplit=%d tb_tile=%d dma16=%d npt=%d reach=%d opmag=%d sparse=%d win=0 tables=0 feedback=%d multi=%d note=%s",
             path, h->LP.TV0, h->LP.TV1, h->lean_grid.x, h->lean_grid.y, h->lean_grid.z, h->lean_block, h->lean_pw1,
             h->lean_lds, h->lean_ok ? h->LP.lsplit : h->F.lsplit, h->LP.tb_tile, h->lean_ok ? h->LP.dma16 : 0,
             h->lean_ok ? h->LP.npt : 1, h->lean_reach, h->lean_opmag,
             (h->d.dtype == PVI_F32 && h->sparse64 && h->vmask) ? 1 : 0, h->jlo ? 1 : 0, h->multi32 == 1 ? 1 : 0, h->lean_why);
    return PVI_OK;
}

extern "C" int pvi_synchronize(pvi_handle h) {
    if (!h) return fail(PVI_EINVAL, "NULL handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    return PVI_OK;
}

// ---- terminal cost ------------------------------------------------------------------------------
template <typename REAL>
static int terminal_cost_t(pvi_problem* h) {
    const unsigned g = grid_for(h->stored);
    REAL* J = (REAL*)h->J[h->cur];
    switch (h->P.n) {
        case 2: hipLaunchKernelGGL((k_terminal_cost<REAL, 2>), g, 256, 0, h->stream, h->P, J); break;
        case 3: hipLaunchKernelGGL((k_terminal_cost<REAL, 3>), g, 256, 0, h->stream, h->P, J); break;
        default: hipLaunchKernelGGL((k_terminal_cost<REAL, 4>), g, 256, 0, h->stream, h->P, J); break;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(h->pi, 0, (size_t)h->owned * h->pi_size, h->stream));
    if (jlo_of(h)) HIPCHK(hipMemsetAsync(jlo_of(h), 0, (size_t)h->owned * sizeof(float), h->stream));  // a new J: no residuals
    HIPCHK(hipStreamSynchronize(h->stream));
    return PVI_OK;
}

extern "C" int pvi_terminal_cost(pvi_handle h) {
    if (!h) return fail(PVI_EINVAL, "NULL handle");
    if (!is_cost_in_kernel(h->d.cost_id))
        return fail(PVI_ESTATE, "terminal cost needs an in-kernel cost (QUADRATIC, TIME or QUADRATIC_DOMAIN)");
    HIPCHK(hipSetDevice(h->device));
    return h->d.dtype == PVI_F64 ? terminal_cost_t<double>(h) : terminal_cost_t<float>(h);
}

// ---- upload / download ----------------------------------------------------------------------------
static int ensure_stage(pvi_problem* h, long long n) {
    if (h->stage_n >= n) return PVI_OK;
    if (h->stage) {
        HIPCHK(hipFree(h->stage));
        for (auto& p : h->dev_allocs)
            if (p == h->stage) p = nullptr;
    }
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, (size_t)n * 8));
    h->dev_allocs.push_back(p);
    h->stage = (double*)p;
    h->stage_n = n;
    return PVI_OK;
}

static int rows_check(pvi_problem* h, int row0, int nrows, bool owned_only) {
    const int lo = owned_only ? h->P.row_begin : h->P.store_begin, hi = owned_only ? h->P.row_end : h->P.store_end;
    if (nrows <= 0 || row0 < lo || row0 + nrows > hi)
        return fail(PVI_EINVAL, "rows [%d,%d) outside [%d,%d)", row0, row0 + nrows, lo, hi);
    return PVI_OK;
}

static const long long STAGE_CHUNK = 1ll << 24;  // elements per staged transfer (128 MiB of f64)

extern "C" int pvi_set_J(pvi_handle h, const double* Jr, int32_t row0, int32_t nrows) {
    if (!h || !Jr) return fail(PVI_EINVAL, "NULL argument");
    int rc = rows_check(h, row0, nrows, false);
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    const long long n = (long long)nrows * h->plane, off = (long long)(row0 - h->P.store_begin) * h->plane;
    if (jlo_of(h)) HIPCHK(hipMemsetAsync(jlo_of(h), 0, (size_t)h->owned * sizeof(float), h->stream));  // a new J: no residuals
    if (h->d.dtype == PVI_F64) {
        HIPCHK(hipMemcpyAsync((double*)h->J[h->cur] + off, Jr, (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
    } else {
        for (long long s = 0; s < n; s += STAGE_CHUNK) {
            const long long c = n - s < STAGE_CHUNK ? n - s : STAGE_CHUNK;
            if ((rc = ensure_stage(h, c))) return rc;
            HIPCHK(hipMemcpyAsync(h->stage, Jr + s, (size_t)c * 8, hipMemcpyHostToDevice, h->stream));
            hipLaunchKernelGGL((k_from_f64<float>), grid_for(c), 256, 0, h->stream, h->stage,
                               (float*)h->J[h->cur] + off + s, c);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(h->stream));
        }
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return PVI_OK;
}

static int get_J_buf(pvi_problem* h, int which, double* Jr, int row0, int nrows) {
    if (!h || !Jr) return fail(PVI_EINVAL, "NULL argument");
    int rc = rows_check(h, row0, nrows, false);
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    const long long n = (long long)nrows * h->plane, off = (long long)(row0 - h->P.store_begin) * h->plane;
    const void* buf = h->J[which ? h->cur ^ 1 : h->cur];
    if (h->d.dtype == PVI_F64) {
        HIPCHK(hipMemcpyAsync(Jr, (const double*)buf + off, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
    } else {
        for (long long s = 0; s < n; s += STAGE_CHUNK) {
            const long long c = n - s < STAGE_CHUNK ? n - s : STAGE_CHUNK;
            if ((rc = ensure_stage(h, c))) return rc;
            hipLaunchKernelGGL((k_to_f64<float>), grid_for(c), 256, 0, h->stream, (const float*)buf + off + s,
                               h->stage, c);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(Jr + s, h->stage, (size_t)c * 8, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
        }
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return PVI_OK;
}

extern "C" int pvi_get_J(pvi_handle h, double* Jr, int32_t row0, int32_t nrows) { return get_J_buf(h, 0, Jr, row0, nrows); }
extern "C" int pvi_get_J_prev(pvi_handle h, double* Jr, int32_t row0, int32_t nrows) {
    return get_J_buf(h, 1, Jr, row0, nrows);
}

extern "C" int pvi_get_pi(pvi_handle h, int64_t* pr, int32_t row0, int32_t nrows) {
    if (!h || !pr) return fail(PVI_EINVAL, "NULL argument");
    int rc = rows_check(h, row0, nrows, true);
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    const long long n = (long long)nrows * h->plane, off = (long long)(row0 - h->P.row_begin) * h->plane;
    for (long long s = 0; s < n; s += STAGE_CHUNK) {
        const long long c = n - s < STAGE_CHUNK ? n - s : STAGE_CHUNK;
        if ((rc = ensure_stage(h, c))) return rc;
        if (h->pi_size == 1)
            hipLaunchKernelGGL((k_pi_to_i64<unsigned char>), grid_for(c), 256, 0, h->stream,
                               (const unsigned char*)h->pi + off + s, (long long*)h->stage, c);
        else
            hipLaunchKernelGGL((k_pi_to_i64<unsigned short>), grid_for(c), 256, 0, h->stream,
                               (const unsigned short*)h->pi + off + s, (long long*)h->stage, c);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(pr + s, h->stage, (size_t)c * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return PVI_OK;
}

extern "C" int pvi_device_J(pvi_handle h, int which, void** p) {
    if (!h || !p) return fail(PVI_EINVAL, "NULL argument");
    *p = h->J[which ? h->cur ^ 1 : h->cur];
    return PVI_OK;
}
extern "C" int pvi_device_pi(pvi_handle h, void** p) {
    if (!h || !p) return fail(PVI_EINVAL, "NULL argument");
    *p = h->pi;
    return PVI_OK;
}

// spline refit of the cost-to-go J: four substitution passes (forward/backward x two axes)
template <typename REAL>
static void spline_fit_launch(pvi_problem* h, const REAL* J, hipStream_t st) {
    const SplineP& S = h->SP;
    const dim3 g0(grid_for(S.n1, 64), (S.n0 + S.chunk0 - 1) / S.chunk0), g1(grid_for(S.n0, 64), (S.n1 + S.chunk1 - 1) / S.chunk1);
    hipLaunchKernelGGL((k_spline_axis0<REAL, false>), g0, 64, 0, st, S, J, S.work);
    hipLaunchKernelGGL((k_spline_axis0<double, true>), g0, 64, 0, st, S, (const double*)S.work, S.coef);
    hipLaunchKernelGGL((k_spline_axis1<false>), g1, 64, 0, st, S, (const double*)S.coef, S.work);
    hipLaunchKernelGGL((k_spline_axis1<true>), g1, 64, 0, st, S, (const double*)S.work, S.coef);
}

// ---- sweep launch -----------------------------------------------------------------------------------
template <typename REAL, typename PI_T>
static int launch_sweep_t(pvi_problem* h, int src, double alpha, hipStream_t st, SweepCtl sc) {
    const unsigned g = grid_for(h->owned);
    sc.nblocks = g;
    const REAL* Jin = (const REAL*)h->J[src];
    REAL* Jout = (REAL*)h->J[src ^ 1];
    PI_T* pi = (PI_T*)h->pi;
    if (h->spline) {
        const SplineP& S = h->SP;
        spline_fit_launch<REAL>(h, Jin, st);
        set_kname(h, "k_sweep_spline", h->d.dynamics_id == PVI_DYN_TABLE ? (int)PVI_DYN_TABLE : (int)PVI_DYN_PENDULUM, tname<REAL>(), tname<PI_T>());
        if (h->d.dynamics_id == PVI_DYN_TABLE) {
            if (!h->d_xnext || !h->d_G) return fail(PVI_ESTATE, "spline sweep without the raw tables: pvi_set_tables after pvi_set_interpolation");
            hipLaunchKernelGGL((k_sweep_spline<PVI_DYN_TABLE, REAL, PI_T>), g, 256, 0, st, h->P, S, h->d_xnext, h->d_G,
                               h->d_ok, Jin, Jout, pi, alpha, sc, h->P.utab, h->P.gu, h->aok32);
        } else {
            hipLaunchKernelGGL((k_sweep_spline<PVI_DYN_PENDULUM, REAL, PI_T>), g, 256, 0, st, h->P, S,
                               (const double*)nullptr, (const double*)nullptr, (const unsigned char*)nullptr, Jin, Jout,
                               pi, alpha, sc, h->P.utab, h->P.gu, h->aok32);
        }
        HIPCHK(hipGetLastError());
        return PVI_OK;
    }
    if constexpr (sizeof(REAL) == 4) {  // the float32 production families (lean.hip)
        h->L4.jlo = h->lean4_ok ? h->jlo : nullptr;   // (error-feedback residuals: the handle's, NULL during a self check)
        h->lean_fb.jlo = h->lean_ok && !h->lean4_ok ? h->jlo : nullptr;
        if (h->lean4_ok && !h->force_exact) {
            h->L4.alpha64 = alpha;  // (read by the error-feedback epilogue only)
            return launch_lean4(h, Jin, Jout, (float)alpha, st, sc);
        }
        if (h->lean_ok && !h->force_exact) {
            h->lean_fb.alpha64 = alpha;
            return launch_lean2(h, Jin, Jout, (float)alpha, st, sc);
        }
        if (h->fast_ok && !is_node_dyn(h->d.dynamics_id) && !h->force_exact) return launch_fast(h, Jin, Jout, (float)alpha, st, sc);
    }
    int nlev_all = 0;
    for (int d = 0; d < h->P.n; ++d) nlev_all += h->P.dim[d];
    const bool lev_in_lds = nlev_all * 8 <= 32 * 1024;
    const size_t lev_bytes = lev_in_lds ? (size_t)nlev_all * 8 : 0;
    const bool sparse_x = sizeof(REAL) == 4 && h->sparse64 && h->vmask && h->P.n == 4 && lev_in_lds && !h->force_exact;
#define EXACT(DYN)                                                                                                     \
    if constexpr (Dyn<DYN>::DOF == 2 && sizeof(REAL) == 4) {                                                           \
        if (sparse_x) {                                                                                                \
            set_kname(h, "k_sweep", (int)DYN, tname<REAL>(), tname<PI_T>(), true, true);                              \
            hipLaunchKernelGGL((k_sweep<DYN, REAL, PI_T, true, true>), g, 256, lev_bytes, st, h->P, Jin, Jout, pi,     \
                               alpha, sc, h->P.utab, h->P.gu, h->aok32, h->vmask);                                     \
            break;                                                                                                     \
        }                                                                                                              \
    }                                                                                                                  \
    set_kname(h, "k_sweep", (int)DYN, tname<REAL>(), tname<PI_T>(), lev_in_lds, false);                               \
    if (lev_in_lds)                                                                                                    \
        hipLaunchKernelGGL((k_sweep<DYN, REAL, PI_T, true>), g, 256, lev_bytes, st, h->P, Jin, Jout, pi, alpha, sc,    \
                           h->P.utab, h->P.gu, h->aok32, (const uint4*)nullptr);                                       \
    else                                                                                                               \
        hipLaunchKernelGGL((k_sweep<DYN, REAL, PI_T, false>), g, 256, 0, st, h->P, Jin, Jout, pi, alpha, sc, h->P.utab,   \
                           h->P.gu, h->aok32, (const uint4*)nullptr);
    if constexpr (sizeof(REAL) == 8) {
        if (h->use64 && !h->force_exact) return launch_f64v2(h, Jin, Jout, alpha, st, sc);  // (f64.hip)
    }
    if constexpr (sizeof(REAL) == 4) {
        if (h->okmask3 && !h->force_exact) {
            unsigned g3 = g;
            if (h->P.n == 3 && h->P.dim[1] >= 64 && !ovr_is("XCD3", 0)) {  // an eighth of axis 1 per XCD (see the kernel)
                const long long rows = h->P.row_end - h->P.row_begin;
                long long mx = 0;
                for (int x8 = 0; x8 < 8; ++x8)
                    mx = std::max(mx, rows * ((long long)h->P.dim[1] * (x8 + 1) / 8 - (long long)h->P.dim[1] * x8 / 8) * h->P.dim[2]);
                g3 = (unsigned)(8 * ((mx + 255) / 256));
                sc.xcd_remap = 3;
                sc.nblocks = g3;
            }
#define FAST3(DYN)                                                                                                  \
    if (h->jlo) {                                                                                                   \
        set_kname(h, "k_sweep3_fast", (int)DYN, tname<PI_T>(), "float*", "double");                                 \
        hipLaunchKernelGGL((k_sweep3_fast<DYN, PI_T, float*, double>), g3, 256, 0, st, h->P, Jin, Jout, pi, (float)alpha, sc, h->P.utab, \
                           h->P.gu, h->okmask3, h->jlo, (double)alpha);                                             \
    } else {                                                                                                        \
        set_kname(h, "k_sweep3_fast", (int)DYN, tname<PI_T>());                                                     \
        hipLaunchKernelGGL((k_sweep3_fast<DYN, PI_T>), g3, 256, 0, st, h->P, Jin, Jout, pi, (float)alpha, sc, h->P.utab, h->P.gu, \
                           h->okmask3);                                                                             \
    }
            switch (h->d.dynamics_id) {
                case PVI_DYN_HELICOPTER: FAST3(PVI_DYN_HELICOPTER); break;
                case PVI_DYN_KINCAR: FAST3(PVI_DYN_KINCAR); break;
                case PVI_DYN_QUARTERCAR: FAST3(PVI_DYN_QUARTERCAR); break;
                case PVI_DYN_HOLONOMIC: FAST3(PVI_DYN_HOLONOMIC); break;
                default: FAST3(PVI_DYN_LONGCAR); break;
            }
#undef FAST3
            HIPCHK(hipGetLastError());
            return PVI_OK;
        }
    }
#define SWEEP3(DYN)                                                                                                  \
    set_kname(h, "k_sweep3", (int)DYN, tname<REAL>(), tname<PI_T>());                                                \
    hipLaunchKernelGGL((k_sweep3<DYN, REAL, PI_T>), g, 256, 0, st, h->P, Jin, Jout, pi, alpha, sc, h->P.utab, h->P.gu, \
                       h->aok32)
    switch (h->d.dynamics_id) {
        case PVI_DYN_HELICOPTER: SWEEP3(PVI_DYN_HELICOPTER); break;
        case PVI_DYN_KINCAR: SWEEP3(PVI_DYN_KINCAR); break;
        case PVI_DYN_QUARTERCAR: SWEEP3(PVI_DYN_QUARTERCAR); break;
        case PVI_DYN_HOLONOMIC: SWEEP3(PVI_DYN_HOLONOMIC); break;
        case PVI_DYN_LONGCAR: SWEEP3(PVI_DYN_LONGCAR); break;
        case PVI_DYN_PENDULUM: EXACT(PVI_DYN_PENDULUM) break;
        case PVI_DYN_CARTPOLE: EXACT(PVI_DYN_CARTPOLE) break;
        case PVI_DYN_CARTPOLE_SW:   // (float32 handles only -- pvi_create; reached by pvi_self_check's plain-gather pass)
            if constexpr (sizeof(REAL) == 4) {
                set_kname(h, "k_sweep", (int)PVI_DYN_CARTPOLE_SW, tname<REAL>(), tname<PI_T>(), lev_in_lds, false);
                if (lev_in_lds)
                    hipLaunchKernelGGL((k_sweep<PVI_DYN_CARTPOLE_SW, REAL, PI_T, true>), g, 256, lev_bytes, st, h->P, Jin, Jout, pi, alpha, sc,
                                       h->P.utab, h->P.gu, h->aok32, (const uint4*)nullptr);
                else
                    hipLaunchKernelGGL((k_sweep<PVI_DYN_CARTPOLE_SW, REAL, PI_T, false>), g, 256, 0, st, h->P, Jin, Jout, pi, alpha, sc,
                                       h->P.utab, h->P.gu, h->aok32, (const uint4*)nullptr);
            }
            break;
        case PVI_DYN_TWOLINK: EXACT(PVI_DYN_TWOLINK) break;
        case PVI_DYN_NODE_1x1: EXACT(PVI_DYN_NODE_1x1) break;
        case PVI_DYN_NODE_2x1: EXACT(PVI_DYN_NODE_2x1) break;
        case PVI_DYN_NODE_2x2: EXACT(PVI_DYN_NODE_2x2) break;
        case PVI_DYN_TABLE:
            if (!h->packed && (!h->d_xnext || !h->d_G)) return fail(PVI_ESTATE, "tier B sweep before pvi_set_tables");
            {
                // nodes per workgroup / actions per LDS chunk: ~2048 cells of Q (<= 16 KB) per pass
                const int A = h->A;
                // cells of Q per pass (16 KB of LDS in float64)
                int tab_cells = 2048;
                const int npb = A >= tab_cells ? 1 : std::max(1, std::min(256, tab_cells / A));
                const int achunk = A >= tab_cells ? tab_cells : A;
                if (h->packed) {
                    const unsigned gp = (unsigned)((h->owned + 255) / 256);
                    sc.nblocks = gp;
                    set_kname(h, "k_sweep_tablep", h->P.n, tname<REAL>(), tname<PI_T>());
                    switch (h->P.n) {
                        case 2:
                            hipLaunchKernelGGL((k_sweep_tablep<2, REAL, PI_T>), gp, 256, 0, st, h->P,
                                               (const TabRec<2, REAL>*)h->d_pack, Jin, Jout, pi, alpha, sc);
                            break;
                        case 3:
                            hipLaunchKernelGGL((k_sweep_tablep<3, REAL, PI_T>), gp, 256, 0, st, h->P,
                                               (const TabRec<3, REAL>*)h->d_pack, Jin, Jout, pi, alpha, sc);
                            break;
                        default:
                            hipLaunchKernelGGL((k_sweep_tablep<4, REAL, PI_T>), gp, 256, 0, st, h->P,
                                               (const TabRec<4, REAL>*)h->d_pack, Jin, Jout, pi, alpha, sc);
                            break;
                    }
                    HIPCHK(hipGetLastError());
                    return PVI_OK;
                }
                const int qs_doubles = (int)(((size_t)npb * achunk * sizeof(REAL) + 7) / 8);
                int nlev = 0;
                for (int d = 0; d < h->P.n; ++d) nlev += h->P.dim[d];
                const int lev_lds = nlev * 8 <= 24 * 1024;
                const size_t lds = (size_t)(qs_doubles + (lev_lds ? nlev : 0)) * 8;
                const unsigned gt = (unsigned)((h->owned + npb - 1) / npb);
                sc.nblocks = gt;
                int lpn_t = 0;
                while ((npb << (lpn_t + 1)) <= 256 && (2 << lpn_t) <= 16 && (4 << lpn_t) <= achunk) ++lpn_t;
#define TABLE(NN, LL)                                                                                                   \
    set_kname(h, "k_sweep_table", (int)NN, tname<REAL>(), tname<PI_T>(), (bool)LL);                                     \
    hipLaunchKernelGGL((k_sweep_table<NN, REAL, PI_T, LL>), gt, 256, lds, st, h->P, h->d_xnext, h->d_G, h->d_ok, Jin, Jout, \
                       pi, alpha, sc, npb, achunk, qs_doubles, lpn_t)
                switch (h->P.n * 2 + lev_lds) {
                    case 4: TABLE(2, false); break;
                    case 5: TABLE(2, true); break;
                    case 6: TABLE(3, false); break;
                    case 7: TABLE(3, true); break;
                    case 8: TABLE(4, false); break;
                    default: TABLE(4, true); break;
                }
#undef TABLE
            }
            break;
        default:
            return fail(PVI_EINVAL, "unknown dynamics_id");
    }
#undef EXACT
#undef SWEEP3
    HIPCHK(hipGetLastError());
    return PVI_OK;
}

int launch_sweep(pvi_problem* h, int src, double alpha, hipStream_t st, int k, double tol, int deferred) {
    SweepCtl sc;
    sc.ctrl = h->ctrl;
    sc.slot = h->slots + (size_t)STAT_WORDS * k;
    sc.result = h->results + 4 * k;
    sc.tol = tol;
    sc.k = k;
    sc.nblocks = 0;
    sc.split_finish = deferred ? 2 : 0;  // (2: the 2-D lean sweep folds the previous sweep's statistics itself)
    sc.xcd_remap = 0;
    sc.regtab = 0;
    sc.win_bytes = 0;
#ifdef PVI_TRACE
    fprintf(stderr, "PVI_TRACE launch h=%p k=%d src=%d Jin=%p Jout=%p pi=%p tol=%g %s\n", (void*)h, k, src, h->J[src], h->J[src ^ 1], h->pi, tol, h->kname);
    struct TraceSync {
        pvi_problem* h;
        hipStream_t st;
        ~TraceSync() {
            if (PVI_TRACE >= 2) {
                const hipError_t e = hipStreamSynchronize(st);
                fprintf(stderr, "PVI_TRACE done h=%p %s\n", (void*)h, hipGetErrorString(e));
            }
        }
    } trace_sync{h, st};
#endif
    if (h->d.dtype == PVI_F64)
        return h->pi_size == 1 ? launch_sweep_t<double, unsigned char>(h, src, alpha, st, sc)
                               : launch_sweep_t<double, unsigned short>(h, src, alpha, st, sc);
    return h->pi_size == 1 ? launch_sweep_t<float, unsigned char>(h, src, alpha, st, sc)
                           : launch_sweep_t<float, unsigned short>(h, src, alpha, st, sc);
}

extern "C" int pvi_sweep(pvi_handle h, int32_t max_sweeps, double alpha, double tol, double* stats,
                         int32_t* sweeps_done) {
    if (!h) return fail(PVI_EINVAL, "NULL handle");
    if (max_sweeps < 0) return fail(PVI_EINVAL, "max_sweeps < 0");
    if (h->P.store_begin != 0 || h->P.store_end != h->P.dim[0] || h->P.row_begin != 0 || h->P.row_end != h->P.dim[0])
        return fail(PVI_ESTATE, "pvi_sweep needs a whole-grid handle; use pvi_sweep_async + halo exchange for slabs");
    HIPCHK(hipSetDevice(h->device));
    int done_total = 0;
    float ms_total = 0.f;
    bool stopped = false;
    while (done_total < max_sweeps && !stopped) {
        const int nb = max_sweeps - done_total < MAX_BATCH ? max_sweeps - done_total : MAX_BATCH;
        hipLaunchKernelGGL(k_reset_stats, grid_for(STAT_WORDS * nb), 256, 0, h->stream, h->slots, STAT_WORDS * nb);
        hipLaunchKernelGGL(k_begin_batch, 1, 1, 0, h->stream, h->ctrl);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(h->ev0, h->stream));
        int src = h->cur;
        if (multi64_applies(h)) {  // ONE launch for the batch: grid barriers between the sweeps, the stop test on the device
            int rc = launch_multi64(h, src, alpha, tol, nb);
            if (rc) return rc;
        } else if (multi32_applies(h)) {  // ... the 2-D float32 window sweep likewise (k_sweep_leanm)
            int rc = launch_multi32(h, src, alpha, tol, nb);
            if (rc) return rc;
        } else {
            // 2-D float32 LDS-window sweep: deferred fold -- sweep k folds sweep k - 1, one k_sweep_finish for the batch's last
            const bool deferred = h->d.dtype == PVI_F32 && h->lean_ok && !h->lean4_ok && !h->force_exact && !h->spline &&
                                  !ovr_is("DEFER", 0);
            for (int k = 0; k < nb; ++k) {
                int rc = launch_sweep(h, src, alpha, h->stream, k, tol, deferred ? 1 : 0);
                if (rc) return rc;
                src ^= 1;
            }
            if (deferred) {
                SweepCtl sc;
                memset(&sc, 0, sizeof(sc));
                sc.ctrl = h->ctrl;
                sc.slot = h->slots + (size_t)STAT_WORDS * (nb - 1);
                sc.result = h->results + 4 * (nb - 1);
                sc.tol = tol;
                sc.k = nb - 1;
                hipLaunchKernelGGL(k_sweep_finish, 1, STAT_SHARDS, 0, h->stream, sc);
            }
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(h->ev1, h->stream));
        Ctrl c;
        HIPCHK(hipMemcpyAsync(&c, h->ctrl, sizeof(c), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
        ms_total += ms;
        if (c.dbg[0] & 4) {  // k_sweep_lean4fbc: the float32 loop and the float64 epilogue disagree about the winner's backup
            float v32, v64;
            memcpy(&v32, &c.dbg[3], 4);
            memcpy(&v64, &c.dbg[4], 4);
            HIPCHK(hipMemsetAsync(h->ctrl->dbg, 0, sizeof(c.dbg), h->stream));
            return fail(PVI_ECORRUPT, "corruption detector: node %d action %d sweep %d of the batch: the action loop computed %.9g, the epilogue %.9g for the same backup -- "
                                      "this build's window sweep does not compute what its source says on this device (DESIGN.md 4.2d); results of this handle are not to be trusted",
                        c.dbg[1], c.dbg[2], c.dbg[5], (double)v32, (double)v64);
        }
        if (c.dbg[0] && c.dbg[1] == -77)
            return fail(PVI_EHIP, "multi-sweep launch: the grid barrier of sweep %d of the batch was never completed (k_sweep_leanm gave up after about a second)", c.dbg[2]);
        if (c.dbg[0])
            return fail(PVI_EHIP, "bounds check: idx=%d limit=%d a=%d bpw=%d vo=%d vorg=%d idxv=%d fl=%d vs=%d exact=%d pos_in=%d",
                        c.dbg[1], c.dbg[2], c.dbg[3], c.dbg[4], c.dbg[5], c.dbg[6], c.dbg[7], c.dbg[8], c.dbg[9], c.dbg[10],
                        c.dbg[11]);
        if (c.halo_err) return fail(PVI_EHALO, "a gather left the stored rows");
        if (stats && c.k_done)
            HIPCHK(hipMemcpy(stats + 4 * (size_t)done_total, h->results, sizeof(double) * 4 * c.k_done,
                             hipMemcpyDeviceToHost));
        if (c.k_done == 0 && !c.done) {  // (never with the shipped kernels: a sweep always records itself)
            h->last_ms = ms_total;
            return fail(PVI_ESTATE, "a batch of %d sweeps recorded no sweep", nb);
        }
        if (c.k_done & 1) h->cur ^= 1;
        done_total += c.k_done;
        stopped = c.done != 0;
    }
    h->last_ms = ms_total;
    if (sweeps_done) *sweeps_done = done_total;
    return PVI_OK;
}

extern "C" int pvi_last_sweep_ms(pvi_handle h, float* ms) {
    if (!h || !ms) return fail(PVI_EINVAL, "NULL argument");
    *ms = h->last_ms;
    return PVI_OK;
}

extern "C" int pvi_sweep_async(pvi_handle h, double alpha, void* stream) {
    if (!h) return fail(PVI_EINVAL, "NULL handle");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    hipLaunchKernelGGL(k_reset_stats, 1, STAT_WORDS, 0, st, h->slots, STAT_WORDS);
    hipLaunchKernelGGL(k_begin_batch, 1, 1, 0, st, h->ctrl);
    int rc = launch_sweep(h, h->cur, alpha, st, 0, -1.0);
    if (rc) return rc;
    h->cur ^= 1;
    return PVI_OK;
}

extern "C" int pvi_sweep_stats(pvi_handle h, double stats3[3], void* stream) {
    if (!h || !stats3) return fail(PVI_EINVAL, "NULL argument");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    double res[4];
    Ctrl c;
    HIPCHK(hipMemcpyAsync(res, h->results, sizeof(res), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&c, h->ctrl, sizeof(c), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (c.halo_err) return fail(PVI_EHALO, "a gather left the stored rows: halo too small");
    stats3[0] = res[0];
    stats3[1] = res[1];
    stats3[2] = res[2];
    return PVI_OK;
}

// ---- self check: production kernel path against the plain-gather kernel (sanitizer-style cross check) -----------
template <typename REAL, typename PI_T>
__global__ void k_compare(const REAL* __restrict__ Ja, const REAL* __restrict__ Jb, const PI_T* __restrict__ pa,
                          const PI_T* __restrict__ pb, long long n, long long joff, unsigned long long* out) {
    // out[0] = max |Ja - Jb|, out[1] = max |Jb| (order-preserving encodings), out[2] = nodes whose action differs
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    double d = 0.0, m = 0.0;
    int diff = 0;
    if (i < n) {
        const double a = (double)Ja[joff + i], b = (double)Jb[joff + i];
        d = fabs(a - b);
        m = fabs(b);
        diff = pa[i] != pb[i];
    }
    d = wave_max(d);
    m = wave_max(m);
    const unsigned long long nd = __popcll(__ballot(diff));
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&out[0], enc_f64(d));
        atomicMax(&out[1], enc_f64(m));
        if (nd) atomicAdd(&out[2], nd);
    }
}

extern "C" int pvi_self_check(pvi_handle h, double alpha, double* max_rel_diff, int64_t* pi_mismatches) {
    if (!h || !max_rel_diff || !pi_mismatches) return fail(PVI_EINVAL, "NULL argument");
    if (h->spline) return fail(PVI_ESTATE, "self check covers the linear interpolant");
    HIPCHK(hipSetDevice(h->device));
    const size_t esz = h->d.dtype == PVI_F64 ? 8 : 4;
    void *Jb = nullptr, *pb = nullptr;
    unsigned long long* out = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(Jb);
        (void)hipFree(pb);
        (void)hipFree(out);
    };
    hipError_t e = hipMalloc(&Jb, (size_t)h->stored * esz + 64);
    if (e == hipSuccess) e = hipMalloc(&pb, (size_t)h->owned * h->pi_size);
    if (e == hipSuccess) e = hipMalloc((void**)&out, 3 * sizeof(unsigned long long));
    if (e != hipSuccess) {
        cleanup();
        return fail(PVI_ENOMEM, "self check scratch: %s", hipGetErrorString(e));
    }
    const unsigned long long init[3] = {enc_f64(0.0), enc_f64(0.0), 0ull};
    int rc = [&]() -> int {
        HIPCHK(hipMemcpyAsync(out, init, sizeof(init), hipMemcpyHostToDevice, h->stream));
        // (1) the handle's production path: J_cur -> the other buffer, pi.  (A dry run: with error-feedback storage the plain
        //     form of the window sweep runs, which leaves the residuals alone.)
        hipLaunchKernelGGL(k_reset_stats, 1, STAT_WORDS, 0, h->stream, h->slots, STAT_WORDS);
        hipLaunchKernelGGL(k_begin_batch, 1, 1, 0, h->stream, h->ctrl);
        float* const lo_keep = jlo_of(h);
        jlo_of(h) = nullptr;
        int r = launch_sweep(h, h->cur, alpha, h->stream, 0, -1.0);
        jlo_of(h) = lo_keep;
        if (r) return r;
        // (2) the plain-gather kernel (float64 dynamics, no windows, no set-up tables) into scratch buffers
        void* Ja = h->J[h->cur ^ 1];
        void* pa = h->pi;
        h->J[h->cur ^ 1] = Jb;
        h->pi = pb;
        h->force_exact = true;
        hipLaunchKernelGGL(k_reset_stats, 1, STAT_WORDS, 0, h->stream, h->slots, STAT_WORDS);
        hipLaunchKernelGGL(k_begin_batch, 1, 1, 0, h->stream, h->ctrl);
        r = launch_sweep(h, h->cur, alpha, h->stream, 0, -1.0);
        h->force_exact = false;
        h->J[h->cur ^ 1] = Ja;
        h->pi = pa;
        if (r) return r;
        const long long joff = (long long)(h->P.row_begin - h->P.store_begin) * h->plane;
        const unsigned g = grid_for(h->owned);
#define CMP(REAL, PI_T) \
    hipLaunchKernelGGL((k_compare<REAL, PI_T>), g, 256, 0, h->stream, (const REAL*)Ja, (const REAL*)Jb, (const PI_T*)pa, \
                       (const PI_T*)pb, (long long)h->owned, joff, out)
        if (esz == 8) {
            if (h->pi_size == 1) CMP(double, unsigned char); else CMP(double, unsigned short);
        } else {
            if (h->pi_size == 1) CMP(float, unsigned char); else CMP(float, unsigned short);
        }
#undef CMP
        HIPCHK(hipGetLastError());
        unsigned long long res[3];
        HIPCHK(hipMemcpyAsync(res, out, sizeof(res), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        const double d = dec_f64(res[0]), m = dec_f64(res[1]);
        *max_rel_diff = m > 0.0 ? d / m : d;
        *pi_mismatches = (int64_t)res[2];
        return PVI_OK;
    }();
    cleanup();
    return rc;
}

// ---- interpolation mode ---------------------------------------------------------------------------------
// LU factors (no pivoting: B-spline collocation matrices are totally positive) of the cubic not-a-knot
// collocation matrix of one axis, packed per row as {l2, l1, 1/d, u1, u2}; knots as FITPACK's fpregr for s=0.
static void spline_axis_host(const double* x, int n, std::vector<double>& t, std::vector<double>& lu,
                             std::vector<double>& rt, double* rho) {
    t.assign(n + 4, 0.0);
    for (int i = 0; i < 4; ++i) {
        t[i] = x[0];
        t[n + i] = x[n - 1];
    }
    for (int i = 2; i <= n - 3; ++i) t[i + 2] = x[i];
    std::vector<double> ab((size_t)n * 5, 0.0);  // ab[i][j - i + 2]
    for (int i = 0; i < n; ++i) {
        const double xv = x[i];
        int l = 3;
        while (l < n - 1 && xv >= t[l + 1]) ++l;  // fpbisp interval
        double h[4] = {1.0, 0.0, 0.0, 0.0}, hh[3];
        for (int j = 1; j <= 3; ++j) {  // fpbspl
            for (int q = 0; q < j; ++q) hh[q] = h[q];
            h[0] = 0.0;
            for (int q = 0; q < j; ++q) {
                const int li = l + q + 1, lj = li - j;
                const double f = hh[q] / (t[li] - t[lj]);
                h[q] = h[q] + f * (t[li] - xv);
                h[q + 1] = f * (xv - t[lj]);
            }
        }
        for (int q = 0; q < 4; ++q) {
            const int col = l - 3 + q, off = col - i + 2;
            if (h[q] != 0.0 && off >= 0 && off < 5) ab[(size_t)i * 5 + off] = h[q];
        }
    }
    for (int k = 0; k < n; ++k)
        for (int i = k + 1; i <= k + 2 && i < n; ++i) {
            const double m = ab[(size_t)i * 5 + (k - i + 2)] / ab[(size_t)k * 5 + 2];
            ab[(size_t)i * 5 + (k - i + 2)] = m;
            for (int j = k + 1; j <= k + 2 && j < n; ++j)
                if (j - i + 2 < 5) ab[(size_t)i * 5 + (j - i + 2)] -= m * ab[(size_t)k * 5 + (j - k + 2)];
        }
    lu.assign((size_t)n * 5, 0.0);
    for (int i = 0; i < n; ++i) {
        lu[(size_t)i * 5 + 0] = ab[(size_t)i * 5 + 0];
        lu[(size_t)i * 5 + 1] = ab[(size_t)i * 5 + 1];
        lu[(size_t)i * 5 + 2] = 1.0 / ab[(size_t)i * 5 + 2];
        lu[(size_t)i * 5 + 3] = ab[(size_t)i * 5 + 3];
        lu[(size_t)i * 5 + 4] = ab[(size_t)i * 5 + 4];
    }
    // how fast the two recurrences forget their initial state (per step), over the rows a warm-up can cross
    // (interior rows only: a warm-up that would start within 4 rows of an end starts AT the end instead, exactly)
    double r = 0.0;
    for (int i = 4; i < n - 4; ++i) {
        r = std::max(r, fabs(lu[(size_t)i * 5 + 0]) + fabs(lu[(size_t)i * 5 + 1]));
        r = std::max(r, (fabs(lu[(size_t)i * 5 + 3]) + fabs(lu[(size_t)i * 5 + 4])) * fabs(lu[(size_t)i * 5 + 2]));
    }
    *rho = r;
    // reciprocal knot differences of the fpbspl recursion at interval l: (j,i) = (1,0) (2,0) (2,1) (3,0) (3,1) (3,2)
    rt.assign((size_t)(n + 4) * 6, 0.0);
    for (int l = 3; l <= n - 1; ++l) {
        int k = 0;
        for (int j = 1; j <= 3; ++j)
            for (int i = 0; i < j; ++i, ++k) {
                const int li = l + i + 1, lj = li - j;
                rt[(size_t)l * 6 + k] = 1.0 / (t[li] - t[lj]);
            }
    }
}

// warm-up length after which a start-up error of the recurrence is below 1e-20 relative (0: do not chunk)
static int spline_warmup(double rho, int limit) {
    if (!(rho < 0.9)) return 0;
    const int w = (int)ceil(log(1e-20) / log(rho)) + 2;
    return w <= limit ? w : 0;
}

extern "C" int pvi_set_interpolation(pvi_handle h, int32_t kind) {
    if (!h) return fail(PVI_EINVAL, "NULL handle");
    if (kind == PVI_INTERP_LINEAR || kind == PVI_INTERP_NEAREST) {
        // 'nearest' (RegularGridInterpolator(method='nearest'), discretizer.py:570-587): table tier only -- the interval and
        // fraction of every cell are fixed when the tables are packed, so the kind must be chosen before pvi_set_tables
        const int want = kind == PVI_INTERP_NEAREST ? 1 : 0;
        if (want && h->d.dynamics_id != PVI_DYN_TABLE) return fail(PVI_EINVAL, "nearest-neighbour interpolation is implemented for the table tier");
        if (want != h->P.nearest && (h->packed || h->d_xnext))
            return fail(PVI_ESTATE, "the tables have been packed with another interpolation kind: call pvi_set_interpolation before pvi_set_tables");
        h->P.nearest = want;
        h->spline = false;
        return PVI_OK;
    }
    if (kind != PVI_INTERP_BICUBIC_SPLINE) return fail(PVI_EINVAL, "unknown interpolation kind %d", kind);
    if (h->P.n != 2) return fail(PVI_EINVAL, "bicubic-spline interpolation is 2-D only (discretizer.py:599-610)");
    if (h->P.dim[0] < 4 || h->P.dim[1] < 4) return fail(PVI_EINVAL, "a cubic spline needs at least 4 levels per axis");
    if (h->P.store_begin != 0 || h->P.store_end != h->P.dim[0] || h->P.row_begin != 0 || h->P.row_end != h->P.dim[0])
        return fail(PVI_EINVAL, "spline interpolation needs a whole-grid handle (the fit couples every row)");
    if (h->d.dynamics_id != PVI_DYN_TABLE && h->d.dynamics_id != PVI_DYN_PENDULUM)
        return fail(PVI_EINVAL, "no 2-D in-kernel dynamics with id %d", h->d.dynamics_id);
    if (h->d.dynamics_id == PVI_DYN_TABLE && h->packed && !h->d_xnext)
        return fail(PVI_ESTATE, "the tables were packed for the linear sweep and the raw copies dropped: call "
                                "pvi_set_interpolation before pvi_set_tables");
    HIPCHK(hipSetDevice(h->device));
    if (!h->SP.coef) {
        std::vector<double> t, lu, rt, lev0(h->P.dim[0]), lev1(h->P.dim[1]);
        double rho0 = 1.0, rho1 = 1.0;
        int rc;
        // (the descriptor's level pointers were only borrowed for pvi_create: read the device copies back)
        HIPCHK(hipMemcpy(lev0.data(), h->P.lev[0], lev0.size() * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(lev1.data(), h->P.lev[1], lev1.size() * 8, hipMemcpyDeviceToHost));
        spline_axis_host(lev0.data(), h->P.dim[0], t, lu, rt, &rho0);
        if ((rc = dev_upload(h, t.data(), t.size(), &h->SP.tx))) return rc;
        if ((rc = dev_upload(h, lu.data(), lu.size(), &h->SP.lu0))) return rc;
        if ((rc = dev_upload(h, rt.data(), rt.size(), &h->SP.rtx))) return rc;
        spline_axis_host(lev1.data(), h->P.dim[1], t, lu, rt, &rho1);
        if ((rc = dev_upload(h, t.data(), t.size(), &h->SP.ty))) return rc;
        if ((rc = dev_upload(h, lu.data(), lu.size(), &h->SP.lu1))) return rc;
        if ((rc = dev_upload(h, rt.data(), rt.size(), &h->SP.rty))) return rc;
        if ((rc = dev_alloc(h, (size_t)h->P.dim[0] * h->P.dim[1], &h->SP.work))) return rc;
        if ((rc = dev_alloc(h, (size_t)h->P.dim[0] * h->P.dim[1], &h->SP.coef))) return rc;
        // chunked substitution: more parallelism than one thread per grid line.  PVI_SPLINE_CHUNK=0 disables.
        const char* ev = ovr("SPLINE_CHUNK");
        const int want = ev ? atoi(ev) : 64;
        const int w0 = spline_warmup(rho0, 64), w1 = spline_warmup(rho1, 64);
        h->SP.chunk0 = (want > 0 && w0 > 0 && h->P.dim[0] > 2 * want) ? want : h->P.dim[0];
        h->SP.warm0 = w0;
        h->SP.chunk1 = (want > 0 && w1 > 0 && h->P.dim[1] > 2 * want) ? (want + 63) / 64 * 64 : h->P.dim[1];
        h->SP.warm1 = 64;
        h->SP.n0 = h->P.dim[0];
        h->SP.n1 = h->P.dim[1];
    }
    h->spline = true;
    return PVI_OK;
}

extern "C" int pvi_spline_coefficients(pvi_handle h, double* coef) {
    if (!h || !coef) return fail(PVI_EINVAL, "NULL argument");
    if (!h->spline) return fail(PVI_ESTATE, "spline interpolation is not enabled on this handle");
    HIPCHK(hipSetDevice(h->device));
    const SplineP& S = h->SP;
    if (h->d.dtype == PVI_F64)
        spline_fit_launch<double>(h, (const double*)h->J[h->cur], h->stream);
    else
        spline_fit_launch<float>(h, (const float*)h->J[h->cur], h->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(coef, S.coef, (size_t)S.n0 * S.n1 * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return PVI_OK;
}

// ---- tables --------------------------------------------------------------------------------------------
extern "C" int pvi_build_tables(pvi_handle h, int32_t row0, int32_t nrows, double* x_next, uint8_t* x_ok,
                                uint8_t* a_ok, double* G) {
    if (!h) return fail(PVI_EINVAL, "NULL handle");
    if (h->d.dynamics_id == PVI_DYN_TABLE) return fail(PVI_ESTATE, "no in-kernel dynamics to build tables from");
    if (h->d.dynamics_id == PVI_DYN_CARTPOLE_SW) return fail(PVI_EINVAL, "PVI_DYN_CARTPOLE_SW: tables are built in the reference's order (PVI_DYN_CARTPOLE)");
    if (nrows <= 0 || row0 < 0 || row0 + nrows > h->P.dim[0]) return fail(PVI_EINVAL, "bad row range");
    HIPCHK(hipSetDevice(h->device));
    const int N = h->P.n, A = h->A;
    const long long nodes = (long long)nrows * h->plane, node0 = (long long)row0 * h->plane;
    const long long chunk_nodes = std::max<long long>(1, (1ll << 22) / A);  // ~4M cells per pass
    double *dx = nullptr, *dG = nullptr;
    unsigned char *dxo = nullptr, *dao = nullptr;
    auto cleanup = [&]() {
        if (dx) (void)hipFree(dx);
        if (dG) (void)hipFree(dG);
        if (dxo) (void)hipFree(dxo);
        if (dao) (void)hipFree(dao);
    };
    const size_t cells = (size_t)chunk_nodes * A;
    hipError_t e = hipSuccess;
    if (x_next && e == hipSuccess) e = hipMalloc((void**)&dx, cells * N * 8);
    if (G && e == hipSuccess) e = hipMalloc((void**)&dG, cells * 8);
    if (x_ok && e == hipSuccess) e = hipMalloc((void**)&dxo, cells);
    if (a_ok && e == hipSuccess) e = hipMalloc((void**)&dao, cells);
    if (e != hipSuccess) {
        cleanup();
        return fail(PVI_ENOMEM, "table staging allocation failed: %s", hipGetErrorString(e));
    }
    for (long long s = 0; s < nodes; s += chunk_nodes) {
        const long long c = nodes - s < chunk_nodes ? nodes - s : chunk_nodes;
        const unsigned g = grid_for(c * A);
        switch (h->d.dynamics_id) {
            case PVI_DYN_HELICOPTER:
                hipLaunchKernelGGL((k_build_tables3<PVI_DYN_HELICOPTER>), g, 256, 0, h->stream, h->P, node0 + s, c, dx, dxo,
                                   dao, dG);
                break;
            case PVI_DYN_KINCAR:
                hipLaunchKernelGGL((k_build_tables3<PVI_DYN_KINCAR>), g, 256, 0, h->stream, h->P, node0 + s, c, dx, dxo,
                                   dao, dG);
                break;
            case PVI_DYN_QUARTERCAR:
                hipLaunchKernelGGL((k_build_tables3<PVI_DYN_QUARTERCAR>), g, 256, 0, h->stream, h->P, node0 + s, c, dx, dxo,
                                   dao, dG);
                break;
            case PVI_DYN_HOLONOMIC:
                hipLaunchKernelGGL((k_build_tables3<PVI_DYN_HOLONOMIC>), g, 256, 0, h->stream, h->P, node0 + s, c, dx, dxo,
                                   dao, dG);
                break;
            case PVI_DYN_LONGCAR:
                hipLaunchKernelGGL((k_build_tables3<PVI_DYN_LONGCAR>), g, 256, 0, h->stream, h->P, node0 + s, c, dx, dxo,
                                   dao, dG);
                break;
            case PVI_DYN_PENDULUM:
                hipLaunchKernelGGL((k_build_tables<PVI_DYN_PENDULUM>), g, 256, 0, h->stream, h->P, node0 + s, c, dx, dxo,
                                   dao, dG);
                break;
            case PVI_DYN_CARTPOLE:
                hipLaunchKernelGGL((k_build_tables<PVI_DYN_CARTPOLE>), g, 256, 0, h->stream, h->P, node0 + s, c, dx, dxo,
                                   dao, dG);
                break;
            case PVI_DYN_NODE_1x1:
                hipLaunchKernelGGL((k_build_tables<PVI_DYN_NODE_1x1>), g, 256, 0, h->stream, h->P, node0 + s, c, dx, dxo,
                                   dao, dG);
                break;
            case PVI_DYN_NODE_2x1:
                hipLaunchKernelGGL((k_build_tables<PVI_DYN_NODE_2x1>), g, 256, 0, h->stream, h->P, node0 + s, c, dx, dxo,
                                   dao, dG);
                break;
            case PVI_DYN_NODE_2x2:
                hipLaunchKernelGGL((k_build_tables<PVI_DYN_NODE_2x2>), g, 256, 0, h->stream, h->P, node0 + s, c, dx, dxo,
                                   dao, dG);
                break;
            default:
                hipLaunchKernelGGL((k_build_tables<PVI_DYN_TWOLINK>), g, 256, 0, h->stream, h->P, node0 + s, c, dx, dxo,
                                   dao, dG);
                break;
        }
        e = hipGetLastError();
        const size_t cc = (size_t)c * A, so = (size_t)s * A;
        if (e == hipSuccess && x_next) e = hipMemcpyAsync(x_next + so * N, dx, cc * N * 8, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess && G) e = hipMemcpyAsync(G + so, dG, cc * 8, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess && x_ok) e = hipMemcpyAsync(x_ok + so, dxo, cc, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess && a_ok) e = hipMemcpyAsync(a_ok + so, dao, cc, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) {
            cleanup();
            return fail(PVI_EHIP, "table build failed: %s", hipGetErrorString(e));
        }
    }
    cleanup();
    return PVI_OK;
}

extern "C" int pvi_policy_tables(pvi_handle h, int32_t controller_id, const double* ctl_params, double* U, double* x_next,
                                 uint8_t* ok, double* G) {
    if (!h || !U || !x_next || !ok || !G) return fail(PVI_EINVAL, "NULL argument");
    const int dyn = h->d.dynamics_id;
    if (dyn != PVI_DYN_PENDULUM && dyn != PVI_DYN_CARTPOLE && dyn != PVI_DYN_TWOLINK)
        return fail(PVI_ESTATE, "policy tables need one of the closed-form mechanical dynamics");
    if (h->d.cost_id != PVI_COST_QUADRATIC) return fail(PVI_ESTATE, "policy tables need the in-kernel quadratic cost");
    if (controller_id != PVI_CTL_TABLE && controller_id != PVI_CTL_COMPUTED_TORQUE) return fail(PVI_EINVAL, "unknown controller_id %d", controller_id);
    if (controller_id == PVI_CTL_COMPUTED_TORQUE && (dyn == PVI_DYN_CARTPOLE || !ctl_params))
        return fail(PVI_EINVAL, "computed torque needs a fully actuated system and its parameters [q_d, 2 zeta w0, w0^2]");
    HIPCHK(hipSetDevice(h->device));
    const int N = h->P.n, M = h->P.m;
    const long long nodes = (long long)h->P.dim[0] * h->plane;
    double *dU = nullptr, *dX = nullptr, *dG = nullptr, *dC = nullptr;
    unsigned char* dO = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(dU); (void)hipFree(dX); (void)hipFree(dG); (void)hipFree(dC); (void)hipFree(dO);
    };
    hipError_t e = hipMalloc((void**)&dU, (size_t)nodes * M * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&dX, (size_t)nodes * N * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&dG, (size_t)nodes * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&dO, (size_t)nodes);
    if (e == hipSuccess) e = hipMalloc((void**)&dC, 8 * 8);
    if (e == hipSuccess && controller_id == PVI_CTL_TABLE) e = hipMemcpyAsync(dU, U, (size_t)nodes * M * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess && ctl_params) e = hipMemcpyAsync(dC, ctl_params, (size_t)(N / 2 + 2) * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        const unsigned g = grid_for(nodes);
        switch (dyn) {
            case PVI_DYN_PENDULUM:
                hipLaunchKernelGGL((k_policy_tables<PVI_DYN_PENDULUM>), g, 256, 0, h->stream, h->P, controller_id, dC, dU, dX, dO, dG, nodes);
                break;
            case PVI_DYN_CARTPOLE:
                hipLaunchKernelGGL((k_policy_tables<PVI_DYN_CARTPOLE>), g, 256, 0, h->stream, h->P, controller_id, dC, dU, dX, dO, dG, nodes);
                break;
            default:
                hipLaunchKernelGGL((k_policy_tables<PVI_DYN_TWOLINK>), g, 256, 0, h->stream, h->P, controller_id, dC, dU, dX, dO, dG, nodes);
                break;
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(U, dU, (size_t)nodes * M * 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(x_next, dX, (size_t)nodes * N * 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(G, dG, (size_t)nodes * 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ok, dO, (size_t)nodes, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    cleanup();
    if (e != hipSuccess) return fail(PVI_EHIP, "pvi_policy_tables failed: %s", hipGetErrorString(e));
    return PVI_OK;
}

extern "C" int pvi_set_tables(pvi_handle h, const double* x_next, const double* G, const uint8_t* ok) {
    if (!h || !x_next || !G) return fail(PVI_EINVAL, "NULL argument");
    if (h->d.dynamics_id != PVI_DYN_TABLE) return fail(PVI_ESTATE, "handle was created with in-kernel dynamics");
    HIPCHK(hipSetDevice(h->device));
    const size_t cells = (size_t)h->owned * h->A;
    const int N = h->P.n;
    // The linear sweep streams PACKED records (k_table_pack), built chunk by chunk from the host tables, so the raw
    // float64 tables ((n+1)*8 bytes per cell) are never resident next to the records.  The raw tables are kept only
    // where a kernel reads them: spline mode, grids beyond int32 offsets, PVI_NO_PACK=1 (the per-sweep table kernel).
    const bool pack = h->stored < 0x7fffffffLL && !ovr("NO_PACK") && !h->spline;
    h->packed = false;
    if (!pack) {
        if (!h->d_xnext) {
            void* p = nullptr;
            HIPCHK(hipMalloc(&p, cells * N * 8));
            h->dev_allocs.push_back(p);
            h->d_xnext = (double*)p;
            HIPCHK(hipMalloc(&p, cells * 8));
            h->dev_allocs.push_back(p);
            h->d_G = (double*)p;
        }
        HIPCHK(hipMemcpyAsync(h->d_xnext, x_next, cells * N * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_G, G, cells * 8, hipMemcpyHostToDevice, h->stream));
        if (ok) {
            if (!h->d_ok) {
                void* p = nullptr;
                HIPCHK(hipMalloc(&p, cells));
                h->dev_allocs.push_back(p);
                h->d_ok = (unsigned char*)p;
            }
            HIPCHK(hipMemcpyAsync(h->d_ok, ok, cells, hipMemcpyHostToDevice, h->stream));
        } else {
            h->d_ok = nullptr;  // (a previously uploaded mask stays allocated until destroy)
        }
        HIPCHK(hipStreamSynchronize(h->stream));
        return PVI_OK;
    }
    // raw tables of an earlier call (e.g. before spline mode was switched off) are not needed any more
    dev_release(h, h->d_xnext);
    dev_release(h, h->d_G);
    h->d_xnext = h->d_G = nullptr;
    h->d_ok = nullptr;
    const bool f64 = h->d.dtype == PVI_F64;
    const size_t recsz = f64 ? (8 + 8 * (size_t)N + 8) : (4 + 4 * (size_t)N + 4);
    if (!h->d_pack) {
        void* p = nullptr;
        const size_t nblk = ((size_t)h->owned + TAB_NB - 1) / TAB_NB;
        HIPCHK(hipMalloc(&p, nblk * TAB_NB * (size_t)h->A * recsz));  // whole blocks of TAB_NB nodes
        h->dev_allocs.push_back(p);
        h->d_pack = p;
    }
    const size_t chunk = std::min<size_t>(cells, (size_t)1 << 22);  // cells per staged pass (<= 168 MB of staging)
    double *sx = nullptr, *sg = nullptr;
    unsigned char* so = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(sx);
        (void)hipFree(sg);
        (void)hipFree(so);
    };
    hipError_t e = hipMalloc((void**)&sx, chunk * N * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&sg, chunk * 8);
    if (e == hipSuccess && ok) e = hipMalloc((void**)&so, chunk);
    if (e == hipSuccess) e = hipMemsetAsync(h->ctrl, 0, sizeof(Ctrl), h->stream);
    for (size_t c0 = 0; c0 < cells && e == hipSuccess; c0 += chunk) {
        const size_t cc = std::min(chunk, cells - c0);
        e = hipMemcpyAsync(sx, x_next + c0 * N, cc * N * 8, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(sg, G + c0, cc * 8, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess && ok) e = hipMemcpyAsync(so, ok + c0, cc, hipMemcpyHostToDevice, h->stream);
        if (e != hipSuccess) break;
        const unsigned g = grid_for((long long)cc);
#define PACK(NN)                                                                                                       \
    if (f64)                                                                                                           \
        hipLaunchKernelGGL((k_table_pack<NN, double>), g, 256, 0, h->stream, h->P, sx, sg, so,                         \
                           (TabRec<NN, double>*)h->d_pack, (long long)c0, (long long)cc, &h->ctrl->halo_err);          \
    else                                                                                                               \
        hipLaunchKernelGGL((k_table_pack<NN, float>), g, 256, 0, h->stream, h->P, sx, sg, so,                          \
                           (TabRec<NN, float>*)h->d_pack, (long long)c0, (long long)cc, &h->ctrl->halo_err);
        switch (N) {
            case 2: PACK(2) break;
            case 3: PACK(3) break;
            default: PACK(4) break;
        }
#undef PACK
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);  // the staging buffers are reused by the next chunk
    }
    Ctrl c;
    memset(&c, 0, sizeof(c));
    if (e == hipSuccess) e = hipMemcpyAsync(&c, h->ctrl, sizeof(c), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    cleanup();
    if (e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? PVI_ENOMEM : PVI_EHIP, "pvi_set_tables failed: %s", hipGetErrorString(e));
    if (c.halo_err) {
        HIPCHK(hipMemsetAsync(h->ctrl, 0, sizeof(Ctrl), h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        return fail(PVI_EHALO, "a table entry gathers outside the stored rows: halo too small");
    }
    h->packed = true;
    return PVI_OK;
}


extern "C" int pvi_set_pi(pvi_handle h, const int64_t* pr, int32_t row0, int32_t nrows) {
    if (!h || !pr) return fail(PVI_EINVAL, "NULL argument");
    int rc = rows_check(h, row0, nrows, true);
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    const long long n = (long long)nrows * h->plane, off = (long long)(row0 - h->P.row_begin) * h->plane;
    for (long long s = 0; s < n; s += STAGE_CHUNK) {
        const long long c = n - s < STAGE_CHUNK ? n - s : STAGE_CHUNK;
        if ((rc = ensure_stage(h, c))) return rc;
        HIPCHK(hipMemcpyAsync(h->stage, pr + s, (size_t)c * 8, hipMemcpyHostToDevice, h->stream));
        if (h->pi_size == 1)
            hipLaunchKernelGGL((k_pi_from_i64<unsigned char>), grid_for(c), 256, 0, h->stream, (const long long*)h->stage,
                               (unsigned char*)h->pi + off + s, c);
        else
            hipLaunchKernelGGL((k_pi_from_i64<unsigned short>), grid_for(c), 256, 0, h->stream, (const long long*)h->stage,
                               (unsigned short*)h->pi + off + s, c);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return PVI_OK;
}

extern "C" int pvi_set_rollout_params(pvi_handle h, const double* params, int32_t n) {
    if (!h || (n > 0 && !params) || n < 0 || n > 64) return fail(PVI_EINVAL, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    std::vector<double> buf(64, 0.0);
    for (int i = 0; i < n; ++i) buf[(size_t)i] = params[i];
    if (h->roll_params) dev_release(h, (void*)h->roll_params);
    h->roll_params = nullptr;
    return dev_upload(h, buf.data(), buf.size(), &h->roll_params);
}

extern "C" int pvi_rollout(pvi_handle h, int64_t B, const double* X0, int32_t npts, double dt, double* X_traj,
                           double* U_traj, double* X_end) {
    if (!h || !X0) return fail(PVI_EINVAL, "NULL argument");
    const int dyn = h->d.dynamics_id;
    if (dyn == PVI_DYN_TABLE || is_node_dyn(dyn))
        return fail(PVI_ESTATE, "rollouts need closed-form dynamics (look-up / per-node tables only cover the grid nodes)");
    if (dyn == PVI_DYN_CARTPOLE_SW) return fail(PVI_EINVAL, "PVI_DYN_CARTPOLE_SW: rollouts run in the reference's order (PVI_DYN_CARTPOLE)");
    if ((dyn == PVI_DYN_QUARTERCAR || dyn == PVI_DYN_LONGCAR) && !h->roll_params)
        return fail(PVI_ESTATE, "this system's continuous closed form needs pvi_set_rollout_params first");
    if (h->P.store_begin != 0 || h->P.store_end != h->P.dim[0] || h->P.row_begin != 0 || h->P.row_end != h->P.dim[0])
        return fail(PVI_ESTATE, "rollouts need a whole-grid handle");
    if (B <= 0 || npts < 1) return PVI_OK;
    HIPCHK(hipSetDevice(h->device));
    const int N = h->P.n, M = h->P.m;
    double *dX0 = nullptr, *dXt = nullptr, *dUt = nullptr, *dXe = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(dX0); (void)hipFree(dXt); (void)hipFree(dUt); (void)hipFree(dXe);
    };
    hipError_t e = hipMalloc((void**)&dX0, (size_t)B * N * 8);
    if (e == hipSuccess && X_traj) e = hipMalloc((void**)&dXt, (size_t)B * npts * N * 8);
    if (e == hipSuccess && U_traj) e = hipMalloc((void**)&dUt, (size_t)B * npts * M * 8);
    if (e == hipSuccess && X_end) e = hipMalloc((void**)&dXe, (size_t)B * N * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(dX0, X0, (size_t)B * N * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        const unsigned g = grid_for(B, 64);
#define ROLL(F)                                                                                                      \
    if (h->pi_size == 1)                                                                                             \
        hipLaunchKernelGGL((k_rollout<F, unsigned char>), g, 64, 0, h->stream, h->P, (const unsigned char*)h->pi,     \
                           h->roll_params, (long long)B, dX0, npts, dt, dXt, dUt, dXe);                              \
    else                                                                                                             \
        hipLaunchKernelGGL((k_rollout<F, unsigned short>), g, 64, 0, h->stream, h->P, (const unsigned short*)h->pi,   \
                           h->roll_params, (long long)B, dX0, npts, dt, dXt, dUt, dXe);
        switch (dyn) {
            case PVI_DYN_PENDULUM: ROLL(RollMech<PVI_DYN_PENDULUM>) break;
            case PVI_DYN_CARTPOLE: ROLL(RollMech<PVI_DYN_CARTPOLE>) break;
            case PVI_DYN_TWOLINK: ROLL(RollMech<PVI_DYN_TWOLINK>) break;
            case PVI_DYN_HELICOPTER: ROLL(RollExpl<PVI_DYN_HELICOPTER>) break;
            case PVI_DYN_KINCAR: ROLL(RollExpl<PVI_DYN_KINCAR>) break;
            case PVI_DYN_QUARTERCAR: ROLL(RollExpl<PVI_DYN_QUARTERCAR>) break;
            case PVI_DYN_HOLONOMIC: ROLL(RollExpl<PVI_DYN_HOLONOMIC>) break;
            default: ROLL(RollExpl<PVI_DYN_LONGCAR>) break;
        }
#undef ROLL
        e = hipGetLastError();
    }
    if (e == hipSuccess && X_traj) e = hipMemcpyAsync(X_traj, dXt, (size_t)B * npts * N * 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess && U_traj) e = hipMemcpyAsync(U_traj, dUt, (size_t)B * npts * M * 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess && X_end) e = hipMemcpyAsync(X_end, dXe, (size_t)B * N * 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    cleanup();
    if (e != hipSuccess) return fail(PVI_EHIP, "pvi_rollout failed: %s", hipGetErrorString(e));
    return PVI_OK;
}

// ---- batched f ---------------------------------------------------------------------------------------------
extern "C" int pvi_eval_f(int32_t dyn, const double* params, int32_t n, int32_t m, int64_t B, const double* X,
                          const double* U, double* dX) {
    if (!params || !X || !U || !dX) return fail(PVI_EINVAL, "NULL argument");
    int en, em;
    if (dyn_shape(dyn, &en, &em) || is_node_dyn(dyn) || is_dyn3(dyn) || dyn == PVI_DYN_CARTPOLE_SW)
        return fail(PVI_EINVAL, "no closed-form dynamics with id %d", dyn);
    if (en != n || em != m) return fail(PVI_EINVAL, "dynamics %d needs n=%d m=%d", dyn, en, em);
    if (B <= 0) return PVI_OK;
    double *dc = nullptr, *dXd = nullptr, *dU = nullptr, *dO = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(dc);
        (void)hipFree(dXd);
        (void)hipFree(dU);
        (void)hipFree(dO);
    };
    hipError_t e = hipMalloc((void**)&dc, 16 * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&dXd, (size_t)B * n * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&dU, (size_t)B * m * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&dO, (size_t)B * n * 8);
    if (e == hipSuccess) e = hipMemcpy(dc, params, 16 * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dXd, X, (size_t)B * n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dU, U, (size_t)B * m * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const unsigned g = grid_for(B);
        if (dyn == PVI_DYN_PENDULUM)
            hipLaunchKernelGGL((k_eval_f<PVI_DYN_PENDULUM>), g, 256, 0, 0, dc, (long long)B, dXd, dU, dO);
        else if (dyn == PVI_DYN_CARTPOLE)
            hipLaunchKernelGGL((k_eval_f<PVI_DYN_CARTPOLE>), g, 256, 0, 0, dc, (long long)B, dXd, dU, dO);
        else
            hipLaunchKernelGGL((k_eval_f<PVI_DYN_TWOLINK>), g, 256, 0, 0, dc, (long long)B, dXd, dU, dO);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(dX, dO, (size_t)B * n * 8, hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return fail(PVI_EHIP, "pvi_eval_f failed: %s", hipGetErrorString(e));
    return PVI_OK;
}

#include "shard.inc"
