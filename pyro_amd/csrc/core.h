// Device-side common code of libpyrovi: problem description on the device (DevP), sweep control block, closed-form dynamics,
// interpolation, wave / workgroup reductions, the statistics protocol.  Included by every translation unit of the library
// (pyrovi.hip: API + exact / table / spline / n = 3 kernels; f64.hip: k_sweep64 family; lean.hip: float32 LDS-window families).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pyrovi.h"

// =================================================================================================
// device-side problem description (kernel argument, passed by value)
// =================================================================================================
struct DevP {
    int n, m, A, dof;
    int udim[PVI_MAX_M];
    int dim[PVI_MAX_N];
    long long strd[PVI_MAX_N];  // element strides of the stored J buffer (C order)
    long long plane;            // nodes per axis-0 row
    int row_begin, row_end;     // owned rows
    int store_begin, store_end; // stored rows (owned + halo)
    const double* lev[PVI_MAX_N];
    const double* trig[PVI_MAX_TRIG];
    const double* utab;         // [A][m]
    const double* gu;           // [A]  (u-ubar)' R (u-ubar)
    const unsigned char* aok;   // [A]  isavalidinput
    double lb[PVI_MAX_N], ub[PVI_MAX_N];   // isavalidstate box
    double glo[PVI_MAX_N], ghi[PVI_MAX_N]; // grid end points (interpolation fill test)
    double inv_step[PVI_MAX_N];
    double dt;
    double c[16];
    double Q[16], S[16], xbar[PVI_MAX_N];
    double R[4], ubar[PVI_MAX_M], ulb[PVI_MAX_M], uub[PVI_MAX_M];  // policy tables: g_u and isavalidinput at arbitrary inputs
    double EPS, INF;
    int ontarget;
    // isavalidstate beyond the box: axis-aligned obstacles (include/pyrovi.h pvi_desc.obs_*), and the cost functions
    // that test the NODE state against it (QuadraticCostFunctionWithDomainCheck)
    int nobs, obs_ax[2], domain_check, hard_inf, reach;
    double obs_half[2];
    double obs[PVI_MAX_OBS][4];
    const double* aux;          // [A] per-action constants of the dynamics (PVI_DYN_KINCAR)
    int all_aok;                // every action passes isavalidinput (the rule for box-bounded systems)
    int nearest;                // table tier: RegularGridInterpolator(method='nearest') -- the fraction of every axis snaps to 0 / 1
};

struct Ctrl {
    int done;      // set by finalize when delta <= tol
    int k_done;    // sweeps executed in the current batch
    int halo_err;  // a gather fell outside the stored rows
    unsigned ticket;  // shards that finished the current sweep
    int dbg[12];   // PVI_CHECK_BOUNDS builds: first out-of-range gather
    unsigned shard_ticket[64];  // workgroups of shard s (blockIdx.x % 64 == s) that finished
};

// The three sweep statistics are reduced through device-scope atomicMax.  One address sustains only
// ~80 atomics/us, so every sweep owns 64 shards x 4 words and a workgroup uses shard blockIdx.x % 64.
#define STAT_SHARDS 64
#define STAT_WORDS (STAT_SHARDS * 4)

// per-launch control block of a sweep kernel
struct SweepCtl {
    Ctrl* ctrl;
    unsigned long long* slot;  // this sweep's three encoded statistics
    double* result;            // [4] (max J, dmax, dmin, delta) written by the last workgroup
    double tol;                // stop criterion (dynamicprogramming.py:305), < 0: never
    int k;                     // sweep index inside the batch
    unsigned nblocks;
    int split_finish;          // 1: the statistics are folded by k_sweep_finish after the sweep kernel (large grids)
    int xcd_remap;             // k_sweep64: contiguous logical block ranges per XCD
    int regtab;                // k_sweep64m, 2-D, few actions: the per-action cells and costs stay in registers over the sweeps
    int win_bytes;             // ... and LDS bytes behind the level tables for the workgroup's window of J (0: gathers from memory)
};

// order-preserving encoding of doubles for integer atomicMax
__device__ __host__ inline unsigned long long enc_f64(double d) {
    unsigned long long u;
    memcpy(&u, &d, 8);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __host__ inline double dec_f64(unsigned long long u) {
    u = (u >> 63) ? (u & 0x7fffffffffffffffull) : ~u;
    double d;
    memcpy(&d, &u, 8);
    return d;
}

// =================================================================================================
// dynamics: per-state prologue + per-action acceleration, float64, reference operation order
// (mechanical.py:222-234 ddq = inv(H) (B u - C dq - g - d); manipulator.py:197-218 adds J^T f_ext = 0)
// =================================================================================================
template <int DYN>
struct Dyn;

// SinglePendulum / InvertedPendulum  (pendulum.py:80-150, :301-312).  c = [1/H, m1*g*lc1 (signed), d1]
template <>
struct Dyn<PVI_DYN_PENDULUM> {
    static constexpr int DOF = 1, M = 1;
    double hinv, gq, dd;
    __device__ void init(const double* c, const double* x, const double* tr) {
        hinv = c[0];
        gq = c[1] * tr[0];  // tr[0] = sin(q)
        dd = c[2] * x[1];
    }
    __device__ static void trig_from_tables(const DevP& P, const int* i, double* tr) { tr[0] = P.trig[0][i[0]]; }
    __device__ static void trig_from_state(const double* x, double* tr) { tr[0] = sin(x[0]); }
    __device__ void accel(const double* u, double* a) const {
        double rhs = (u[0] - gq) - dd;
        a[0] = hinv * rhs;
    }
    // acc(u) = a + B u (exact algebra; used by the f32 fast path, which re-checks near the bounds)
    __device__ void affine(double* a, double (*B)[M]) const {
        a[0] = hinv * ((0.0 - gq) - dd);
        B[0][0] = hinv;
    }
};

// CartPole (cartpole.py:369-437).  c = [m1+m2, m2*lcg, m2*lcg^2, -m2*lcg, m2*g*lcg]
// SW (PVI_DYN_CARTPOLE_SW, opt-in): the generalised coordinates in the order q = (theta, x) -- state (theta, x, dtheta, dx).
// Same system, same arithmetic; the two rows of inv(H) and of its product with the gravity term change places at the end of
// init, the angle and its rate are read from the other slots.  (The float32 window sweep of 4-D grids runs its lanes along the
// LAST axis: in this order that is dx, which the displacement does not depend on -- pyro_amd/planning/permuted.py.)  A separate
// instantiation, so that the reference-order kernels stay the code objects that have run (tools/kernel_manifest.py).
template <bool SW>
struct DynCartPole {
    static constexpr int DOF = 2, M = 1;
    static constexpr int TH = SW ? 0 : 1;   // the angle's slot among the coordinates (its rate: 2 + TH)
    double i00, i10, t0, t1, cdq0;
    // ONE_DIV: the three entries of inv(H) from one reciprocal of the determinant (an ulp off the reference's three divisions):
    // only where bit-identity with the reference is not the point -- the float64 epilogue of the float32 feedback sweep
    template <bool ONE_DIV = false>
    __device__ void init(const double* c, const double* x, const double* tr) {
        const double cth = tr[0], sth = tr[1], dth = x[2 + TH];
        const double H00 = c[0], H01 = c[1] * cth, H11 = c[2];
        const double C01 = (c[3] * sth) * dth;
        cdq0 = C01 * dth;
        const double r1 = -(c[4] * sth);
        const double det = H00 * H11 - H01 * H01;
        double i01, i11;
        if constexpr (ONE_DIV) {
            const double rd = 1.0 / det;
            i00 = H11 * rd;
            i01 = -H01 * rd;
            i11 = H00 * rd;
        } else {
            i00 = H11 / det;
            i01 = -H01 / det;
            i11 = H00 / det;
        }
        i10 = i01;
        t0 = i01 * r1;
        t1 = i11 * r1;
        if constexpr (SW) {  // (accel and affine below then yield (ddtheta, ddx))
            const double a = i00, b = t0;
            i00 = i10;
            t0 = t1;
            i10 = a;
            t1 = b;
        }
    }
    __device__ static void trig_from_tables(const DevP& P, const int* i, double* tr) {
        tr[0] = P.trig[0][i[TH]];  // cos(theta)
        tr[1] = P.trig[1][i[TH]];  // sin(theta)
    }
    __device__ static void trig_from_state(const double* x, double* tr) {
        tr[0] = cos(x[TH]);
        tr[1] = sin(x[TH]);
    }
    __device__ void accel(const double* u, double* a) const {
        const double r0 = u[0] - cdq0;
        a[0] = i00 * r0 + t0;
        a[1] = i10 * r0 + t1;
    }
    __device__ void affine(double* a, double (*B)[M]) const {
        a[0] = t0 - i00 * cdq0;
        a[1] = t1 - i10 * cdq0;
        B[0][0] = i00;
        B[1][0] = i10;
    }
};
template <>
struct Dyn<PVI_DYN_CARTPOLE> : DynCartPole<false> {};
template <>
struct Dyn<PVI_DYN_CARTPOLE_SW> : DynCartPole<true> {};

// TwoLinkManipulator / DoublePendulum (manipulator.py:897-992, pendulum.py:400-493)
// c = [k0, m2, k1, k2, I2, k3, k4, g1c, g2c, d1, d2]  (see pyro_amd/dynamic/manipulator.py)
template <>
struct Dyn<PVI_DYN_TWOLINK> {
    static constexpr int DOF = 2, M = 2;
    double i00, i01, i10, i11, cdq0, cdq1, G0, G1, D0, D1;
    template <bool ONE_DIV = false>
    __device__ void init(const double* c, const double* x, const double* tr) {
        const double s1 = tr[0], c2 = tr[1], s2 = tr[2], s12 = tr[3];
        const double dq0 = x[2], dq1 = x[3];
        const double H00 = (c[0] + c[1] * (c[2] + c[3] * c2)) + c[4];
        const double H01 = (c[5] + c[6] * c2) + c[4];
        const double H11 = c[5] + c[4];
        const double h = c[6] * s2;
        const double C00 = -h * dq1, C10 = h * dq0, C01 = -h * (dq0 + dq1);
        cdq0 = C00 * dq0 + C01 * dq1;
        cdq1 = C10 * dq0;
        G0 = -c[7] * s1 - c[8] * s12;
        G1 = -c[8] * s12;
        D0 = c[9] * dq0;
        D1 = c[10] * dq1;
        const double det = H00 * H11 - H01 * H01;
        if constexpr (ONE_DIV) {
            const double rd = 1.0 / det;
            i00 = H11 * rd;
            i01 = -H01 * rd;
            i11 = H00 * rd;
        } else {
            i00 = H11 / det;
            i01 = -H01 / det;
            i11 = H00 / det;
        }
        i10 = i01;
    }
    __device__ static void trig_from_tables(const DevP& P, const int* i, double* tr) {
        tr[0] = P.trig[0][i[0]];                  // sin q0
        tr[1] = P.trig[1][i[1]];                  // cos q1
        tr[2] = P.trig[2][i[1]];                  // sin q1
        tr[3] = P.trig[3][i[0] * P.dim[1] + i[1]];// sin(q0+q1)
    }
    __device__ static void trig_from_state(const double* x, double* tr) {
        tr[0] = sin(x[0]);
        tr[1] = cos(x[1]);
        tr[2] = sin(x[1]);
        tr[3] = sin(x[0] + x[1]);
    }
    __device__ void accel(const double* u, double* a) const {
        const double r0 = ((u[0] - cdq0) - G0) - D0;
        const double r1 = ((u[1] - cdq1) - G1) - D1;
        a[0] = i00 * r0 + i01 * r1;
        a[1] = i10 * r0 + i11 * r1;
    }
    __device__ void affine(double* a, double (*B)[M]) const {
        const double c0 = (cdq0 + G0) + D0, c1 = (cdq1 + G1) + D1;
        a[0] = -(i00 * c0 + i01 * c1);
        a[1] = -(i10 * c0 + i11 * c1);
        B[0][0] = i00;
        B[0][1] = i01;
        B[1][0] = i10;
        B[1][1] = i11;
    }
};

// Any mechanical system through per-node tables (include/pyrovi.h PVI_DYN_NODE_*): ddq = a0(q, dq) + Bn(q) u with
// a0 = inv(H)(-C dq - g - d) per grid node and Bn = inv(H) B per position node, evaluated by the host with the
// system's own H, C, B, g, d (mechanical.py:222-234).  tr[0:DOF] = a0, tr[DOF:] = Bn (row major).
template <int DOF_, int M_>
struct DynNode {
    static constexpr int DOF = DOF_, M = M_;
    double a0[DOF], Bn[DOF][M];
    __device__ void init(const double*, const double*, const double* tr) {
#pragma unroll
        for (int i = 0; i < DOF; ++i) {
            a0[i] = tr[i];
#pragma unroll
            for (int k = 0; k < M; ++k) Bn[i][k] = tr[DOF + i * M + k];
        }
    }
    __device__ static void trig_from_tables(const DevP& P, const int* idx, double* tr) {
        long long node = idx[0], pos = idx[0];
#pragma unroll
        for (int d = 1; d < 2 * DOF; ++d) node = node * P.dim[d] + idx[d];
#pragma unroll
        for (int d = 1; d < DOF; ++d) pos = pos * P.dim[d] + idx[d];
#pragma unroll
        for (int i = 0; i < DOF; ++i) tr[i] = P.trig[0][node * DOF + i];
#pragma unroll
        for (int j = 0; j < DOF * M; ++j) tr[DOF + j] = P.trig[1][pos * (DOF * M) + j];
    }
    __device__ void accel(const double* u, double* a) const {
#pragma unroll
        for (int i = 0; i < DOF; ++i) {
            double s = Bn[i][0] * u[0];
            if (M == 2) s = s + Bn[i][M - 1] * u[M - 1];
            a[i] = a0[i] + s;
        }
    }
    __device__ void affine(double* a, double (*B)[M]) const {
#pragma unroll
        for (int i = 0; i < DOF; ++i) {
            a[i] = a0[i];
#pragma unroll
            for (int k = 0; k < M; ++k) B[i][k] = Bn[i][k];
        }
    }
};
template <>
struct Dyn<PVI_DYN_NODE_1x1> : DynNode<1, 1> {};
template <>
struct Dyn<PVI_DYN_NODE_2x1> : DynNode<2, 1> {};
template <>
struct Dyn<PVI_DYN_NODE_2x2> : DynNode<2, 2> {};

// =================================================================================================
// cost (costfunction.py:151-204): rows of M.dx first, then the outer dot, all left to right
// =================================================================================================
template <int N>
__device__ inline double quad_form(const double* M, const double* dx) {
    double out = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double row = M[i * N] * dx[0];
#pragma unroll
        for (int j = 1; j < N; ++j) row = row + M[i * N + j] * dx[j];
        const double term = dx[i] * row;
        out = (i == 0) ? term : out + term;
    }
    return out;
}
template <int N>
__device__ inline double l2norm(const double* dx) {
    double s = dx[0] * dx[0];
#pragma unroll
    for (int j = 1; j < N; ++j) s = s + dx[j] * dx[j];
    return sqrt(s);
}

// =================================================================================================
// interpolation (scipy RegularGridInterpolator 'linear', bounds_error=False, fill_value=0;
// restated in oracle/vi_oracle.py interp_nlinear)
// =================================================================================================
// interval i with lev[i] <= x < lev[i+1], clipped to [0, N-2]   (_rgi_cython.find_indices)
__device__ inline int find_interval(const double* lev, int N, double lo, double inv_step, double x) {
    double t = floor((x - lo) * inv_step);
    int i = (t < 0.0) ? 0 : (t > (double)(N - 2) ? N - 2 : (int)t);
    while (i > 0 && x < lev[i]) --i;
    while (i < N - 2 && x >= lev[i + 1]) ++i;
    return i;
}

// the same interval together with its two end levels (the fraction needs them): one loop, one pair of level reads
// per trip -- a single trip on linspace grids unless the float estimate is off by one
__device__ inline int find_interval_lv(const double* lev, int N, double lo, double inv_step, double x, double& l0,
                                       double& l1) {
    double t = floor((x - lo) * inv_step);
    int i = (t < 0.0) ? 0 : (t > (double)(N - 2) ? N - 2 : (int)t);
    l0 = lev[i];
    l1 = lev[i + 1];
    // the estimate is the interval unless rounding put x across a level: the search is entered on a wave vote, so that
    // the common case stays straight-line code (as a plain per-lane loop every cell pays the loop's bookkeeping)
    if (__builtin_amdgcn_ballot_w64((i > 0 && x < l0) || (i < N - 2 && x >= l1)) != 0ull) {
        for (;;) {
            if (i > 0 && x < l0)
                --i;
            else if (i < N - 2 && x >= l1)
                ++i;
            else
                break;
            l0 = lev[i];
            l1 = lev[i + 1];
        }
    }
    return i;
}

// float64: bit-for-bit the oracle's order.  2-D follows evaluate_linear_2d, n>2 _evaluate_linear.
template <int N>
__device__ inline double interp_f64(const double* __restrict__ J, const long long* strd, long long base,
                                    const double* y) {
    // the two corners along the last axis are neighbours in memory (stride 1): one 16-byte load per pair
    // (8-byte aligned -- global memory takes that), i.e. 2^(N-1) vector loads instead of 2^N
    typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
    if (N == 2) {
        const d2u r0 = *(const d2u*)(J + base), r1 = *(const d2u*)(J + base + strd[0]);
        const double v00 = r0.x, v01 = r0.y, v10 = r1.x, v11 = r1.y;
        const double a0 = 1.0 - y[0], a1 = 1.0 - y[1];
        return v00 * a0 * a1 + v01 * a0 * y[1] + v10 * y[0] * a1 + v11 * y[0] * y[1];
    }
    double val = 0.0;
#pragma unroll
    for (int pair = 0; pair < (1 << (N - 1)); ++pair) {
        double w = 1.0;
        long long off = base;
#pragma unroll
        for (int d = 0; d < N - 1; ++d) {
            const int bit = (pair >> (N - 2 - d)) & 1;
            w = w * (bit ? y[d] : (1.0 - y[d]));
            off += bit ? strd[d] : 0;
        }
        const d2u r = *(const d2u*)(J + off);
        // corner order of _evaluate_linear: the last axis varies fastest (bit 0), weights multiplied axis by axis
        val = val + r.x * (w * (1.0 - y[N - 1]));
        val = val + r.y * (w * y[N - 1]);
    }
    return val;
}

// float32: nested lerps along the last axis first, explicit FMAs
template <int N>
__device__ inline float interp_f32(const float* __restrict__ J, const long long* strd, long long base,
                                   const float* y) {
    float v[1 << N];
    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
#pragma unroll
    for (int pair = 0; pair < (1 << (N - 1)); ++pair) {  // the last-axis neighbours share one 8-byte load
        long long off = base;
#pragma unroll
        for (int d = 0; d < N - 1; ++d) off += ((pair >> (N - 2 - d)) & 1) ? strd[d] : 0;
        const f2u r = *(const f2u*)(J + off);
        v[2 * pair] = r.x;
        v[2 * pair + 1] = r.y;
    }
#pragma unroll
    for (int d = N - 1; d >= 0; --d) {
        const int half = 1 << d;
#pragma unroll
        for (int k = 0; k < half; ++k) v[k] = fmaf(y[d], v[2 * k + 1] - v[2 * k], v[2 * k]);
    }
    return v[0];
}

// ... the same gathers of a float32 J combined in float64 (error-feedback epilogues: the chosen action's backup once per node)
template <int N>
__device__ inline double interp_f32_in_f64(const float* __restrict__ J, const long long* strd, long long base, const double* y) {
    double v[1 << N];
    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
#pragma unroll
    for (int pair = 0; pair < (1 << (N - 1)); ++pair) {
        long long off = base;
#pragma unroll
        for (int d = 0; d < N - 1; ++d) off += ((pair >> (N - 2 - d)) & 1) ? strd[d] : 0;
        const f2u r = *(const f2u*)(J + off);
        v[2 * pair] = (double)r.x;
        v[2 * pair + 1] = (double)r.y;
    }
#pragma unroll
    for (int d = N - 1; d >= 0; --d) {
        const int half = 1 << d;
#pragma unroll
        for (int k = 0; k < half; ++k) v[k] = __builtin_fma(y[d], v[2 * k + 1] - v[2 * k], v[2 * k]);
    }
    return v[0];
}

template <typename REAL, int N>
struct Interp;
template <int N>
struct Interp<double, N> {
    __device__ static double eval(const double* J, const long long* s, long long b, const double* y) {
        return interp_f64<N>(J, s, b, y);
    }
};
template <int N>
struct Interp<float, N> {
    __device__ static float eval(const float* J, const long long* s, long long b, const double* y) {
        float yf[N];
#pragma unroll
        for (int d = 0; d < N; ++d) yf[d] = (float)y[d];
        return interp_f32<N>(J, s, b, yf);
    }
};

// =================================================================================================
// block reduction of the three sweep statistics -> encoded atomicMax
// =================================================================================================
// wave-wide maximum of a double, result in every lane.  The six steps move the two halves with DPP (vector-ALU
// register moves: row_shr 1/2/4/8 inside the 16-lane rows, then row_bcast 15 / 31) instead of ds_bpermute, which
// occupies the LDS pipe for ~15 clk per dword on gfx950; the total lands in lane 63 and is broadcast from there.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_max_step(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    return fmax(v, __hiloint2double(hi2, lo2));
}
__device__ inline double wave_max(double v) {
    v = dpp_max_step<0x111, 0xf>(v);  // row_shr:1
    v = dpp_max_step<0x112, 0xf>(v);  // row_shr:2
    v = dpp_max_step<0x114, 0xf>(v);  // row_shr:4
    v = dpp_max_step<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of each row holds the row maximum
    v = dpp_max_step<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_max_step<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave maximum
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

__device__ inline int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// `red`: 48 doubles of LDS scratch
__device__ inline void block_stats_at(double* red, double j, double dmax, double ndmin, unsigned long long* slot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    j = wave_max(j);
    dmax = wave_max(dmax);
    ndmin = wave_max(ndmin);
    if (lane == 0) {
        red[wave] = j;
        red[16 + wave] = dmax;
        red[32 + wave] = ndmin;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double v = red[16 * threadIdx.x];
        for (int w = 1; w < nw; ++w) v = fmax(v, red[16 * threadIdx.x + w]);
        const unsigned long long old = atomicMax(&slot[4 * (blockIdx.x & (STAT_SHARDS - 1)) + threadIdx.x], enc_f64(v));
        asm volatile("" ::"v"(old));  // consume the return value: the atomic has been performed
    }
}

// float32 kernels: the three statistics are float32 values (max J exactly; delta = J_new - J_old rounded once to
// float32, 6e-8 relative), reduced as order-preserving int32 keys -- one DPP-fused v_max_i32 per step, no
// canonicalisation, no LDS-pipe traffic -- and widened to the float64 slots only by the three publishing threads.
__device__ __forceinline__ int f32_key(float f) {
    const int b = __float_as_int(f);
    return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float f32_unkey(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_imax_step(int v) {
    // (`old` = the identity of the maximum: lanes the control leaves without a source take INT_MIN and keep their own value, and
    //  the compiler folds move + maximum into ONE v_max_i32_dpp -- with old = v it kept v_mov_b32_dpp + v_max_i32: 36 instead of
    //  18 vector instructions for the three statistics of a wave)
    return max(v, __builtin_amdgcn_update_dpp((int)0x80000000, v, CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ int wave_max_key(int v) {
    v = dpp_imax_step<0x111, 0xf>(v);
    v = dpp_imax_step<0x112, 0xf>(v);
    v = dpp_imax_step<0x114, 0xf>(v);
    v = dpp_imax_step<0x118, 0xf>(v);
    v = dpp_imax_step<0x142, 0xa>(v);
    v = dpp_imax_step<0x143, 0xc>(v);
    return __builtin_amdgcn_readlane(v, 63);
}
// three at once, step by step: the steps of one reduction depend on each other through a DPP read (two wait states behind the
// write), three independent chains fill each other's slots (the sequential form compiled to 18 v_max_i32_dpp with an s_nop
// behind each)
__device__ __forceinline__ void wave_max_key3(int& a, int& b, int& c) {
#define PVI_STEP3(CTRL, MASK)          \
    a = dpp_imax_step<CTRL, MASK>(a);  \
    b = dpp_imax_step<CTRL, MASK>(b);  \
    c = dpp_imax_step<CTRL, MASK>(c);
    PVI_STEP3(0x111, 0xf) PVI_STEP3(0x112, 0xf) PVI_STEP3(0x114, 0xf) PVI_STEP3(0x118, 0xf) PVI_STEP3(0x142, 0xa) PVI_STEP3(0x143, 0xc)
#undef PVI_STEP3
    a = __builtin_amdgcn_readlane(a, 63);
    b = __builtin_amdgcn_readlane(b, 63);
    c = __builtin_amdgcn_readlane(c, 63);
}
// `red`: 48 ints of LDS scratch.  WAIT: the publishing threads consume the atomics' return values, i.e. they have been
// performed when the function returns (needed by the in-kernel ticket of sweep_finish); without it they are fire and forget.
template <bool WAIT = true>
__device__ inline void block_stats_f32_at(int* red, float j, float dmax, float ndmin, unsigned long long* slot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    int kj = f32_key(j), kd = f32_key(dmax), kn = f32_key(ndmin);
    wave_max_key3(kj, kd, kn);
    if (lane == 0) {
        red[wave] = kj;
        red[16 + wave] = kd;
        red[32 + wave] = kn;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        int v = red[16 * threadIdx.x];
        for (int w = 1; w < nw; ++w) v = max(v, red[16 * threadIdx.x + w]);
        if constexpr (WAIT) {
            const unsigned long long old =
                atomicMax(&slot[4 * (blockIdx.x & (STAT_SHARDS - 1)) + threadIdx.x], enc_f64((double)f32_unkey(v)));
            asm volatile("" ::"v"(old));  // consume the return value: the atomic has been performed
        } else {
            (void)__hip_atomic_fetch_max(&slot[4 * (blockIdx.x & (STAT_SHARDS - 1)) + threadIdx.x],
                                         enc_f64((double)f32_unkey(v)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// the workgroup's three maxima -> out[0..2] (plain stores by threads 0..2; the multi-sweep kernel's barrier publishes them)
template <bool WRITE_THROUGH = false>
__device__ inline void block_max3_store(double j, double dmax, double ndmin, double* out) {
    __shared__ double red3[3][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    j = wave_max(j);
    dmax = wave_max(dmax);
    ndmin = wave_max(ndmin);
    if (lane == 0) {
        red3[0][wave] = j;
        red3[1][wave] = dmax;
        red3[2][wave] = ndmin;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double v = red3[threadIdx.x][0];
        for (int w = 1; w < nw; ++w) v = fmax(v, red3[threadIdx.x][w]);
        if constexpr (WRITE_THROUGH)  // (an sc1 store: it leaves the XCD's L2 for memory, no release fence needed)
            __hip_atomic_store((unsigned long long*)(out + threadIdx.x), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        else
            out[threadIdx.x] = v;
    }
}

#define MULTI_MAX_WG 512  // workgroups of a multi-sweep launch (statistics records per set)

// The same barrier for data that is published WRITE-THROUGH (sc1 stores: they leave the XCD's L2 for memory) and read with
// sc1 loads (which bypass the CU's L1): no L2 write-back, no invalidate -- the two fences are 1.7 us each
// (MI355X_MICROARCH.md, inter-workgroup visibility: producer "sc1 payload -> asm vmcnt(0) -> flag", consumer "sc1 loads may
// replace the acquire only when the producer stored sc1").  Every thread waits for its own stores to have left.
__device__ __forceinline__ void grid_barrier_wt(unsigned* counter, unsigned target) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

// ... with a bound on the wait (about a second): false = the other workgroups never arrived.  For a kernel that has not yet proven
// itself on hardware (k_sweep_leanm, written while the GPU boxes were closed): a wrong arrival count must end as an error code of
// pvi_sweep, not as a GPU that spins until it is reset.
__device__ __forceinline__ bool grid_barrier_wt_bounded(unsigned* counter, unsigned target) {
    __shared__ int s_arrived;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24)) {
                ok = 0;
                break;
            }
        }
        s_arrived = ok;
    }
    __syncthreads();
    return s_arrived != 0;
}

__device__ inline void block_stats(double j, double dmax, double ndmin, unsigned long long* slot) {
    __shared__ double red[3][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    j = wave_max(j);
    dmax = wave_max(dmax);
    ndmin = wave_max(ndmin);
    if (lane == 0) {
        red[0][wave] = j;
        red[1][wave] = dmax;
        red[2][wave] = ndmin;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double v = red[threadIdx.x][0];
        for (int w = 1; w < nw; ++w) v = fmax(v, red[threadIdx.x][w]);
        const unsigned long long old = atomicMax(&slot[4 * (blockIdx.x & (STAT_SHARDS - 1)) + threadIdx.x], enc_f64(v));
        asm volatile("" ::"v"(old));  // consume the return value: the atomic has been performed
    }
}

// finalize_backward_step (dynamicprogramming.py:247-261) without a second launch: the workgroup that
// draws the last ticket folds the three statistics, records them and decides the stop.  The slot
// values are read back through atomic RMWs (performed at the device coherence point, like the
// atomicMax that produced them); results / done are consumed by the NEXT kernel, after the boundary.
__device__ inline void sweep_finish(const SweepCtl& sc) {
    // No fence: a release fence would write back this XCD's dirty L2 lines (all of J_{k+1}) once per
    // workgroup.  The statistics travel in device-scope atomics only; block_stats consumes their
    // return values, so they have been performed before the barrier below is passed.
    __syncthreads();
    if (threadIdx.x < 64) {
        int last = 0;
        if (threadIdx.x == 0) {
            const unsigned sh = blockIdx.x & (STAT_SHARDS - 1);
            const unsigned in_shard = (sc.nblocks - sh + STAT_SHARDS - 1) / STAT_SHARDS;
            if (atomicAdd(&sc.ctrl->shard_ticket[sh], 1u) == in_shard - 1u) {
                sc.ctrl->shard_ticket[sh] = 0u;
                const unsigned nshards = sc.nblocks < STAT_SHARDS ? sc.nblocks : STAT_SHARDS;
                last = atomicAdd(&sc.ctrl->ticket, 1u) == nshards - 1u;
            }
        }
        last = __shfl(last, 0, 64);
        if (last) {  // wave 0 of the last workgroup folds the shards: lane = shard
            const int l = threadIdx.x;
            double v0 = dec_f64(atomicMax(&sc.slot[4 * l + 0], 0ull));
            double v1 = dec_f64(atomicMax(&sc.slot[4 * l + 1], 0ull));
            double v2 = dec_f64(atomicMax(&sc.slot[4 * l + 2], 0ull));
            v0 = wave_max(v0);
            v1 = wave_max(v1);
            v2 = wave_max(v2);
            if (l == 0) {
                const double dmin = -v2, delta = fmax(fabs(v1), fabs(dmin));
                sc.result[0] = v0;
                sc.result[1] = v1;
                sc.result[2] = dmin;
                sc.result[3] = delta;
                sc.ctrl->k_done = sc.k + 1;
                if (sc.tol >= 0.0 && delta <= sc.tol) sc.ctrl->done = 1;
                sc.ctrl->ticket = 0u;
            }
        }
    }
}

// (defined in pyrovi.hip)
__global__ void k_sweep_finish(SweepCtl sc);
__global__ void k_reset_stats(unsigned long long* slots, int n);
__global__ void k_begin_batch(Ctrl* ctrl);

// =================================================================================================
// node decoding
// =================================================================================================
template <int N>
__device__ inline void decode_node(const DevP& P, long long o, int* idx) {
    long long row = o / P.plane;
    int rem = (int)(o - row * P.plane);
    idx[0] = P.row_begin + (int)row;
#pragma unroll
    for (int d = N - 1; d >= 1; --d) {
        const int q = rem / P.dim[d];
        idx[d] = rem - q * P.dim[d];
        rem = q;
    }
}

// =================================================================================================
// terminal cost  J0[s] = h(x_s)   (dynamicprogramming.py:159-171; costfunction.py:151-165)
// =================================================================================================
// sys.isavalidstate (system.py:198-205 inclusive box; drone.py:590-611, vehicle_steering.py:1004-1021 obstacles)
template <int N>
__device__ inline bool state_valid(const DevP& P, const double* x) {
    bool bad = false;
#pragma unroll
    for (int d = 0; d < N; ++d) bad = bad || (x[d] < P.lb[d]) || (x[d] > P.ub[d]);
    const double px = x[P.obs_ax[0]], py = x[P.obs_ax[1]];
    for (int b = 0; b < P.nobs; ++b) {
        const bool on_obs = ((px + P.obs_half[0]) > P.obs[b][0]) && ((py + P.obs_half[1]) > P.obs[b][1]) &&
                            ((px - P.obs_half[0]) < P.obs[b][2]) && ((py - P.obs_half[1]) < P.obs[b][3]);
        bad = bad || on_obs;
    }
    return !bad;
}

