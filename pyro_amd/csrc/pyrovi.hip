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
template <>
struct Dyn<PVI_DYN_CARTPOLE> {
    static constexpr int DOF = 2, M = 1;
    double i00, i10, t0, t1, cdq0;
    __device__ void init(const double* c, const double* x, const double* tr) {
        const double cth = tr[0], sth = tr[1], dth = x[3];
        const double H00 = c[0], H01 = c[1] * cth, H11 = c[2];
        const double C01 = (c[3] * sth) * dth;
        cdq0 = C01 * dth;
        const double r1 = -(c[4] * sth);
        const double det = H00 * H11 - H01 * H01;
        i00 = H11 / det;
        const double i01 = -H01 / det;
        i10 = i01;
        const double i11 = H00 / det;
        t0 = i01 * r1;
        t1 = i11 * r1;
    }
    __device__ static void trig_from_tables(const DevP& P, const int* i, double* tr) {
        tr[0] = P.trig[0][i[1]];  // cos(theta)
        tr[1] = P.trig[1][i[1]];  // sin(theta)
    }
    __device__ static void trig_from_state(const double* x, double* tr) {
        tr[0] = cos(x[1]);
        tr[1] = sin(x[1]);
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

// TwoLinkManipulator / DoublePendulum (manipulator.py:897-992, pendulum.py:400-493)
// c = [k0, m2, k1, k2, I2, k3, k4, g1c, g2c, d1, d2]  (see pyro_amd/dynamic/manipulator.py)
template <>
struct Dyn<PVI_DYN_TWOLINK> {
    static constexpr int DOF = 2, M = 2;
    double i00, i01, i10, i11, cdq0, cdq1, G0, G1, D0, D1;
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
        i00 = H11 / det;
        i01 = -H01 / det;
        i10 = i01;
        i11 = H00 / det;
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
    return max(v, __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false));
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
// `red`: 48 ints of LDS scratch.  WAIT: the publishing threads consume the atomics' return values, i.e. they have been
// performed when the function returns (needed by the in-kernel ticket of sweep_finish); without it they are fire and forget.
template <bool WAIT = true>
__device__ inline void block_stats_f32_at(int* red, float j, float dmax, float ndmin, unsigned long long* slot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    const int kj = wave_max_key(f32_key(j)), kd = wave_max_key(f32_key(dmax)), kn = wave_max_key(f32_key(ndmin));
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
// Float64 sweep, second form ("exact-f64v2"): the arithmetic of k_sweep -- every value the same bit for bit -- issued with
// fewer instructions.  k_sweep is bound by SIMD issue of float64 work (C2 in float64: ~460 clk per 64 cells), so what is
// cut is instructions, not memory traffic:
//   * the state box equals the grid box (checked by the host), so ONE pair of compares per axis decides validity and fill;
//     a node whose position row leaves the box skips its action loop (every Q is INF + alpha*0 = INF, arg 0);
//   * per-action constants {u0, u1, gu, isavalidinput} sit in one 32-byte record: one scalar load per action;
//   * the fraction (x - l0) / (l1 - l0) is formed with a tabulated reciprocal: r = RN(1/d) from the host, q = RN(t r),
//     e = t - q d (one FMA, exact), y = RN(q + e r).  With a correctly rounded reciprocal this is the correctly rounded
//     quotient (Markstein's theorem; the hardware's own division sequence is the same recurrence behind a scaled rcp),
//     i.e. the bits of the reference's division -- 3 instructions instead of 13 per axis and cell;
//   * grid levels and reciprocals share one LDS table ({level, reciprocal} per entry: one 16-byte read);
//   * 4-D: the products of the position-axis weights -- the first two factors of scipy's weight product, the same for
//     every action of a node -- are formed once per node;
//   * 32-bit offsets into J while the stored slab is below 2 GiB (scalar base + 32-bit lane offset addressing).
// =================================================================================================
#ifndef PVI_T64
#define PVI_T64 4  // cells per trip of the 2-D float64 loop
#endif
struct Act64 {
    double u0, u1, gu, aok;
};

template <bool OFF32>
struct JOff;
template <>
struct JOff<true> {
    typedef unsigned T;
};
template <>
struct JOff<false> {
    typedef long long T;
};

template <bool OFF32>
__device__ __forceinline__ const double* j_at(const double* __restrict__ J, typename JOff<OFF32>::T elem) {
    if constexpr (OFF32)
        return (const double*)((const char*)J + (size_t)(elem * 8u));  // zero-extended 32-bit byte offset
    else
        return J + elem;
}

// interval of x on a linspace axis (as find_interval_lv) from the {level, reciprocal} table, and the fraction by the
// reciprocal recurrence above
__device__ __forceinline__ int interval_frac64(const double2* __restrict__ tab, int N, double lo, double inv_step, double x,
                                               double& y) {
    const double t0 = floor((x - lo) * inv_step);
    int i = (t0 < 0.0) ? 0 : (t0 > (double)(N - 2) ? N - 2 : (int)t0);
    // the estimate is the interval itself except when rounding put x across a level: straight-line reads first (so that
    // the reads of several cells can be in flight together), the search loop only for lanes that still have to move
    double2 e0 = tab[i];
    double l1 = tab[i + 1].x;
    // (the branch is on a wave vote: written as a plain per-lane loop, the compiler rotates it so that EVERY cell walks
    //  through the loop's exec-mask bookkeeping and waits for its LDS reads one at a time)
    if (__builtin_amdgcn_ballot_w64((i > 0 && x < e0.x) || (i < N - 2 && x >= l1)) != 0ull) {
        while ((i > 0 && x < e0.x) || (i < N - 2 && x >= l1)) {
            i += (i > 0 && x < e0.x) ? -1 : 1;
            e0 = tab[i];
            l1 = tab[i + 1].x;
        }
    }
    const double t = x - e0.x, d = l1 - e0.x, r = e0.y;
    const double q = t * r;
    const double e = __builtin_fma(-q, d, t);
    y = __builtin_fma(e, r, q);
    return i;
}

//   * 4-D, PATCH: a wave owns an 8 x 8 patch of the (i2, i3) velocity plane of one position node instead of 64
//     consecutive nodes along i3.  The expensive part of a cell -- two interval searches, 8 gathers, the 16-corner sum --
//     is only needed where x_next lands inside the box, but a wave pays for it as soon as ONE of its lanes does; for an
//     action the in-box nodes form a rectangle of the velocity plane, which a compact patch meets far less often than
//     a 64-node line does (two-link 101^4 x 121: 7 % of the cells are in the box, ~40 % of the (line, action) pairs hit it).
//   * SPARSE (4-D, A <= 128): which cells land in the box does not change from sweep to sweep, so it is decided once at
//     set-up (k_valid_mask: the same float64 expressions) and kept as a 128-bit mask per node.  A lane then walks the
//     set bits of ITS mask -- the action constants come from an LDS copy of the table instead of a scalar load -- and
//     the cells outside the box, whose Q is INF + alpha*0 = INF exactly, enter the argmin as one candidate (INF, first
//     clear bit).  A wave runs as many trips as its busiest lane has cells in the box instead of A (two-link 101^4 x
//     121: 7 % of the cells are in the box).
// Grid-wide barrier of the multi-sweep kernels (every workgroup of the launch is resident: cooperative launch).  `counter`
// counts arrivals monotonically over the sweeps of the launch (k_begin_batch zeroes it); `target` = arrivals after this
// sweep.  Thread 0 publishes the workgroup's stores device-wide (release: the L2s of the 8 XCDs are not coherent with each
// other -- the fence writes this XCD's dirty lines back), arrives, spins, and invalidates stale lines (acquire) before
// the workgroup reads the other workgroups' J.
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();  // (every wave's stores have been issued and acknowledged: s_waitcnt vmcnt(0) ahead of the barrier)
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

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

// MULTI (round 4): the device form of the driver loops dynamicprogramming.py:265-314 for grids whose
// workgroups are all resident.  ONE launch runs up to `nsweeps` backups: everything of a node that does not change between
// sweeps (coordinates, position row, weights, dynamics prologue) stays in its thread's registers, J ping-pongs between the
// two buffers, the three statistics of sweep k go to slot k, a grid barrier separates the sweeps, and every workgroup folds
// the statistics itself and takes the same stop decision (delta <= tol).  Same arithmetic per cell as one launch per sweep:
// J, pi, the statistics and the stop sweep are bit-identical.
template <int DYN, typename PI_T, bool OFF32, bool PATCH, bool SPARSE, bool MULTI>
__device__ __forceinline__ void sweep64_body(const DevP& P, const double* Jin, double* Jout, PI_T* __restrict__ pi, double alpha, SweepCtl sc,
                                             const Act64* __restrict__ act64, const double2* __restrict__ levr,
                                             const uint4* __restrict__ vmask, int nsweeps) {
    using D = Dyn<DYN>;
    constexpr int DOF = D::DOF, N = 2 * DOF, M = D::M;
    static_assert(!PATCH || DOF == 2, "patches tile the velocity plane of 4-D grids");
    static_assert(!SPARSE || DOF == 2, "validity masks are kept for 4-D grids");
    static_assert(!MULTI || (!PATCH && !SPARSE), "the multi-sweep form is the dense walk over consecutive nodes");
    typedef typename JOff<OFF32>::T off_t;
    typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
    if (sc.ctrl->done) return;
    extern __shared__ __attribute__((aligned(16))) double2 lr_lds[];
    // Workgroups go round-robin over the 8 XCDs, each with its own L2.  The gathers of a node land on the position rows
    // next to its own, anywhere in their velocity planes: neighbours in (i0, i1) share those planes, so every XCD gets
    // a CONTIGUOUS range of logical blocks (physical block b = 8 j + x  ->  logical x * chunk + j) and a plane is fetched
    // into one L2 instead of eight.  (Placement only; sc.xcd_remap = 0 keeps the identity.)
    unsigned lb = blockIdx.x;
    if (sc.xcd_remap == 1) {
        const unsigned nb = gridDim.x, xq = nb >> 3, xr = nb & 7u, xx = lb & 7u, jj = lb >> 3;
        lb = (xx < xr ? xx * (xq + 1u) : xr * (xq + 1u) + (xx - xr) * xq) + jj;
    } else if (sc.xcd_remap > 1) {
        // chunks of C consecutive logical blocks dealt round-robin to the XCDs (C = the blocks of a few rows of axis 0): an XCD
        // still works on neighbouring position rows, but every XCD samples the WHOLE range of axis 0 -- the sparse walk's work
        // per node depends on the position (rows near the faces of the grid leave the box), and an XCD that owns one
        // contiguous eighth of the rows is done early or late
        const unsigned C = (unsigned)sc.xcd_remap, nb = gridDim.x, full = nb / (8u * C) * (8u * C);
        if (lb < full) {
            const unsigned xx = lb & 7u, jj = lb >> 3;
            lb = ((jj / C) * 8u + xx) * C + jj % C;
        }
    }
    long long o = (long long)lb * blockDim.x + threadIdx.x;
    const long long owned = (long long)(P.row_end - P.row_begin) * P.plane;
    bool live = o < owned;
    bool store_ok = live;
    if constexpr (MULTI) {  // every thread walks the sweep loop (barriers inside): threads past the grid stand in for its last node
        if (!live) o = owned - 1;
        live = true;
    }
    int idx[N];
    if constexpr (PATCH) {
        const int np2 = (P.dim[2] + 7) >> 3, np3 = (P.dim[3] + 7) >> 3;
        const long long wid = (long long)lb * (blockDim.x >> 6) + (threadIdx.x >> 6);  // wave-uniform
        const long long pl = wid / (np2 * np3);  // position node (owned rows x dim[1])
        const int rem = (int)(wid - pl * (np2 * np3)), p2 = rem / np3, p3 = rem - p2 * np3, lane = threadIdx.x & 63;
        const long long row = pl / P.dim[1];
        idx[0] = P.row_begin + (int)row;
        idx[1] = (int)(pl - row * P.dim[1]);
        idx[2] = p2 * 8 + (lane >> 3);
        idx[3] = p3 * 8 + (lane & 7);
        live = row < (P.row_end - P.row_begin) && idx[2] < P.dim[2] && idx[3] < P.dim[3];
        store_ok = live;
        o = pl * ((long long)P.dim[2] * P.dim[3]) + (long long)idx[2] * P.dim[3] + idx[3];
    } else if (live) {
        decode_node<N>(P, o, idx);
    }
    // the node's own coordinates, from the global tables: issued BEFORE the table copy below so that the two memory round
    // trips overlap (the copy loop waits for its loads before it can write LDS)
    double xown[N];
    {
        int at = 0;
#pragma unroll
        for (int d = 0; d < N; ++d) {
            xown[d] = live ? levr[at + idx[d]].x : 0.0;
            at += P.dim[d];
        }
    }
    // (the node's own levels were requested above: their round trip and the copy's are one)
    const double2* tab[N];
    const double2* act_lds = nullptr;  // SPARSE: {u0, u1}, {gu, aok} per action behind the level tables
    {
        // only the velocity axes' tables are read per action: they go to LDS; the position axes' (a few reads per node)
        // stay in global memory -- a 1001 x 1001 grid otherwise copies 32 KB into every workgroup and holds five
        // workgroups per CU
        int at = 0, al = 0;
#pragma unroll
        for (int d = 0; d < N; ++d) {
            if (d < DOF) {
                tab[d] = levr + at;
            } else {
                for (int i = threadIdx.x; i < P.dim[d]; i += blockDim.x) lr_lds[al + i] = levr[at + i];
                tab[d] = lr_lds + al;
                al += P.dim[d];
            }
            at += P.dim[d];
        }
        if constexpr (SPARSE) {
            const double2* src = (const double2*)act64;
            for (int i = threadIdx.x; i < 2 * P.A; i += blockDim.x) lr_lds[al + i] = src[i];
            act_lds = lr_lds + al;
        }
        __syncthreads();
    }
    double st_j = -INFINITY, st_dmax = -INFINITY, st_ndmin = -INFINITY;
    if (live) {
        double x[N], dx[N];
        long long self = (long long)(idx[0] - P.store_begin) * P.strd[0];
#pragma unroll
        for (int d = 0; d < N; ++d) {
            x[d] = xown[d];
            dx[d] = x[d] - P.xbar[d];
            if (d > 0) self += idx[d] * P.strd[d];
        }
        const double gx = quad_form<N>(P.Q, dx);
        const bool on_target = P.ontarget && (l2norm<N>(dx) < P.EPS);
        // position rows of x_next: the same for every action (true division: once per node)
        bool pos_in = true, halo_bad = false;
        int ci[N];
        double y[N];
        // (estimate, both end levels in ONE round trip to the global table, a wave vote on whether anybody has to step --
        //  find_interval's dependent loads were two to three round trips at the head of every workgroup)
        {
            double xn[DOF], l0[DOF], l1[DOF];
            bool mv = false;
#pragma unroll
            for (int i = 0; i < DOF; ++i) {
                xn[i] = x[DOF + i] * P.dt + x[i];
                pos_in = pos_in && !(xn[i] < P.glo[i]) && !(xn[i] > P.ghi[i]);
                const double t0 = floor((xn[i] - P.glo[i]) * P.inv_step[i]);
                ci[i] = (t0 < 0.0) ? 0 : (t0 > (double)(P.dim[i] - 2) ? P.dim[i] - 2 : (int)t0);
                l0[i] = tab[i][ci[i]].x;
                l1[i] = tab[i][ci[i] + 1].x;
            }
#pragma unroll
            for (int i = 0; i < DOF; ++i) mv = mv || (ci[i] > 0 && xn[i] < l0[i]) || (ci[i] < P.dim[i] - 2 && xn[i] >= l1[i]);
            if (__builtin_amdgcn_ballot_w64(mv) != 0ull) {
#pragma unroll
                for (int i = 0; i < DOF; ++i) {
                    ci[i] = find_interval(P.lev[i], P.dim[i], P.glo[i], P.inv_step[i], xn[i]);
                    l0[i] = tab[i][ci[i]].x;
                    l1[i] = tab[i][ci[i] + 1].x;
                }
            }
#pragma unroll
            for (int i = 0; i < DOF; ++i) y[i] = (xn[i] - l0[i]) / (l1[i] - l0[i]);
        }
        // ---- REGTAB (multi-sweep launch, 2-D grid, at most RT actions): what a cell needs from one sweep to the next is only
        // J.  The cell of every action (offset of its lower corner, fraction along axis 1, in the box or not) and its cost
        // G do not change, so they are formed ONCE, by the expressions of the loops below, and stay in registers; a sweep is
        // then 2 A independent 16-byte loads -- ONE memory round trip instead of A / 4 dependent ones -- A bilinear sums and
        // the argmin.  J is stored write-through and loaded with sc1 loads, so the barrier between two sweeps needs no L2
        // write-back and no invalidate (grid_barrier_wt).  Same operations per cell in the same order: the same bits.
        constexpr int RT = 12;
        [[maybe_unused]] unsigned rt_off[RT];
        [[maybe_unused]] double rt_y[RT], rt_G[RT];
        [[maybe_unused]] unsigned rt_in = 0u;
        [[maybe_unused]] bool regtab = false;
        [[maybe_unused]] double jprev = 0.0;
        [[maybe_unused]] int win_r0 = 0, win_n = 0;
        [[maybe_unused]] bool win_ok = false;
        [[maybe_unused]] double* win = nullptr;
        if constexpr (MULTI && DOF == 1) {
            regtab = sc.regtab != 0 && P.A <= RT;
            if (regtab) {
                jprev = Jin[self];
#pragma unroll
                for (int a = 0; a < RT; ++a) {
                    rt_off[a] = 0u;
                    rt_y[a] = 0.0;
                    rt_G[a] = P.INF;
                }
                // The workgroup's window of J: the rows of axis 0 its nodes' position rows touch (every thread's two rows lie
                // within a few rows of its node's), whole rows.  Per sweep the window comes in ONCE, coalesced, and the 2 A
                // gathers of a thread read LDS: 22 sc1 loads per thread -- 90 KB per workgroup through the L2, none of it
                // shared in the L1 they bypass -- were 4 800 of the 12 500 cycles of a sweep on C1 (s_memtime stamps).
                __shared__ int s_wr[2];
                if (threadIdx.x == 0) {
                    s_wr[0] = 0x7fffffff;
                    s_wr[1] = -1;
                }
                __syncthreads();
                if (pos_in) {
                    atomicMin(&s_wr[0], ci[0]);
                    atomicMax(&s_wr[1], ci[0] + 1);
                }
                __syncthreads();
                win = (double*)(lr_lds + P.dim[1]);  // behind the level table of axis 1 (the only one in LDS on a 2-D grid)
                win_r0 = s_wr[0];
                win_n = s_wr[1] >= s_wr[0] ? (s_wr[1] - s_wr[0] + 1) * (int)P.strd[0] : 0;  // doubles
                win_ok = win_n > 0 && (long long)win_n * 8 <= (long long)sc.win_bytes;
                if (pos_in) {
                    const unsigned base = win_ok ? (unsigned)((long long)(ci[0] - win_r0) * P.strd[0])
                                                 : (unsigned)((long long)(ci[0] - P.store_begin) * P.strd[0]);
                    double tr[8];
                    D::trig_from_tables(P, idx, tr);
                    D dyn;
                    dyn.init(P.c, x, tr);
#pragma unroll
                    for (int a = 0; a < RT; ++a) {
                        if (a < P.A) {
                            const Act64 ac = act64[a];
                            double u[2] = {ac.u0, ac.u1}, acc[1];
                            dyn.accel(u, acc);
                            const double xa = acc[0] * P.dt + x[1];
                            const bool in = !(xa < P.glo[1]) && !(xa > P.ghi[1]);
                            double ya = 0.0;
                            const int ca = interval_frac64(tab[1], P.dim[1], P.glo[1], P.inv_step[1], in ? xa : P.glo[1], ya);
                            rt_off[a] = (base + (unsigned)ca) * 8u;
                            rt_y[a] = ya;
                            rt_in |= in ? (1u << a) : 0u;
                            const double g = on_target ? 0.0 : (gx + ac.gu);
                            rt_G[a] = (in && ac.aok != 0.0) ? g * P.dt : P.INF;
                        }
                    }
                }
            }
        }
      // REGTAB: the statistics of sweep k are loaded behind sweep k's barrier but folded while the J loads of sweep k + 1 are in
      // flight (one memory round trip for both); a sweep that turns out to come after the stop is dropped before it stores
      [[maybe_unused]] double pv0 = -INFINITY, pv1 = -INFINITY, pv2 = -INFINITY;
      [[maybe_unused]] bool pending = false;
      __shared__ double folded[4];
      for (int ks = 0;; ++ks) {  // (one trip unless MULTI)
        double best = P.INF;  // position row outside the box: every action costs INF + alpha*0, the first one wins
        int arg = 0;
        [[maybe_unused]] bool rt_done = false;
        if constexpr (MULTI && DOF == 1) {
            if (regtab) {
                rt_done = true;
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                v4u r0[RT], r1[RT];
                constexpr int WCH = 5;  // 16-byte chunks of the window per thread
                v4u wv[WCH];
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Jin, 0, 0xfffffff0u, 0x00020000);
                if (win_ok) {  // (block-uniform) the window: chunk c of thread t = doubles 2 (t + 256 c), 2 (t + 256 c) + 1
                    const unsigned org = (unsigned)((long long)(win_r0 - P.store_begin) * P.strd[0]) * 8u;
#pragma unroll
                    for (int c = 0; c < WCH; ++c) {
                        const int e = 2 * ((int)threadIdx.x + 256 * c);
                        wv[c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, e < win_n ? org + (unsigned)e * 8u : 0xffffffffu, 0, 16);
                    }
                } else if (pos_in) {
                    constexpr unsigned OOB = 0xffffffffu;  // beyond num_records: the hardware returns zeros without an access
                    const unsigned s0B = (unsigned)P.strd[0] * 8u;
#pragma unroll
                    for (int a = 0; a < RT; ++a) {
                        if (a < P.A) {  // (uniform)
                            const unsigned vo = ((rt_in >> a) & 1u) ? rt_off[a] : OOB;
                            r0[a] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, 0, 16);    // aux 16 = sc1: bypasses the CU's L1
                            r1[a] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, s0B, 16);
                        }
                    }
                }
                if (pending) {  // (uniform) the previous sweep's statistics: did it meet the tolerance?
                    pending = false;
                    if (threadIdx.x < 64) {
                        pv0 = wave_max(pv0);
                        pv1 = wave_max(pv1);
                        pv2 = wave_max(pv2);
                        if (threadIdx.x == 0) {
                            folded[0] = pv0;
                            folded[1] = pv1;
                            folded[2] = -pv2;
                            folded[3] = fmax(fabs(pv1), fabs(-pv2));
                        }
                    }
                    __syncthreads();
                    const double delta = folded[3];
                    const bool stop = sc.tol >= 0.0 && delta <= sc.tol;
                    if (blockIdx.x == 0 && threadIdx.x == 0) {
                        double* res = sc.result + 4 * (ks - 1);
                        res[0] = folded[0];
                        res[1] = folded[1];
                        res[2] = folded[2];
                        res[3] = delta;
                        sc.ctrl->k_done = ks;
                        if (stop) sc.ctrl->done = 1;
                    }
                    if (stop) break;  // sweep ks - 1 was the last one: nothing of this sweep has been stored
                }
                if (win_ok) {
                    // (a window of more than WCH x 512 doubles does not pass win_ok: see the host's win_bytes)
#pragma unroll
                    for (int c = 0; c < WCH; ++c) {
                        const int e = 2 * ((int)threadIdx.x + 256 * c);
                        if (e < win_n) win[e] = __hiloint2double((int)wv[c].y, (int)wv[c].x);
                        if (e + 1 < win_n) win[e + 1] = __hiloint2double((int)wv[c].w, (int)wv[c].z);
                    }
                    __syncthreads();
                    if (pos_in) {
                        const int s0 = (int)P.strd[0];
#pragma unroll
                        for (int a = 0; a < RT; ++a) {
                            if (a < P.A) {
                                const double* w0 = win + (rt_off[a] >> 3);  // (a cell outside the box points at the window's start)
                                const double q00 = w0[0], q01 = w0[1], q10 = w0[s0], q11 = w0[s0 + 1];
                                r0[a] = (v4u){(unsigned)__double2loint(q00), (unsigned)__double2hiint(q00), (unsigned)__double2loint(q01), (unsigned)__double2hiint(q01)};
                                r1[a] = (v4u){(unsigned)__double2loint(q10), (unsigned)__double2hiint(q10), (unsigned)__double2loint(q11), (unsigned)__double2hiint(q11)};
                            }
                        }
                    }
                }
                if (pos_in) {
                    const double a0 = 1.0 - y[0];
#pragma unroll
                    for (int a = 0; a < RT; ++a) {
                        if (a < P.A) {
                            const double q00 = __hiloint2double((int)r0[a].y, (int)r0[a].x), q01 = __hiloint2double((int)r0[a].w, (int)r0[a].z);
                            const double q10 = __hiloint2double((int)r1[a].y, (int)r1[a].x), q11 = __hiloint2double((int)r1[a].w, (int)r1[a].z);
                            const double ya = rt_y[a], a1 = 1.0 - ya;
                            const double Jt = q00 * a0 * a1 + q01 * a0 * ya + q10 * y[0] * a1 + q11 * y[0] * ya;
                            const double Jn = ((rt_in >> a) & 1u) ? Jt : 0.0;
                            const double q = rt_G[a] + alpha * Jn;
                            if (a == 0 || q < best) {
                                best = q;
                                arg = a;
                            }
                        }
                    }
                }
            }
        }
        if (pos_in && !rt_done) {
            int r0 = ci[0];
            if (r0 < P.store_begin || r0 + 1 >= P.store_end) {
                halo_bad = true;
                r0 = min(max(r0, P.store_begin), P.store_end - 2);
            }
            off_t base = (off_t)((long long)(r0 - P.store_begin) * P.strd[0]);
#pragma unroll
            for (int i = 1; i < DOF; ++i) base += (off_t)(ci[i] * P.strd[i]);
            off_t vs[DOF];
#pragma unroll
            for (int i = 0; i < DOF; ++i) vs[i] = (off_t)P.strd[DOF + i];
            const off_t s0 = (off_t)P.strd[0], s1 = DOF == 2 ? (off_t)P.strd[1] : (off_t)0;
            // 4-D: scipy's weight product over the two position axes (1 * w0 * w1: the first factor is exact)
            double wp[4] = {0.0, 0.0, 0.0, 0.0};
            if constexpr (DOF == 2) {
                const double a0 = 1.0 - y[0], a1 = 1.0 - y[1];
                wp[0] = a0 * a1;
                wp[1] = a0 * y[1];
                wp[2] = y[0] * a1;
                wp[3] = y[0] * y[1];
            }
            double tr[8];
            D::trig_from_tables(P, idx, tr);
            D dyn;
            dyn.init(P.c, x, tr);
            int a_first = 0;
            if constexpr (DOF == 1) {
                // 2-D: two actions per trip, staged (both x_next, both intervals, all four gathers, then the two sums), so
                // that the LDS and memory latencies of the second cell overlap those of the first
                const off_t s0 = (off_t)P.strd[0];
                const double a0 = 1.0 - y[0];
                // Every action valid (the rule): ONE float64 select per cell.  interval_frac64 clamps its interval, so a
                // cell outside the box may run through the interpolation on whatever it finds -- its Q is replaced by
                // INF (= INF + alpha * 0, what the general trip below forms) by the select that in-box cells need anyway.
                // Waves with a lane on the target (g = 0 there: one node of the grid, usually) take the general trip.
                if (P.all_aok && !__any(on_target)) {
                    // four cells per trip, staged: four x_next, four table reads behind ONE wave vote, eight gathers in
                    // flight, four sums -- the loop waits for its LDS and L2 round trips, not for the float64 pipe
                    constexpr int T = PVI_T64;
                    for (; a_first + T - 1 < P.A; a_first += T) {
                        double xn[T], gu[T];
                        bool in[T], any = false;
#pragma unroll
                        for (int t = 0; t < T; ++t) {
                            const Act64 ac = act64[a_first + t];
                            double u[2] = {ac.u0, ac.u1}, acc[1];
                            dyn.accel(u, acc);
                            gu[t] = ac.gu;
                            xn[t] = acc[0] * P.dt + x[1];
                            in[t] = !(xn[t] < P.glo[1]) && !(xn[t] > P.ghi[1]);
                            any = any || in[t];
                        }
                        double q[T];
#pragma unroll
                        for (int t = 0; t < T; ++t) q[t] = P.INF;
                        if (any) {
                            int ci4[T];
                            double2 e0[T];
                            double l1[T];
                            bool mv = false;
#pragma unroll
                            for (int t = 0; t < T; ++t) {
                                const double t0 = floor((xn[t] - P.glo[1]) * P.inv_step[1]);
                                ci4[t] = (t0 < 0.0) ? 0 : (t0 > (double)(P.dim[1] - 2) ? P.dim[1] - 2 : (int)t0);
                                e0[t] = tab[1][ci4[t]];
                                l1[t] = tab[1][ci4[t] + 1].x;
                            }
#pragma unroll
                            for (int t = 0; t < T; ++t)
                                mv = mv || (ci4[t] > 0 && xn[t] < e0[t].x) || (ci4[t] < P.dim[1] - 2 && xn[t] >= l1[t]);
                            if (__builtin_amdgcn_ballot_w64(mv) != 0ull) {  // rounding put some x across a level (rare)
#pragma unroll
                                for (int t = 0; t < T; ++t)
                                    while ((ci4[t] > 0 && xn[t] < e0[t].x) || (ci4[t] < P.dim[1] - 2 && xn[t] >= l1[t])) {
                                        ci4[t] += (ci4[t] > 0 && xn[t] < e0[t].x) ? -1 : 1;
                                        e0[t] = tab[1][ci4[t]];
                                        l1[t] = tab[1][ci4[t] + 1].x;
                                    }
                            }
                            d2u r0[T], r1[T];
#pragma unroll
                            for (int t = 0; t < T; ++t) {
                                const off_t bt = base + (off_t)ci4[t];
                                r0[t] = *(const d2u*)j_at<OFF32>(Jin, bt);
                                r1[t] = *(const d2u*)j_at<OFF32>(Jin, bt + s0);
                            }
#pragma unroll
                            for (int t = 0; t < T; ++t) {
                                const double tt = xn[t] - e0[t].x, dd = l1[t] - e0[t].x, rr = e0[t].y;
                                const double qq = tt * rr;
                                const double ee = __builtin_fma(-qq, dd, tt);
                                const double yt = __builtin_fma(ee, rr, qq);
                                const double c1 = 1.0 - yt;
                                const double Jt = r0[t].x * a0 * c1 + r0[t].y * a0 * yt + r1[t].x * y[0] * c1 + r1[t].y * y[0] * yt;
                                const double gt = gx + gu[t];
                                const double vt = gt * P.dt + alpha * Jt;
                                q[t] = in[t] ? vt : P.INF;
                            }
                        }
#pragma unroll
                        for (int t = 0; t < T; ++t)
                            if ((t == 0 && a_first == 0) || q[t] < best) {
                                best = q[t];
                                arg = a_first + t;
                            }
                    }
                }
                for (; a_first + 1 < P.A; a_first += 2) {
                    const Act64 ac0 = act64[a_first], ac1 = act64[a_first + 1];
                    double u0[2] = {ac0.u0, ac0.u1}, u1[2] = {ac1.u0, ac1.u1}, acc0[1], acc1[1];
                    dyn.accel(u0, acc0);
                    dyn.accel(u1, acc1);
                    const double xa = acc0[0] * P.dt + x[1], xb = acc1[0] * P.dt + x[1];
                    const bool ina = !(xa < P.glo[1]) && !(xa > P.ghi[1]), inb_ = !(xb < P.glo[1]) && !(xb > P.ghi[1]);
                    double Ja = 0.0, Jb = 0.0;
                    if (ina || inb_) {
                        double ya, yb;  // (a lane with only one of the two cells in the box evaluates the other at the box edge)
                        const int ca = interval_frac64(tab[1], P.dim[1], P.glo[1], P.inv_step[1], ina ? xa : P.glo[1], ya);
                        const int cb = interval_frac64(tab[1], P.dim[1], P.glo[1], P.inv_step[1], inb_ ? xb : P.glo[1], yb);
                        const off_t ba = base + (off_t)ca, bb = base + (off_t)cb;
                        const d2u qa0 = *(const d2u*)j_at<OFF32>(Jin, ba), qa1 = *(const d2u*)j_at<OFF32>(Jin, ba + s0);
                        const d2u qb0 = *(const d2u*)j_at<OFF32>(Jin, bb), qb1 = *(const d2u*)j_at<OFF32>(Jin, bb + s0);
                        const double a1 = 1.0 - ya, b1 = 1.0 - yb;
                        Ja = qa0.x * a0 * a1 + qa0.y * a0 * ya + qa1.x * y[0] * a1 + qa1.y * y[0] * ya;
                        Jb = qb0.x * a0 * b1 + qb0.y * a0 * yb + qb1.x * y[0] * b1 + qb1.y * y[0] * yb;
                        Ja = ina ? Ja : 0.0;
                        Jb = inb_ ? Jb : 0.0;
                    }
                    const double ga = on_target ? 0.0 : (gx + ac0.gu), gb = on_target ? 0.0 : (gx + ac1.gu);
                    const double Ga = (ina && ac0.aok != 0.0) ? ga * P.dt : P.INF, Gb = (inb_ && ac1.aok != 0.0) ? gb * P.dt : P.INF;
                    const double qa = Ga + alpha * Ja, qb = Gb + alpha * Jb;
                    if (a_first == 0 || qa < best) {
                        best = qa;
                        arg = a_first;
                    }
                    if (qb < best) {
                        best = qb;
                        arg = a_first + 1;
                    }
                }
            }
            if constexpr (SPARSE) {
                const uint4 mk = vmask[o];
                const unsigned w[4] = {mk.x, mk.y, mk.z, mk.w};
                int first_out = -1;  // lowest action whose cell leaves the box (lowest clear bit below A)
#pragma unroll
                for (int k = 3; k >= 0; --k) {
                    const unsigned z = ~w[k];
                    const int i = 32 * k + __ffs((int)z) - 1;
                    if (z && i < P.A) first_out = i;
                }
                bool have = false;
                // every lane walks its own set bits: the lanes of a wave are at different actions, so the action constants
                // come from LDS and the gathers of a wave do not coalesce.  (Measured and dropped: the wave walking the
                // UNION of its lanes' masks with scalar constants and coalesced gathers -- the lanes of a wave have nearly
                // disjoint in-box actions on the two-link arm, the union is most of A: 23.2 against 19.2 ms on 101^4.)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    unsigned m = w[k];
                    while (m != 0u) {  // (divergent: a wave leaves when its busiest lane is done)
                        // two cells per trip, staged (both x_next, the four intervals, the 16 gathers, the two sums): the
                        // gathers are scattered, their latency is what the loop waits for
                        int av[2];
                        av[0] = 32 * k + __ffs((int)m) - 1;
                        m &= m - 1u;
                        const bool two = m != 0u;
                        av[1] = two ? 32 * k + __ffs((int)m) - 1 : av[0];
                        m &= m - 1u;  // (0 & anything = 0)
                        double2 ag[2];
                        double yv[2][DOF];
                        off_t b[2];
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const double2 au = act_lds[2 * av[t]];
                            ag[t] = act_lds[2 * av[t] + 1];
                            double u[2] = {au.x, au.y}, acc[DOF];
                            dyn.accel(u, acc);
                            b[t] = base;
#pragma unroll
                            for (int i = 0; i < DOF; ++i) {
                                const int d = DOF + i;
                                const double xn = acc[i] * P.dt + x[d];  // in the box: that is what the mask bit says
                                const int c = interval_frac64(tab[d], P.dim[d], P.glo[d], P.inv_step[d], xn, yv[t][i]);
                                b[t] += (off_t)c * vs[i];
                            }
                        }
                        d2u r[2][8];
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int pr = 0; pr < 8; ++pr) {
                                const int c0 = pr >> 2, c1 = (pr >> 1) & 1, c2 = pr & 1;
                                const off_t off = b[t] + (c0 ? s0 : (off_t)0) + (c1 ? s1 : (off_t)0) + (c2 ? vs[0] : (off_t)0);
                                r[t][pr] = *(const d2u*)j_at<OFF32>(Jin, off);
                            }
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            double Jn = 0.0;
                            const double b2[2] = {1.0 - yv[t][0], yv[t][0]}, b3[2] = {1.0 - yv[t][1], yv[t][1]};
#pragma unroll
                            for (int pr = 0; pr < 8; ++pr) {
                                const int c0 = pr >> 2, c1 = (pr >> 1) & 1, c2 = pr & 1;
                                const double wgt = wp[c0 * 2 + c1] * b2[c2];
                                Jn = Jn + r[t][pr].x * (wgt * b3[0]);
                                Jn = Jn + r[t][pr].y * (wgt * b3[1]);
                            }
                            const double g = on_target ? 0.0 : (gx + ag[t].x);
                            const double G = (ag[t].y != 0.0) ? g * P.dt : P.INF;
                            const double q = G + alpha * Jn;
                            if ((t == 0 || two) && (!have || q < best)) {
                                best = q;
                                arg = av[t];
                                have = true;
                            }
                        }
                    }
                }
                // the cells outside the box: Q = INF + alpha * 0 = INF, first at action first_out
                if (first_out >= 0 && (!have || P.INF < best || (P.INF == best && first_out < arg))) {
                    best = P.INF;
                    arg = first_out;
                }
                a_first = P.A;
            }
            for (int a = a_first; a < P.A; ++a) {
                const Act64 ac = act64[a];  // wave-uniform: one scalar load
                double u[2] = {ac.u0, ac.u1}, acc[DOF], xnv[DOF];
                dyn.accel(u, acc);
                bool inb = true;
#pragma unroll
                for (int i = 0; i < DOF; ++i) {
                    const int d = DOF + i;
                    xnv[i] = acc[i] * P.dt + x[d];
                    inb = inb && !(xnv[i] < P.glo[d]) && !(xnv[i] > P.ghi[d]);
                }
                double Jn = 0.0;
                if (inb) {
                    off_t b = base;
                    double yv[DOF];
#pragma unroll
                    for (int i = 0; i < DOF; ++i) {
                        const int d = DOF + i;
                        const int c = interval_frac64(tab[d], P.dim[d], P.glo[d], P.inv_step[d], xnv[i], yv[i]);
                        b += (off_t)c * vs[i];
                    }
                    if constexpr (DOF == 1) {  // evaluate_linear_2d
                        const d2u q0 = *(const d2u*)j_at<OFF32>(Jin, b), q1 = *(const d2u*)j_at<OFF32>(Jin, b + s0);
                        const double a0 = 1.0 - y[0], a1 = 1.0 - yv[0];
                        Jn = q0.x * a0 * a1 + q0.y * a0 * yv[0] + q1.x * y[0] * a1 + q1.y * y[0] * yv[0];
                    } else {  // _evaluate_linear: corners with axis 0 slowest, weights multiplied axis by axis
                        const double b2[2] = {1.0 - yv[0], yv[0]}, b3[2] = {1.0 - yv[1], yv[1]};
#pragma unroll
                        for (int pr = 0; pr < 8; ++pr) {
                            const int c0 = pr >> 2, c1 = (pr >> 1) & 1, c2 = pr & 1;
                            const off_t off = b + (c0 ? s0 : (off_t)0) + (c1 ? s1 : (off_t)0) + (c2 ? vs[0] : (off_t)0);
                            const d2u r = *(const d2u*)j_at<OFF32>(Jin, off);
                            const double w = wp[c0 * 2 + c1] * b2[c2];
                            Jn = Jn + r.x * (w * b3[0]);
                            Jn = Jn + r.y * (w * b3[1]);
                        }
                    }
                }
                const double g = on_target ? 0.0 : (gx + ac.gu);
                const double G = (inb && ac.aok != 0.0) ? g * P.dt : P.INF;
                const double q = G + alpha * Jn;
                if (a == 0 || q < best) {
                    best = q;
                    arg = a;
                }
            }
        }
        if (halo_bad) atomicOr(&sc.ctrl->halo_err, 1);
        if (store_ok) {
            double d;
            if (rt_done) {  // write-through; the node's previous value is this thread's own last result
                __hip_atomic_store((unsigned long long*)(Jout + self), (unsigned long long)__double_as_longlong(best), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                d = best - jprev;
                jprev = best;
            } else {
                Jout[self] = best;
                d = best - Jin[self];
            }
            pi[o] = (PI_T)arg;
            st_j = best;
            st_dmax = d;
            st_ndmin = -d;
        }
        if constexpr (!MULTI) {
            break;
        } else {
            // The statistics ride on the barrier: every workgroup stores its three maxima (plain stores, published by the
            // barrier's release), and behind the barrier every workgroup reads all of them (lane = workgroup: at most 64,
            // see multi64_applies) -- no atomics, no second round trip.  Two sets, alternating: a workgroup can run at most
            // one barrier ahead of the slowest reader.
            double* part = (double*)sc.slot + (size_t)(ks & 1) * 64 * 4;
            double v0 = -INFINITY, v1 = -INFINITY, v2 = -INFINITY;
            if (rt_done) {
                // REGTAB: J and the statistics went out write-through and are read with sc1 loads: nothing to fence.
                // (Measured and not kept: arrival and statistics as ONE tagged 16-byte granule per value, polled by every
                //  workgroup -- no counter, one round trip less on paper, the same 5.0 us per sweep on C1.)
                block_max3_store<true>(st_j, st_dmax, st_ndmin, part + 4 * blockIdx.x);
                grid_barrier_wt(&sc.ctrl->ticket, (unsigned)(ks + 1) * gridDim.x);
                if (threadIdx.x < 64) {
                    const int l = threadIdx.x;
                    if (l < (int)gridDim.x) {  // (sc1 loads: they bypass the CU's L1, which may hold these words from two sweeps ago)
                        auto ld = [&](int k) {
                            return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)(part + 4 * l + k), __ATOMIC_RELAXED,
                                                                                      __HIP_MEMORY_SCOPE_AGENT));
                        };
                        v0 = ld(0);
                        v1 = ld(1);
                        v2 = ld(2);
                    }
                }
                if (ks + 1 < nsweeps) {  // folded behind the next sweep's J loads
                    pv0 = v0;
                    pv1 = v1;
                    pv2 = v2;
                    pending = true;
                    const double* t = Jin;  // ping-pong
                    Jin = Jout;
                    Jout = const_cast<double*>(t);
                    st_j = st_dmax = st_ndmin = -INFINITY;
                    continue;
                }
            } else {
                block_max3_store(st_j, st_dmax, st_ndmin, part + 4 * blockIdx.x);
                grid_barrier(&sc.ctrl->ticket, (unsigned)(ks + 1) * gridDim.x);
                if (threadIdx.x < 64) {
                    const int l = threadIdx.x;
                    const bool has = l < (int)gridDim.x;
                    v0 = has ? __builtin_nontemporal_load(part + 4 * l + 0) : -INFINITY;
                    v1 = has ? __builtin_nontemporal_load(part + 4 * l + 1) : -INFINITY;
                    v2 = has ? __builtin_nontemporal_load(part + 4 * l + 2) : -INFINITY;
                }
            }
            if (threadIdx.x < 64) {
                const int l = threadIdx.x;
                v0 = wave_max(v0);
                v1 = wave_max(v1);
                v2 = wave_max(v2);
                if (l == 0) {
                    folded[0] = v0;
                    folded[1] = v1;
                    folded[2] = -v2;
                    folded[3] = fmax(fabs(v1), fabs(-v2));
                }
            }
            __syncthreads();
            const double delta = folded[3];
            const bool stop = sc.tol >= 0.0 && delta <= sc.tol;
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                double* res = sc.result + 4 * ks;
                res[0] = folded[0];
                res[1] = folded[1];
                res[2] = folded[2];
                res[3] = delta;
                sc.ctrl->k_done = ks + 1;
                if (stop) sc.ctrl->done = 1;
            }
            if (stop || ks + 1 >= nsweeps) break;
            const double* t = Jin;  // ping-pong
            Jin = Jout;
            Jout = const_cast<double*>(t);
            st_j = st_dmax = st_ndmin = -INFINITY;
            __syncthreads();  // (`folded` is rewritten by the next sweep)
        }
      }
    }
    if constexpr (!MULTI) {
        block_stats(st_j, st_dmax, st_ndmin, sc.slot);
        sweep_finish(sc);
    }
}

template <int DYN, typename PI_T, bool OFF32, bool PATCH, bool SPARSE = false>
__global__ __launch_bounds__(256) void k_sweep64(DevP P, const double* __restrict__ Jin, double* __restrict__ Jout,
                                                 PI_T* __restrict__ pi, double alpha, SweepCtl sc,
                                                 const Act64* __restrict__ act64, const double2* __restrict__ levr,
                                                 const uint4* __restrict__ vmask) {
    sweep64_body<DYN, PI_T, OFF32, PATCH, SPARSE, false>(P, Jin, Jout, pi, alpha, sc, act64, levr, vmask, 1);
}
// (Jin / Jout without __restrict__: the kernel swaps them between its sweeps)
template <int DYN, typename PI_T>
__global__ __launch_bounds__(256) void k_sweep64m(DevP P, const double* Jin, double* Jout, PI_T* __restrict__ pi, double alpha, SweepCtl sc,
                                                  const Act64* __restrict__ act64, const double2* __restrict__ levr, int nsweeps) {
    sweep64_body<DYN, PI_T, true, false, false, true>(P, Jin, Jout, pi, alpha, sc, act64, levr, nullptr, nsweeps);
}

// Validity masks of the SPARSE float64 sweep: bit a of a node's 128-bit word is set when the position row and the cell of
// action a land inside the box -- the float64 expressions of k_sweep64, evaluated once.  count[0] += cells in the box.
template <int DYN>
__global__ __launch_bounds__(256) void k_valid_mask(DevP P, const Act64* __restrict__ act64, uint4* __restrict__ vmask,
                                                    unsigned long long* __restrict__ count) {
    using D = Dyn<DYN>;
    constexpr int DOF = D::DOF, N = 2 * DOF;
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long owned = (long long)(P.row_end - P.row_begin) * P.plane;
    unsigned w[4] = {0u, 0u, 0u, 0u};
    int nin = 0;
    if (o < owned) {
        int idx[N];
        decode_node<N>(P, o, idx);
        double x[N];
#pragma unroll
        for (int d = 0; d < N; ++d) x[d] = P.lev[d][idx[d]];
        bool pos_in = true;
#pragma unroll
        for (int i = 0; i < DOF; ++i) {
            const double xn = x[DOF + i] * P.dt + x[i];
            pos_in = pos_in && !(xn < P.glo[i]) && !(xn > P.ghi[i]);
        }
        if (pos_in) {
            double tr[8];
            D::trig_from_tables(P, idx, tr);
            D dyn;
            dyn.init(P.c, x, tr);
            for (int a = 0; a < P.A; ++a) {
                const Act64 ac = act64[a];
                double u[2] = {ac.u0, ac.u1}, acc[DOF];
                dyn.accel(u, acc);
                bool inb = true;
#pragma unroll
                for (int i = 0; i < DOF; ++i) {
                    const int d = DOF + i;
                    const double xn = acc[i] * P.dt + x[d];
                    inb = inb && !(xn < P.glo[d]) && !(xn > P.ghi[d]);
                }
                if (inb) {
                    ++nin;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if ((a >> 5) == k) w[k] |= 1u << (a & 31);
                }
            }
        }
        vmask[o] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    // block total -> one atomic
    __shared__ int s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    if (nin) atomicAdd(&s_n, nin);
    __syncthreads();
    if (threadIdx.x == 0 && s_n) atomicAdd(count, (unsigned long long)s_n);
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

template <int DYN, typename PI_T>
__global__ __launch_bounds__(256) void k_sweep3_fast(DevP P, const float* __restrict__ Jin, float* __restrict__ Jout, PI_T* __restrict__ pi,
                                                     float alpha, SweepCtl sc, const double* __restrict__ utab,
                                                     const double* __restrict__ gutab, const unsigned long long* __restrict__ okmask) {
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
        double xn[N], u0[M];
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
                    y[d] = (float)((xn[d] - l0) / (l1 - l0));
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
// f32 fast path ("v1").  Same recursion; the inner loop is float32 and system independent:
//   * per state (float64, reference operation order): node coordinates, position rows of x_next
//     (exact), dynamics prologue, and the AFFINE form of the velocity displacement measured in grid
//     cells,  rel_d(u) = (x_next_d - x_d)/dx_d = ta_d + sum_k tB_dk u_k   (mechanical systems are
//     affine in u), rounded once to float32;
//   * per action (float32): rel, interval = own index + floor(rel), fraction = rel - floor(rel),
//     2^n gathers, lerps, Bellman min.  Working relative to the node keeps |rel| small, so the
//     fraction carries ~1e-6 cells of error instead of the 6e-5 of an absolute float32 coordinate;
//   * validity is decided from the float32 margin to the box unless that margin is inside a guard
//     band; then the cell is re-evaluated in float64 with the exact operation order (rare branch),
//     so the in/out-of-bounds classification equals the float64 kernel's bit for bit.
// Requires isavalidstate box == grid end points (always true for GridDynamicSystem grids).
// `lsplit`: log2 of the lanes that share one state (small grids), actions interleaved over them.
// =================================================================================================
struct FastP {
    const float4* act;  // [A] {u0, u1, gu*dt, isavalidinput}
    float guard;        // guard band in cells
    int lsplit;
};

template <int DYN, typename PI_T, bool UNIFORM>
__global__ __launch_bounds__(256) void k_sweep_fast(DevP P, FastP F, const float* __restrict__ Jin,
                                                    float* __restrict__ Jout, PI_T* __restrict__ pi, float alpha,
                                                    SweepCtl sc, const float4* __restrict__ actp) {
    using D = Dyn<DYN>;
    constexpr int DOF = D::DOF, N = 2 * DOF, M = D::M, NP = 1 << DOF;
    if (sc.ctrl->done) return;
    const int split = UNIFORM ? 1 : (1 << F.lsplit);
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long o = UNIFORM ? t : (t >> F.lsplit);
    const int part = UNIFORM ? 0 : (int)(t & (split - 1));
    const long long owned = (long long)(P.row_end - P.row_begin) * P.plane;
    const bool live = o < owned;
    const float INF_F = (float)P.INF;

    int idx[N];
    int bp[NP];
    float wp[NP], ta[DOF], tB[DOF][M], selff[DOF], nm1f[DOF];
    int vstr[DOF], vdim[DOF];
    bool pos_in = false, on_target = false;
    float gxdt = 0.f;
    long long self = 0;
    if (live) {
        decode_node<N>(P, o, idx);
        double x[N], dx[N];
        self = (long long)(idx[0] - P.store_begin) * P.strd[0];
#pragma unroll
        for (int d = 0; d < N; ++d) {
            x[d] = P.lev[d][idx[d]];
            dx[d] = x[d] - P.xbar[d];
            if (d > 0) self += idx[d] * P.strd[d];
        }
        const double gx = quad_form<N>(P.Q, dx);
        on_target = P.ontarget && (l2norm<N>(dx) < P.EPS);
        gxdt = (float)(gx * P.dt);
        // position rows (exact, float64)
        pos_in = true;
        int ci[DOF];
        float yp[DOF];
#pragma unroll
        for (int i = 0; i < DOF; ++i) {
            const double xn = x[DOF + i] * P.dt + x[i];
            pos_in = pos_in && !(xn < P.glo[i]) && !(xn > P.ghi[i]);
            ci[i] = find_interval(P.lev[i], P.dim[i], P.glo[i], P.inv_step[i], xn);
            yp[i] = (float)((xn - P.lev[i][ci[i]]) / (P.lev[i][ci[i] + 1] - P.lev[i][ci[i]]));
        }
        if (pos_in) {
            int r0 = ci[0];
            if (r0 < P.store_begin || r0 + 1 >= P.store_end) {
                atomicOr(&sc.ctrl->halo_err, 1);
                r0 = min(max(r0, P.store_begin), P.store_end - 2);
            }
            ci[0] = r0 - P.store_begin;
        } else {
#pragma unroll
            for (int i = 0; i < DOF; ++i) ci[i] = 0;
        }
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            int b = 0;
            float w = 1.f;
#pragma unroll
            for (int i = 0; i < DOF; ++i) {
                const int bit = (c >> (DOF - 1 - i)) & 1;
                b += (ci[i] + bit) * (int)P.strd[i];
                w *= bit ? yp[i] : (1.f - yp[i]);
            }
            bp[c] = b;
            wp[c] = w;
        }
        double tr[8], a64[DOF], B64[DOF][M];
        D::trig_from_tables(P, idx, tr);
        D dyn;
        dyn.init(P.c, x, tr);
        dyn.affine(a64, B64);
#pragma unroll
        for (int i = 0; i < DOF; ++i) {
            const double sc = P.dt * P.inv_step[DOF + i];
            ta[i] = (float)(a64[i] * sc);
#pragma unroll
            for (int k = 0; k < M; ++k) tB[i][k] = (float)(B64[i][k] * sc);
            selff[i] = (float)idx[DOF + i];
            nm1f[i] = (float)(P.dim[DOF + i] - 1 - idx[DOF + i]);
            vstr[i] = (int)P.strd[DOF + i];
            vdim[i] = P.dim[DOF + i];
        }
    }

    float best = INFINITY;
    int arg = 0x7fffffff;
    if (live) {
        for (int a = part; a < P.A; a += split) {
            const float4 act = actp[a];  // (a restrict kernel argument: scalar loads when a is wave-uniform)
            float rel[DOF], m = INFINITY;
#pragma unroll
            for (int i = 0; i < DOF; ++i) {
                float r = fmaf(tB[i][0], act.x, ta[i]);
                if (M == 2) r = fmaf(tB[i][M - 1], act.y, r);
                rel[i] = r;
                m = fminf(m, fminf(r + selff[i], nm1f[i] - r));
            }
            const bool aok = act.w != 0.f;
            bool inb = pos_in && (m >= 0.f);
            if (pos_in && fabsf(m) < F.guard) {
                // rare: within the guard band of a bound -> exact float64 classification
                double x[N], tr[8], u[M], acc[DOF];
#pragma unroll
                for (int d = 0; d < N; ++d) x[d] = P.lev[d][idx[d]];
                D::trig_from_tables(P, idx, tr);
                D dyn;
                dyn.init(P.c, x, tr);
#pragma unroll
                for (int k = 0; k < M; ++k) u[k] = P.utab[a * M + k];
                dyn.accel(u, acc);
                inb = true;
#pragma unroll
                for (int i = 0; i < DOF; ++i) {
                    const double xn = acc[i] * P.dt + x[DOF + i];
                    inb = inb && !(xn < P.glo[DOF + i]) && !(xn > P.ghi[DOF + i]);
                }
            }
            float Jn = 0.f;
            if (inb) {
                int off = 0;
                float yv[DOF];
#pragma unroll
                for (int i = 0; i < DOF; ++i) {
                    const float fl = floorf(rel[i]);
                    int di = (int)fl;
                    const int ii = min(max(idx[DOF + i] + di, 0), vdim[i] - 2);
                    di = ii - idx[DOF + i];
                    yv[i] = fminf(fmaxf(rel[i] - (float)di, 0.f), 1.f);
                    off += ii * vstr[i];
                }
                float sv[NP];
#pragma unroll
                for (int v = 0; v < NP; ++v) {
                    int vo = off;
#pragma unroll
                    for (int i = 0; i < DOF; ++i) vo += ((v >> (DOF - 1 - i)) & 1) ? vstr[i] : 0;
                    float acc = wp[0] * Jin[bp[0] + vo];
#pragma unroll
                    for (int c = 1; c < NP; ++c) acc = fmaf(wp[c], Jin[bp[c] + vo], acc);
                    sv[v] = acc;
                }
#pragma unroll
                for (int i = DOF - 1; i >= 0; --i) {
#pragma unroll
                    for (int k = 0; k < (1 << i); ++k) sv[k] = fmaf(yv[i], sv[2 * k + 1] - sv[2 * k], sv[2 * k]);
                }
                Jn = sv[0];
            }
            const float G = (inb && aok) ? (on_target ? 0.f : gxdt + act.z) : INF_F;
            const float q = fmaf(alpha, Jn, G);
            if (q < best) {  // strict: keeps the first (smallest a) minimum within this lane
                best = q;
                arg = a;
            }
        }
    }
    if (!UNIFORM) {
        for (int off = split >> 1; off > 0; off >>= 1) {
            const float q2 = __shfl_xor(best, off, 64);
            const int a2 = __shfl_xor(arg, off, 64);
            if (q2 < best || (q2 == best && a2 < arg)) {
                best = q2;
                arg = a2;
            }
        }
    }
    double st_j = -INFINITY, st_dmax = -INFINITY, st_ndmin = -INFINITY;
    if (live && part == 0) {
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


#include "sweep_lean.inc"
#include "sweep_lean4.inc"

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
// host side
// =================================================================================================
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                                    \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(e_ == hipErrorOutOfMemory ? PVI_ENOMEM : PVI_EHIP, "%s failed: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                     \
    } while (0)

// dynamic-LDS ceiling set on the windowed kernels: a per-FUNCTION attribute, so it is always raised to the device
// maximum (160 KiB per CU on gfx950) -- a per-handle value would be lowered by the next handle with a smaller window
#define PVI_LDS_MAX (160 * 1024)

static inline unsigned grid_for(long long n, int block = 256) { return (unsigned)((n + block - 1) / block); }

static const int MAX_BATCH = 1024;  // sweeps per device-side batch (stats slots)

// ---- variant overrides (pvi_override) ---------------------------------------------------------------------------------
// pvi_create picks between kernel variants that compute the same recursion (tile shapes, lanes per node, nodes per
// thread, wave mappings, dense / sparse walks, which float32 path) by heuristics and timed sweeps.  Tests and profiling
// passes need to pin a variant; they do so through pvi_override(key, value) -- an explicit call, process-wide, read
// when a handle is created.  The ENVIRONMENT is never consulted: no environment variable changes what the library
// computes.  Only the keys below exist; every one selects among product variants whose results agree (bit for bit
// within a dtype path, within the float32 tolerance across float32 paths).
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
    "MULTI",       // 0: one launch per sweep also where a batch could run as ONE multi-sweep launch (k_sweep64m)
};
static std::vector<std::pair<std::string, std::string>> g_overrides;
static std::mutex g_override_mu;

// value of an override or NULL.  The string is a thread-local COPY taken under the lock (one slot per key, so several
// values can be held at once): a concurrent pvi_override cannot pull it from under the reader.
static const char* ovr(const char* key) {
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
static inline bool ovr_is(const char* key, int v) {
    const char* e = ovr(key);
    return e && atoi(e) == v;
}

struct pvi_problem {
    pvi_desc d;
    DevP P;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = 0.f;
    long long plane = 0, stored = 0, owned = 0;
    int A = 0, pi_size = 1;
    void* J[2] = {nullptr, nullptr};
    bool own_J = true, own_pi = true;
    void* pi = nullptr;
    int cur = 0;  // index of the current cost-to-go buffer
    std::vector<void*> dev_allocs;
    Ctrl* ctrl = nullptr;
    unsigned long long* slots = nullptr;
    double* results = nullptr;
    double* d_xnext = nullptr;  // tier B tables
    double* d_G = nullptr;
    unsigned char* d_ok = nullptr;  // tier B base-class semantics (NULL: LUT semantics)
    void* d_pack = nullptr;         // tier B, float32 handles: packed records (TabRec<n>), see k_table_pack
    bool packed = false;
    double* stage = nullptr;  // f64 staging for up/download
    long long stage_n = 0;
    FastP F;                  // f32 fast path tables
    bool fast_ok = false;
    Lean4P L4;                // f32 lean path of 4-D grids (sweep_lean4.inc)
    bool lean4_ok = false;
    int lean4_block = 512, lean4_rsk = 0, lean4_bands = 1, lean4_tables = 0;
    void* lean4_tiles = nullptr;  // [grid] Lean4Tile, launch order
    int lean4_stage = 2;          // actions whose gathers are in flight together (sweep_lean4.inc)
    int lean4_ptab_inv = 0;       // bit 0 / 1: the (position node, action) table does not depend on axis 0 / 1
    long long lean4_ptab_groups = 0;
    char lean4_choice[32] = "-";  // rows cap / threads / widest tile of the tiling in use (pvi_override L4PIN takes it back)
    char lean4_cands[960] = "";  // the timed tile shapes of set-up: rows x columns : ms
    unsigned lean4_grid = 0;
    size_t lean4_lds = 0;
    LeanP LP;                 // f32 lean path (sweep_lean.inc)
    bool lean_ok = false;
    int lean_pw1 = 2, lean_block = 256;
    dim3 lean_grid;
    size_t lean_lds = 0;
    bool lean_lds_attr = false;
    char lean_why[160] = "";
    int lean_reach = 0;       // largest |velocity displacement| of an in-box cell, grid cells
    int lean_opmag = 0;       // largest |ta| + sum |tB u| (cells): operand magnitude of the float32 displacement
    const int* aok32 = nullptr;  // isavalidinput per action as int32 (scalar loads in the exact kernel)
    const Act64* act64 = nullptr;   // float64 second form (k_sweep64): per-action records, {level, reciprocal} tables
    const double2* levr = nullptr;
    bool force_exact = false; // pvi_self_check: route the next launch to the plain-gather kernel k_sweep
    bool use64 = false;
    const uint4* vmask = nullptr;  // SPARSE float64 sweep: 128 validity bits per owned node (k_valid_mask)
    int sparse64 = 0;         // 1: every lane walks the set bits of its validity mask instead of all A actions
    double infrac64 = -1.0;   // share of the (node, action) cells that land in the box (4-D float64 handles)
    int patch64 = -1;         // 4-D wave mapping of k_sweep64: 1 = 8 x 8 velocity patches, 0 = consecutive nodes (timed at set-up)
    size_t levr_bytes = 0;
    const unsigned long long* okmask3 = nullptr;  // fast3: validity of every (node, action) cell of an explicit system
    const double* roll_params = nullptr;  // constants of the continuous closed form (pvi_set_rollout_params)
    SplineP SP;               // bicubic-spline interpolation mode (sweep_spline.inc)
    bool spline = false;
    int multi64 = -1;         // multi-sweep launch of the float64 sweep (k_sweep64m): -1 not decided, 0 no, 1 yes
    char multi_why[96] = "";
    int regtab64 = -1;            // the multi-sweep launches keep the per-action cells in registers (2-D, <= 12 actions)
    char kname[128] = "";     // the sweep kernel of the last launch, as a kernel trace prints it (spaces removed): pvi_describe `kernel=`
};

// Name of a kernel template instantiation the way the demangler (rocprofv3's kernel trace) prints it, without spaces:
// "k_sweep64<3,unsignedchar,true,true,true>".  Recorded at every sweep launch, reported by pvi_describe, so that counter
// passes and trace summaries can be tied to the kernel a handle really runs (tools/make_counters_json.py, bench.py).
template <typename T> static const char* tname();
template <> const char* tname<float>() { return "float"; }
template <> const char* tname<double>() { return "double"; }
template <> const char* tname<unsigned char>() { return "unsignedchar"; }
template <> const char* tname<unsigned short>() { return "unsignedshort"; }
static inline void kn_arg(std::string& s, int v) { s += std::to_string(v); }
static inline void kn_arg(std::string& s, bool v) { s += v ? "true" : "false"; }
static inline void kn_arg(std::string& s, const char* v) { s += v; }
template <typename... T>
static void set_kname(pvi_problem* h, const char* base, T... args) {
    std::string s(base);
    s += "<";
    bool first = true;
    ((s += first ? "" : ",", first = false, kn_arg(s, args)), ...);
    s += ">";
    snprintf(h->kname, sizeof(h->kname), "%s", s.c_str());
}

template <typename T>
static int dev_upload(pvi_problem* h, const T* src, size_t n, const T** out) {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, n * sizeof(T) > 0 ? n * sizeof(T) : 8));
    h->dev_allocs.push_back(p);
    if (n) HIPCHK(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)p;
    return PVI_OK;
}

// host twin of quad_form (same operation order, this TU is built with -ffp-contract=off)
static double quad_form_host(const double* M, const double* dx, int n) {
    double out = 0.0;
    for (int i = 0; i < n; ++i) {
        double row = M[i * n] * dx[0];
        for (int j = 1; j < n; ++j) row = row + M[i * n + j] * dx[j];
        const double term = dx[i] * row;
        out = (i == 0) ? term : out + term;
    }
    return out;
}


// ---- lean path set-up (sweep_lean.inc): per-node coefficients, pair tables, per-tile windows -------------
static inline unsigned magic32(unsigned d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d); }

template <typename T>
static int dev_alloc(pvi_problem* h, size_t n, T** out) {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, (n ? n : 1) * sizeof(T)));
    h->dev_allocs.push_back(p);
    *out = (T*)p;
    return PVI_OK;
}

static void dev_release(pvi_problem* h, void* p) {
    if (!p) return;
    for (auto& q : h->dev_allocs)
        if (q == p) q = nullptr;
    (void)hipFree(p);
}

static int launch_sweep(pvi_problem* h, int src, double alpha, hipStream_t st, int k, double tol, int deferred = 0);

static int lean_try(pvi_problem* h, int tv0_t, int tv1_t, int lds_budget_floats) {
    const DevP& P = h->P;
    LeanP& L = h->LP;
    const int DOF = P.dof;
    // 4-D: tiles of the (i2, i3) velocity plane, one (i0, i1) position node per workgroup;
    // 2-D: tiles of the (i0, i1) grid itself (TV0 rows share most of their window rows)
    L.dof1 = DOF == 1;
    L.V0 = DOF == 2 ? P.dim[2] : (P.row_end - P.row_begin);
    L.V1 = P.dim[P.n - 1];
    L.ntx = (L.V1 + tv1_t - 1) / tv1_t;
    L.TV1 = ovr("TV_EXACT") ? tv1_t : (L.V1 + L.ntx - 1) / L.ntx;
    const int tv0 = std::max(1, std::min(L.V0, tv0_t));
    L.nty = (L.V0 + tv0 - 1) / tv0;
    L.TV0 = (L.V0 + L.nty - 1) / L.nty;
    L.tv1_magic = ((1 << 20) + L.TV1 - 1) / L.TV1;
    L.half = ((L.TV0 + L.npt - 1) / L.npt) * L.TV1;  // nodes per band: thread t owns tile nodes t + k * half, k < npt
    L.posdim1 = DOF == 2 ? P.dim[1] : 1;
    L.pd_magic = magic32((unsigned)L.posdim1);
    L.vplane = (long long)L.V0 * L.V1;
    L.owned = h->owned;
    const int pnodes = DOF == 2 ? (P.row_end - P.row_begin) * L.posdim1 : 1;
    const long long ntiles = (long long)L.ntx * L.nty * pnodes;
    if (ntiles >= 0x7fffffffLL) return 1;
    L.ntx_magic = magic32((unsigned)L.ntx);
    L.ntxy_magic = magic32((unsigned)(L.ntx * L.nty));
    L.nblocks = (unsigned)ntiles;
    L.xq = L.nblocks / 8u;
    L.xrem = L.nblocks % 8u;
    L.xcd_remap = ovr("NO_XCD") ? 0 : 1;
    h->lean_grid = dim3((unsigned)ntiles, 1, 1);
    int rc;
    if (L.win) dev_release(h, L.win);
    if (L.tbt) dev_release(h, L.tbt);
    L.win = nullptr;
    L.tbt = nullptr;
    L.tb_tile = 0;
    if ((rc = dev_alloc(h, (size_t)ntiles * 8, &L.win))) return rc;
    if ((rc = dev_alloc(h, (size_t)ntiles * 4, &L.tbt))) return rc;
    hipLaunchKernelGGL(k_lean_winit, grid_for(std::max<long long>(ntiles * 4, 4)), 256, 0, h->stream, L.win, ntiles,
                       L.summary);
    const int sthreads = ((L.TV0 * L.TV1 + 63) / 64) * 64;
    if (sthreads > 1024) return 1;
    switch (h->d.dynamics_id) {
        case PVI_DYN_PENDULUM:
            hipLaunchKernelGGL((k_lean_setup<PVI_DYN_PENDULUM>), h->lean_grid, sthreads, 0, h->stream, P, L, h->F.act);
            break;
        case PVI_DYN_CARTPOLE:
            hipLaunchKernelGGL((k_lean_setup<PVI_DYN_CARTPOLE>), h->lean_grid, sthreads, 0, h->stream, P, L, h->F.act);
            break;
        case PVI_DYN_NODE_1x1:
            hipLaunchKernelGGL((k_lean_setup<PVI_DYN_NODE_1x1>), h->lean_grid, sthreads, 0, h->stream, P, L, h->F.act);
            break;
        case PVI_DYN_NODE_2x1:
            hipLaunchKernelGGL((k_lean_setup<PVI_DYN_NODE_2x1>), h->lean_grid, sthreads, 0, h->stream, P, L, h->F.act);
            break;
        case PVI_DYN_NODE_2x2:
            hipLaunchKernelGGL((k_lean_setup<PVI_DYN_NODE_2x2>), h->lean_grid, sthreads, 0, h->stream, P, L, h->F.act);
            break;
        default:
            hipLaunchKernelGGL((k_lean_setup<PVI_DYN_TWOLINK>), h->lean_grid, sthreads, 0, h->stream, P, L, h->F.act);
            break;
    }
    HIPCHK(hipGetLastError());
    int summary[8];
    HIPCHK(hipMemcpyAsync(summary, L.summary, sizeof(summary), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->lean_reach = summary[4];
    h->lean_opmag = summary[5];
    if (summary[3]) {
        snprintf(h->lean_why, sizeof(h->lean_why), "%s", (summary[3] & 2) ? "an action fails isavalidinput" : "halo too small");
        return (summary[3] & 1) ? 2 : 1;
    }
    // row pitch: >= longest row + 1 (the j+1 corner), odd so that rows start on different banks.
    // (A pitch congruent to the tile width modulo 32 -- consecutive lanes on consecutive banks across
    // tile rows -- was measured: fewer conflict cycles per LDS instruction, but the larger pitch costs
    // LDS capacity and one more address add per corner; no net gain.  PVI_RS_MODE=1 selects it.)
    L.tb_tile = (summary[2] == 0 && !ovr("NO_TBTILE")) ? 1 : 0;
    int rs = (summary[1] + 1) | 1;
    // 2-D pair windows are a few KB: pitches 64 and 128 have their own kernels (row + 1 is an immediate offset of the read)
    if (DOF == 1 && !ovr("NO_RS64")) rs = rs <= 64 ? 64 : (rs <= 128 ? 128 : rs);
    // 16-byte window DMA (4-D; J buffers with slack behind them): rows are packed with a pitch that is a multiple of
    // 4 floats, one instruction then moves 256 consecutive window floats (about four rows).  PVI_DMA16=0: 4-byte DMA.
    L.dma16 = (DOF == 2 && (h->own_J || (h->d.flags & PVI_FLAG_EXT_J_SLACK)) &&
               !(ovr("DMA16") && !atoi(ovr("DMA16")))) ? 1 : 0;
    if (L.dma16) {
        rs = (summary[1] + 3) & ~3;
        // pitch 64 has its own kernel (the next velocity row is an immediate offset of the LDS read): rows a little
        // shorter are padded to it when the window still fits
        if (rs > 48 && rs < 64 && (long long)summary[0] * 64 + 128 <= lds_budget_floats && !ovr("NO_RS64")) rs = 64;
        L.rs_magic = magic32((unsigned)rs);
    }
    // (2-D windows are staged as pairs -- 8 bytes per column, sweep_lean.inc lds_corners -- 4-D windows as single floats)
    const long long need = (long long)summary[0] * rs * (DOF == 1 ? 2 : 1) + 128;
    if (need <= lds_budget_floats) {
        L.RS = rs;
        h->lean_pw1 = rs;
        L.lds_floats = (int)((need + 3) & ~3ll);
        h->lean_lds = (size_t)L.lds_floats * 4;
        return 0;
    }
    snprintf(h->lean_why, sizeof(h->lean_why), "tile %dx%d needs %lld LDS floats (%d rows x pitch %d; budget %d)", L.TV0,
             L.TV1, need, summary[0], rs, lds_budget_floats);
    return 1;
}

// =====================================================================================================================
// lean path of 4-D grids (sweep_lean4.inc): set-up tables, tiling candidates, launch schedule
// =====================================================================================================================

template <typename PI_T>
static int launch_lean4_t(pvi_problem* h, const float* Jin, float* Jout, float alpha, hipStream_t st, SweepCtl sc, bool probe = false) {
    const Lean4P& L = h->L4;
    sc.nblocks = h->lean4_grid;
    sc.split_finish = 1;
    PI_T* pi = (PI_T*)h->pi;
#define L4K(KFN)                                                                                                       \
    {                                                                                                                  \
        auto kfn = KFN;                                                                                                \
        if (h->lean4_lds > 48 * 1024)                                                                                  \
            HIPCHK(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, PVI_LDS_MAX));    \
        hipLaunchKernelGGL(kfn, dim3(h->lean4_grid), dim3(h->lean4_block), h->lean4_lds, st, h->P, L, Jin, Jout, pi, alpha, \
                           sc, (const float*)L.ptab, (const Lean4Tile*)h->lean4_tiles);                                \
    }
#define L4(DYN)                                   \
    if (probe)                                    \
        L4K((k_sweep_lean4_probe<DYN, PI_T>))     \
    else {                                        \
        set_kname(h, "k_sweep_lean4", (int)DYN, tname<PI_T>()); \
        L4K((k_sweep_lean4<DYN, PI_T>))           \
    }
    switch (h->d.dynamics_id) {
        case PVI_DYN_CARTPOLE: L4(PVI_DYN_CARTPOLE) break;
        case PVI_DYN_NODE_2x1: L4(PVI_DYN_NODE_2x1) break;
        case PVI_DYN_NODE_2x2: L4(PVI_DYN_NODE_2x2) break;
        default: L4(PVI_DYN_TWOLINK) break;
    }
#undef L4
#undef L4K
    hipLaunchKernelGGL(k_sweep_finish, 1, STAT_SHARDS, 0, st, sc);
    HIPCHK(hipGetLastError());
    return PVI_OK;
}

// Row pieces of axis 2 for ONE row i0 of axis 0: cut where the axis-0 corner index of the position row steps (so that a
// tile sees one pair plane), then into near-equal parts of at most `cap` rows.  Per i0, because a step that falls on a level
// exactly (v dt / dx an integer) lands one row earlier or later depending on the rounding of x0 + v dt for that i0: one
// segmentation for all rows would have to cut on both sides and leave one-row pieces (measured: 21 pieces instead of 16 on
// C3, 63 % of the lanes live).
static void lean4_row_pieces(const std::vector<int2>& pt0, int i0, int V0, int cap, std::vector<int2>& out) {
    out.clear();
    int a = 0;
    for (int j = 1; j <= V0; ++j) {
        if (j < V0) {
            const int2 p = pt0[(size_t)i0 * V0 + j - 1], q = pt0[(size_t)i0 * V0 + j];
            if ((p.x < 0 ? -0x40000000 : p.x) == (q.x < 0 ? -0x40000000 : q.x)) continue;
        }
        const int len = j - a, n = (len + cap - 1) / cap;
        for (int k = 0; k < n; ++k) {
            const int s0 = a + (int)((long long)len * k / n), s1 = a + (int)((long long)len * (k + 1) / n);
            out.push_back(make_int2(s0, s1 - s0));
        }
        a = j;
    }
}
// Tiles of the velocity plane for ONE row i0: every row piece is split into near-equal column chunks as wide as the
// workgroup allows (rows x columns <= threads), so that short pieces get wide tiles and every workgroup is about full
// (with one column split for all pieces 20 % of the lanes were idle on C3).  `wmax` bounds the width: the window of a wide tile
// of a far-reaching system (C4: 32 pairs of reach along axis 3) may cost a workgroup of occupancy.
static void lean4_row_tiles(const std::vector<int2>& pt0, int i0, int V0, int V1, int cap, int threads, int wmax, std::vector<int4>& out) {
    std::vector<int2> pieces;
    lean4_row_pieces(pt0, i0, V0, cap, pieces);
    out.clear();
    for (const int2& pc : pieces) {
        const int w = std::max(1, std::min(std::min(V1, wmax), threads / pc.y));
        const int n = (V1 + w - 1) / w;
        for (int k = 0; k < n; ++k) {
            const int c0 = (int)((long long)V1 * k / n), c1 = (int)((long long)V1 * (k + 1) / n);
            out.push_back(make_int4(pc.x, pc.y, c0, c1 - c0));
        }
    }
}

// Launch order.  Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8), each with a private 4 MiB L2.
// XCD x sweeps ITS chunk of axis 1 for every owned row of axis 0 in turn (bands of the row's tile list outermost, so that
// the planes of three consecutive axis-0 rows over the chunk fit its L2): the axis-0 planes a tile gathers from were
// fetched for the previous row a moment ago.  Lists are interleaved into physical order and padded to equal length.
static void lean4_schedule(int R, int N1, int ntr, int nbands, std::vector<unsigned>& out, int r0 = 0) {
    // Round 4: an XCD's share of a row is a contiguous EIGHTH of the row's (axis-1 index, tile) list -- not a whole number of
    // axis-1 indices.  Splitting by index gave 13, 13, 12, 13, ... of C3's 101 to the XCDs: the lists were padded to the
    // longest with empty workgroups and five XCDs idled 8 % of every row (the plain order, which balances by construction,
    // ran 2.90 ms against 3.10 ms with the same tiles; profiles/r04_launch_order.log).
    std::vector<std::vector<unsigned>> lists(8);
    for (int b = 0; b < nbands; ++b) {
        const int k0 = (int)((long long)ntr * b / nbands), k1 = (int)((long long)ntr * (b + 1) / nbands), kb = k1 - k0;
        const long long E = (long long)N1 * kb;  // entries of one row of axis 0 in this band: i1-major, tile-minor
        for (int r = r0; r < r0 + R; ++r)  // (r0, R: the rows of a timed candidate; the whole slab otherwise)
            for (int x = 0; x < 8; ++x) {
                const long long e0 = E * x / 8, e1 = E * (x + 1) / 8;
                for (long long e = e0; e < e1; ++e) {
                    const int i1 = (int)(e / kb), k = k0 + (int)(e - (long long)i1 * kb);
                    lists[x].push_back((unsigned)((long long)(r * N1 + i1) * ntr + k));
                }
            }
    }
    size_t mx = 0;
    for (auto& l : lists) mx = std::max(mx, l.size());
    out.assign(8 * mx, 0xffffffffu);
    for (int x = 0; x < 8; ++x)
        for (size_t j = 0; j < lists[x].size(); ++j) out[8 * j + x] = lists[x][j];
}

struct Lean4Cand {
    int cap, w, wmax;  // rows cap, workgroup threads, widest tile
};
// the tiling a create of this process chose for a problem shape (device, dynamics, dims, actions, rows, dt, velocity box):
// a second handle of the same shape -- the float32 / float64 pair of a convergence check, the pieces of a shard, a bench
// that builds its workload twice -- takes it without timing anything (pvi_override("TUNE", "2") times again)
static std::map<std::string, Lean4Cand> g_lean4_choice;
static std::mutex g_lean4_choice_mu;

// one candidate tiling: tile lists, window boxes, row pitch, schedule.  rc 0 = usable, 1 = does not fit, < 0 error.
static int lean4_try(pvi_problem* h, const std::vector<int2>& pt0, int cap, int threads, int wmax, size_t lds_budget, int* narrower = nullptr,
                     int sub_r0 = 0, int sub_rows = -1) {
    const DevP& P = h->P;
    Lean4P& L = h->L4;
    const int rows = P.row_end - P.row_begin;
    if (threads > 512 || threads < 64 || (threads & 63)) return 1;
    std::vector<std::vector<int4>> per((size_t)rows);
    int ntr = 1, tv0 = 1, tv1 = 1;
    for (int r = 0; r < rows; ++r) {
        lean4_row_tiles(pt0, P.row_begin + r, P.dim[2], P.dim[3], cap, threads, wmax, per[(size_t)r]);
        ntr = std::max(ntr, (int)per[(size_t)r].size());
    }
    for (auto& v : per)
        for (auto& t : v) {
            tv0 = std::max(tv0, t.y);
            tv1 = std::max(tv1, t.w);
        }
    const size_t nlists = per.size();
    std::vector<int4> tlist(nlists * ntr, make_int4(0, 0, 0, 0));
    for (size_t r = 0; r < nlists; ++r)
        for (size_t k = 0; k < per[r].size(); ++k) tlist[r * ntr + k] = per[r][k];
    L.V0 = P.dim[2];
    L.V1 = P.dim[3];
    L.TV0 = tv0;
    L.TV1 = tv1;
    L.ntr = ntr;
    L.posdim1 = P.dim[1];
    L.pd_magic = magic32((unsigned)L.posdim1);
    L.ntr_magic = magic32((unsigned)ntr);
    L.vplane = (long long)L.V0 * L.V1;
    L.owned = h->owned;
    const long long npos = (long long)rows * P.dim[1], ntiles = npos * ntr;
    if (ntiles >= 0x7fffffffLL / 8) return 1;
    int rc;
    if (L.tlist) dev_release(h, (void*)L.tlist);
    if (L.win) dev_release(h, L.win);
    if (L.sched) dev_release(h, (void*)L.sched);
    L.tlist = nullptr;
    L.win = nullptr;
    L.sched = nullptr;
    int4* d_tl = nullptr;
    if ((rc = dev_alloc(h, tlist.size(), &d_tl))) return rc;
    HIPCHK(hipMemcpyAsync(d_tl, tlist.data(), tlist.size() * sizeof(int4), hipMemcpyHostToDevice, h->stream));
    L.tlist = d_tl;
    if ((rc = dev_alloc(h, (size_t)ntiles * 8, &L.win))) return rc;
    HIPCHK(hipMemsetAsync(L.summary, 0, 6 * sizeof(int), h->stream));
    hipLaunchKernelGGL(k_lean4_tiles, dim3((unsigned)ntiles), dim3(threads), 0, h->stream, P, L, ntiles);
    HIPCHK(hipGetLastError());
    int summary[8];
    HIPCHK(hipMemcpyAsync(summary, L.summary, sizeof(summary), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));  // (also: `tlist` has been copied)
    // Row pitch in slots: the fill writes whole groups of four columns.  (Tiles differ in width, so no pitch is congruent to
    // all of them; tools/ldsgather.hip: a row wrap or a displacement step inside a wave costs a ds_read_b64 4.3 -> 5.0 clk at
    // worst, whatever the pitch -- the 4.4 clk of the conflict-free read is what counts.)
    int rs = std::max(4, (summary[1] + 3) & ~3);
    size_t lds = (size_t)std::max(summary[0], 1) * (size_t)rs * 8 + 256;
    if (narrower) {  // the widest tile whose window lets as many workgroups onto a CU as the register budget does (24 waves)
        *narrower = 0;
        const size_t room = (size_t)160 * 1024 / (size_t)std::max(1, 1536 / threads) - 512;
        if (lds > room && summary[0] > 0) {
            const int rs_fit = (int)((room - 256) / 8 / (size_t)summary[0]) & ~3, w_fit = tv1 - (rs - rs_fit);
            if (w_fit >= 12 && w_fit < tv1) *narrower = w_fit;
        }
    }
    if (lds > lds_budget) {
        snprintf(h->lean_why, sizeof(h->lean_why), "tiles of %d threads, <= %d rows need %zu LDS bytes (%d window rows x %d pairs; budget %zu)",
                 threads, cap, lds, summary[0], rs, lds_budget);
        return 1;
    }
    L.RS = rs;
    h->lean4_rsk = 0;
    h->lean4_lds = lds;
    h->lean4_block = threads;
    hipLaunchKernelGGL(k_lean4_off, grid_for(h->lean4_ptab_groups * 4), 256, 0, h->stream, L, h->lean4_ptab_groups);
    // bands of the tile list: an XCD's working set is three axis-0 rows x (its share of axis 1 + position reach) x band rows x
    // V1 floats.  Round 3 sized the bands for 1.5 MB of the 4 MB L2 (C3: 2 bands, C4: 3); measured on the balanced schedule
    // of round 4, ONE band is fastest on both (C3 2.77 against 2.91 ms, C4 19.13 against 19.30 ms; profiles/r04_launch_order.log)
    // -- the sweep is not bound by what the bands save -- but the bytes it moves through the L2s are: C4 with 1 / 2 / 3 / 4 / 6
    // bands: 17.1 / 15.3 / 14.3 / 11.6 / 13.3 GB per launch at 19.1 / 18.9 / 19.0 / 18.9 / 19.1 ms
    // (profiles/r04_c4_bands_traffic.log).  2.2 MB of working set per band gives C3 its one band (2 MB) and C4 four.
    const int n1c = (P.dim[1] + 7) / 8 + 3;
    const double per_row = 3.0 * n1c * (double)L.V1 * 4.0;
    const int band_rows = std::max(8, (int)(2.2e6 / per_row) - (summary[0] ? 12 : 0));  // (bands are ranges of the tile LIST: no need to hold a whole tile's rows)
    int nbands = std::max(1, std::min(ntr, (L.V0 + band_rows - 1) / band_rows));
    if (ovr("BANDS") && atoi(ovr("BANDS")) > 0) nbands = std::min(ntr, atoi(ovr("BANDS")));  // (experiments: the launch order only)
    h->lean4_bands = nbands;
    std::vector<unsigned> sched;
    // (sub_rows: a timed candidate sweeps a few rows of axis 0 from the middle of the slab -- the tiling is the same for every
    //  row, so a tenth of the grid ranks the candidates at a tenth of the cost; window boxes and pitch are those of the slab)
    if (sub_rows > 0 && sub_rows < rows && !ovr_is("NO_XCD", 1))
        lean4_schedule(sub_rows, P.dim[1], ntr, nbands, sched, sub_r0);
    else
        lean4_schedule(rows, P.dim[1], ntr, ovr_is("NO_XCD", 1) ? 1 : nbands, sched);
    if (ovr_is("NO_XCD", 1)) {  // plain order (experiments, tests): tile ids ascending
        sched.resize((size_t)ntiles);
        for (long long t = 0; t < ntiles; ++t) sched[(size_t)t] = (unsigned)t;
    }
    unsigned* d_sched = nullptr;
    if ((rc = dev_alloc(h, sched.size(), &d_sched))) return rc;
    HIPCHK(hipMemcpy(d_sched, sched.data(), sched.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    L.sched = d_sched;
    h->lean4_grid = (unsigned)sched.size();
    if (h->lean4_tiles) dev_release(h, h->lean4_tiles);
    Lean4Tile* d_tiles = nullptr;
    if ((rc = dev_alloc(h, sched.size(), &d_tiles))) return rc;
    h->lean4_tiles = d_tiles;
    hipLaunchKernelGGL(k_lean4_desc, grid_for((long long)sched.size()), 256, 0, h->stream, P, L, (const unsigned*)d_sched,
                       (long long)sched.size(), d_tiles);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int pvi_plan_plane_tiles(int32_t V0, int32_t V1, const int32_t* corner0, int32_t cap, int32_t threads, int32_t wmax,
                                    int32_t* tiles4, int32_t max_tiles) {
    if (V0 < 1 || V1 < 1 || !corner0 || cap < 1 || wmax < 1 || (!tiles4 && max_tiles > 0) || max_tiles < 0)
        return fail(PVI_EINVAL, "pvi_plan_plane_tiles: bad argument");
    if (threads < 64 || threads > 512 || (threads & 63)) return fail(PVI_EINVAL, "threads must be 64 ... 512 in steps of 64");
    std::vector<int2> pt0((size_t)V0);
    for (int j = 0; j < V0; ++j) pt0[(size_t)j] = make_int2(corner0[j], 0);
    std::vector<int4> out;
    lean4_row_tiles(pt0, 0, V0, V1, cap, threads, wmax, out);
    for (size_t k = 0; k < out.size() && (int)k < max_tiles; ++k) {
        tiles4[4 * k] = out[k].x;
        tiles4[4 * k + 1] = out[k].y;
        tiles4[4 * k + 2] = out[k].z;
        tiles4[4 * k + 3] = out[k].w;
    }
    return (int)out.size();
}

extern "C" int64_t pvi_plan_schedule(int32_t rows, int32_t n1, int32_t tiles_per_plane, int32_t bands, uint32_t* out, int64_t max_blocks) {
    if (rows < 1 || n1 < 1 || tiles_per_plane < 1 || bands < 1 || bands > tiles_per_plane || (!out && max_blocks > 0) || max_blocks < 0)
        return fail(PVI_EINVAL, "pvi_plan_schedule: bad argument");
    if ((long long)rows * n1 * tiles_per_plane >= 0x7fffffffLL / 8) return fail(PVI_EINVAL, "pvi_plan_schedule: too many tiles");
    std::vector<unsigned> sched;
    lean4_schedule(rows, n1, tiles_per_plane, bands, sched);
    for (size_t k = 0; k < sched.size() && (long long)k < max_blocks; ++k) out[k] = sched[k];
    return (int64_t)sched.size();
}

static int lean4_setup(pvi_problem* h) {
    const DevP& P = h->P;
    Lean4P& L = h->L4;
    memset(&L, 0, sizeof(L));
    h->lean4_ok = false;
    if (!h->fast_ok || P.dof != 2 || ovr("NO_LEAN") || ovr_is("WIN", 0) || h->stored >= 0x7fffffffLL) return PVI_OK;
    if (!(h->own_J || (h->d.flags & PVI_FLAG_EXT_J_SLACK))) return PVI_OK;  // the 16-byte window loads may run 12 bytes past a row
    if (P.strd[0] * 16 >= (1LL << 31)) return PVI_OK;  // the window fill addresses planes by 32-bit byte offsets from the window origin
    const int rows = P.row_end - P.row_begin;
    const long long npos = (long long)rows * P.dim[1];
    int rc;
    L.owned = h->owned;
    h->lean4_stage = 2;
    L.ngroups = (P.A + 3) / 4;
    float2* tsp_node = nullptr;
    float* gx_node = nullptr;
    int2 *pt0 = nullptr, *pt1 = nullptr;
    if ((rc = dev_alloc(h, (size_t)h->owned, &L.flag))) return rc;
    if ((rc = dev_alloc(h, (size_t)P.dim[0] * P.dim[2], &pt0))) return rc;
    if ((rc = dev_alloc(h, (size_t)P.dim[1] * P.dim[3], &pt1))) return rc;
    if ((rc = dev_alloc(h, 8, &L.summary))) return rc;
    float* ptab_full = nullptr;
    if ((rc = dev_alloc(h, (size_t)npos * L.ngroups * 24, &ptab_full))) return rc;
    if ((rc = dev_alloc(h, (size_t)2 * h->owned, &tsp_node))) return rc;
    L.pt0 = pt0;
    L.pt1 = pt1;
    // g_x: a sum of per-axis terms when Q is diagonal (then no per-node array); TABLES=0 keeps the per-node arrays
    const bool want_tables = !ovr_is("TABLES", 0);
    bool diag = want_tables;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (i != j && h->d.Q[i * 4 + j] != 0.0) diag = false;
    if (diag) {
        for (int d = 0; d < 4; ++d) {
            std::vector<double> t((size_t)P.dim[d]);
            for (int k = 0; k < P.dim[d]; ++k) {
                const double dx = h->d.x_level[d][k] - h->d.xbar[d];
                // quad_form: row_d = sum_j Q[d][j] dx_j with the off-diagonal terms exactly zero, term = dx_d * row_d
                t[(size_t)k] = dx * (h->d.Q[d * 4 + d] * dx);
            }
            if ((rc = dev_upload(h, t.data(), t.size(), &L.gt[d]))) return rc;
        }
    } else {
        if ((rc = dev_alloc(h, (size_t)h->owned, &gx_node))) return rc;
        // (the sweep loads both forms of g_x without a branch and keeps the one that applies: one-word zero tables to read)
        const double zero = 0.0;
        for (int d = 0; d < 4; ++d)
            if ((rc = dev_upload(h, &zero, 1, &L.gt[d]))) return rc;
    }
    L.gx = gx_node;
    HIPCHK(hipMemsetAsync(L.summary, 0, 8 * sizeof(int), h->stream));
    hipLaunchKernelGGL(k_lean_pairs, grid_for((long long)P.dim[0] * P.dim[2]), 256, 0, h->stream, P, 0, pt0);
    hipLaunchKernelGGL(k_lean_pairs, grid_for((long long)P.dim[1] * P.dim[3]), 256, 0, h->stream, P, 1, pt1);
#define L4DISPATCH(MACRO)                                  \
    switch (h->d.dynamics_id) {                            \
        case PVI_DYN_CARTPOLE: MACRO(PVI_DYN_CARTPOLE) break; \
        case PVI_DYN_NODE_2x1: MACRO(PVI_DYN_NODE_2x1) break; \
        case PVI_DYN_NODE_2x2: MACRO(PVI_DYN_NODE_2x2) break; \
        default: MACRO(PVI_DYN_TWOLINK) break;                \
    }
#define L4PT(DYN) hipLaunchKernelGGL((k_lean4_ptab<DYN>), grid_for(npos * L.ngroups * 4), 256, 0, h->stream, P, L, ptab_full);
    L4DISPATCH(L4PT)
#undef L4PT
    // the (position node, action) table rarely depends on both position axes (cart-pole, two-link arm: H(q) depends on the
    // second joint only): keep it over the axes it does depend on -- a CU then finds its rows in the scalar cache
    L.pcs[0] = P.dim[1];
    L.pcs[1] = 1;
    L.ptab = ptab_full;
    h->lean4_ptab_inv = 0;
    if (want_tables) {
        const int all = 3;
        int inv = 0;
        HIPCHK(hipMemcpyAsync(L.summary + 7, &all, sizeof(int), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_lean4_ptab_inv, grid_for(npos * L.ngroups * 24), 256, 0, h->stream, P, L, (const float*)ptab_full, L.summary + 7);
        HIPCHK(hipMemcpyAsync(&inv, L.summary + 7, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (inv) {
            const int n0 = (inv & 1) ? 1 : rows, n1 = (inv & 2) ? 1 : P.dim[1];
            float* compact = nullptr;
            if ((rc = dev_alloc(h, (size_t)n0 * n1 * L.ngroups * 24, &compact))) return rc;
            hipLaunchKernelGGL(k_lean4_ptab_compact, grid_for((long long)n0 * n1 * L.ngroups * 24), 256, 0, h->stream, P, L,
                               (const float*)ptab_full, compact, n0, n1);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(h->stream));
            dev_release(h, ptab_full);
            L.ptab = compact;
            L.pcs[0] = (inv & 1) ? 0 : n1;
            L.pcs[1] = (inv & 2) ? 0 : 1;
            h->lean4_ptab_inv = inv;
        }
    }
    h->lean4_ptab_groups = (long long)((h->lean4_ptab_inv & 1) ? 1 : rows) * ((h->lean4_ptab_inv & 2) ? 1 : P.dim[1]) * L.ngroups;
#define L4ND(DYN) hipLaunchKernelGGL((k_lean4_node<DYN>), grid_for(h->owned), 256, 0, h->stream, P, L, tsp_node, gx_node);
    L4DISPATCH(L4ND)
#undef L4ND
#undef L4DISPATCH
    HIPCHK(hipGetLastError());
    int summary[8];
    HIPCHK(hipMemcpyAsync(summary, L.summary, sizeof(summary), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    auto give_up = [&](const char* why) {
        snprintf(h->lean_why, sizeof(h->lean_why), "%s", why);
        dev_release(h, L.flag); dev_release(h, pt0); dev_release(h, pt1); dev_release(h, L.summary); dev_release(h, L.ptab);
        dev_release(h, tsp_node); dev_release(h, gx_node);
        if (L.tsp && L.tsp != tsp_node) dev_release(h, (void*)L.tsp);
        if (L.tlist) dev_release(h, (void*)L.tlist);
        if (L.win) dev_release(h, L.win);
        if (L.sched) dev_release(h, (void*)L.sched);
        for (int d = 0; d < 4; ++d) dev_release(h, (void*)L.gt[d]);
        memset(&L, 0, sizeof(L));
        return PVI_OK;
    };
    if (summary[3]) return give_up((summary[3] & 2) ? "an action fails isavalidinput" : "halo too small");
    // ---- the axes the displacement does not depend on -> compact table ---------------------------------------------------
    const long long full[4] = {P.plane, (long long)P.dim[2] * P.dim[3], P.dim[3], 1};
    int inv = 0;
    if (want_tables) {
        const int all = 0xf;
        HIPCHK(hipMemcpyAsync(L.summary + 6, &all, sizeof(int), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_lean4_invariance, grid_for(h->owned), 256, 0, h->stream, P, L, (const float2*)tsp_node);
        HIPCHK(hipMemcpyAsync(&inv, L.summary + 6, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    h->lean4_tables = inv;
    if (inv) {
        const int dims[4] = {rows, P.dim[1], P.dim[2], P.dim[3]};
        int nk[4];
        long long cs = 1;
        for (int d = 3; d >= 0; --d) {
            nk[d] = (inv >> d) & 1 ? 1 : dims[d];
            L.cs[d] = (inv >> d) & 1 ? 0 : (int)cs;
            cs *= nk[d];
        }
        L.csize = cs;
        float2* compact = nullptr;
        if ((rc = dev_alloc(h, (size_t)2 * cs, &compact))) return rc;
        hipLaunchKernelGGL(k_lean4_compact, grid_for(cs), 256, 0, h->stream, P, L, (const float2*)tsp_node, compact, nk[0], nk[1], nk[2], nk[3]);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(h->stream));
        dev_release(h, tsp_node);
        tsp_node = nullptr;
        L.tsp = compact;
    } else {
        for (int d = 0; d < 4; ++d) L.cs[d] = (int)full[d];
        L.csize = h->owned;
        L.tsp = tsp_node;
    }
    // ---- the velocity cells every node reaches, once: each candidate tiling folds them into its window boxes ---------------
    struct BoxGuard {  // (set-up scratch: four bytes per owned node, gone on every way out)
        Lean4P& L;
        ~BoxGuard() {
            if (L.box) (void)hipFree((void*)L.box);
            L.box = nullptr;
        }
    } box_guard{L};
    if (!ovr_is("TUNE", 0) && !(ovr("TV0") && ovr("TV1"))) {  // (a single candidate computes its boxes directly)
        char4* box = nullptr;
        if (hipMalloc((void**)&box, (size_t)h->owned * sizeof(char4)) != hipSuccess) {
            (void)hipGetLastError();  // no room: the direct path
            box = nullptr;
        }
        if (box) {
            L.box = box;
            L.V0 = P.dim[2];  // (the geometry fields every candidate sets again in lean4_try)
            L.V1 = P.dim[3];
            L.posdim1 = P.dim[1];
            L.vplane = (long long)L.V0 * L.V1;
            L.owned = h->owned;
            int over = 0;
            HIPCHK(hipMemcpyAsync(L.summary + 7, &over, sizeof(int), hipMemcpyHostToDevice, h->stream));
            hipLaunchKernelGGL(k_lean4_nodebox, grid_for(h->owned), 256, 0, h->stream, P, L, box);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(&over, L.summary + 7, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            if (over) {  // a reach beyond +-126 cells does not fit a byte
                (void)hipFree(box);
                L.box = nullptr;
            }
        }
    }
    // ---- tiling candidates, timed ------------------------------------------------------------------------------------------
    std::vector<int2> hpt0((size_t)P.dim[0] * P.dim[2]);
    HIPCHK(hipMemcpy(hpt0.data(), pt0, hpt0.size() * sizeof(int2), hipMemcpyDeviceToHost));
    const size_t budget = ovr("LDS_KB") ? (size_t)atoi(ovr("LDS_KB")) * 1024 : (size_t)80 * 1024;  // two workgroups per CU
    std::vector<Lean4Cand> cands;
    const int V1 = P.dim[3];
    int pin[3] = {0, 0, 0};
    if (ovr("L4PIN") && sscanf(ovr("L4PIN"), "%d/%d/%d", &pin[0], &pin[1], &pin[2]) == 3 && pin[0] > 0 && pin[1] >= 64 && pin[2] > 0) {
        // exactly one candidate of the list below -- rows cap / threads / widest tile, as `choice=` of pvi_describe prints it:
        // the counter passes pin the shape an unprofiled create chose (the timed choice can flip under the profiler)
        cands.push_back({pin[0], pin[1], pin[2]});
    } else if (ovr("TV0") && ovr("TV1")) {  // rows cap, and the tile width the workgroup is sized for (cap x width threads)
        cands.push_back({atoi(ovr("TV0")), std::min(512, ((atoi(ovr("TV0")) * atoi(ovr("TV1")) + 63) / 64) * 64), atoi(ovr("TV1"))});
    } else {
        // Tilings worth timing: workgroups of 3 .. 8 waves, and for each the row caps whose tiles (step-aligned row pieces,
        // each split into columns as wide as the workgroup allows) keep the largest share of the lanes busy = nodes of a
        // velocity plane / (tiles x threads), judged on a middle row.  The best caps per workgroup size are timed.
        std::vector<int4> tl;
        for (int min_threads : {192, 64}) {  // (small grids: whatever fills a wave)
            for (int threads : {512, 384, 320, 256, 192, 128, 64}) {
                if (threads < min_threads || (min_threads == 64 && threads >= 192)) continue;
                struct Eff {
                    double e;
                    int cap;
                };
                std::vector<Eff> effs;
                size_t last_n = 0;
                for (int cap = std::min(P.dim[2], threads / 8); cap >= 2; --cap) {
                    lean4_row_tiles(hpt0, P.row_begin + rows / 2, P.dim[2], V1, cap, threads, V1, tl);
                    if (tl.size() == last_n) continue;  // (most caps give the same pieces as their neighbour)
                    last_n = tl.size();
                    int wmin = V1;
                    for (auto& t : tl) wmin = std::min(wmin, t.w);
                    if (wmin < std::min(V1, 12)) continue;  // very narrow tiles: the window is all halo
                    effs.push_back({(double)P.dim[2] * V1 / ((double)tl.size() * threads), cap});
                }
                std::sort(effs.begin(), effs.end(), [](const Eff& a, const Eff& b) { return a.e > b.e; });
                for (size_t i = 0; i < effs.size() && i < 4 && cands.size() < 24; ++i)
                    if (effs[i].e >= 0.85 * effs[0].e) cands.push_back({effs[i].cap, threads, V1});
            }
            if (!cands.empty()) break;
        }
    }
    // ---- the choice of an earlier create of the same problem shape in this process -------------------------------------------
    char key[256];
    snprintf(key, sizeof(key), "%d/%d:%dx%dx%dx%d:A%d:rows%d:dt%.17g:lb%.17g,%.17g:ub%.17g,%.17g:lds%zu", h->device, h->d.dynamics_id,
             P.dim[0], P.dim[1], P.dim[2], P.dim[3], P.A, rows, P.dt, P.lb[2], P.lb[3], P.ub[2], P.ub[3], budget);
    bool from_cache = false;
    if (!ovr_is("TUNE", 0) && !(ovr("TV0") && ovr("TV1")) && !ovr("L4PIN") && !ovr_is("TUNE", 2)) {
        std::lock_guard<std::mutex> lk(g_lean4_choice_mu);
        auto it = g_lean4_choice.find(key);
        if (it != g_lean4_choice.end()) {
            cands.assign(1, it->second);
            from_cache = true;
        }
    }
    const bool tune = !ovr_is("TUNE", 0) && cands.size() > 1;
    // Timed candidates (round 4): every candidate sweeps the SAME few rows of axis 0 from the middle of the slab -- one warm-up
    // and five timed sweeps, each between its own pair of events -- and is judged by the MEDIAN; a later candidate displaces
    // the best so far only by 5 %.  Round 3 timed two whole-grid sweeps per candidate (C4: 52 x 25 ms) and a 2 % margin: the
    // choice flipped between runs (55x26 / 55x22, a 9 % swing of the bench line) and set-up took 3.5-6 s.
    const int sub_rows = std::min(rows, 12), sub_r0 = (rows - sub_rows) / 2;
    hipEvent_t tev[7] = {};
    struct EvGuard {
        hipEvent_t* e;
        ~EvGuard() {
            for (int i = 0; i < 7; ++i)
                if (e[i]) (void)hipEventDestroy(e[i]);
        }
    } ev_guard{tev};
    if (tune)
        for (auto& e : tev) HIPCHK(hipEventCreate(&e));
    float best_ms = 1e30f;
    int best = -1;
    for (size_t ci = 0; ci < cands.size(); ++ci) {
        int narrower = 0;
        rc = lean4_try(h, hpt0, cands[ci].cap, cands[ci].w, cands[ci].wmax, budget, &narrower, tune ? sub_r0 : 0,
                       tune ? sub_rows : -1);
        if (rc < 0) return rc;
        if (rc) continue;
        if (!tune) {
            best = (int)ci;
            break;
        }
        float ms = 0.f;
        SweepCtl sc;
        memset(&sc, 0, sizeof(sc));
        sc.ctrl = h->ctrl;
        sc.slot = h->slots;
        sc.result = h->results;
        sc.tol = -1.0;
        bool hopeless = false;
        for (int rep = 0; rep < 6 && rc == 0 && !hopeless; ++rep) {  // one warm-up, five timed
            HIPCHK(hipEventRecord(tev[rep], h->stream));
            hipLaunchKernelGGL(k_reset_stats, 1, STAT_WORDS, 0, h->stream, h->slots, STAT_WORDS);
            hipLaunchKernelGGL(k_begin_batch, 1, 1, 0, h->stream, h->ctrl);
            rc = h->pi_size == 1 ? launch_lean4_t<unsigned char>(h, (const float*)h->J[h->cur], (float*)h->J[h->cur ^ 1], 1.f, h->stream, sc, true)
                                 : launch_lean4_t<unsigned short>(h, (const float*)h->J[h->cur], (float*)h->J[h->cur ^ 1], 1.f, h->stream, sc, true);
            if (rc) return rc;
            if (rep == 0) {  // (the warm-up sweep of a shape far off the best: not worth five more)
                float warm = 0.f;
                HIPCHK(hipEventRecord(tev[6], h->stream));
                HIPCHK(hipStreamSynchronize(h->stream));
                HIPCHK(hipEventElapsedTime(&warm, tev[0], tev[6]));
                if (best >= 0 && warm > 1.6f * best_ms) {
                    hopeless = true;
                    ms = warm;
                }
            }
        }
        if (rc) return rc;
        if (!hopeless) {
            HIPCHK(hipEventRecord(tev[6], h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            float t[5];
            for (int i = 0; i < 5; ++i) HIPCHK(hipEventElapsedTime(&t[i], tev[1 + i], tev[i + 2 <= 5 ? i + 2 : 6]));
            std::sort(t, t + 5);
            ms = t[2];
        }
        {
            // (milliseconds of the timed rows scaled to the slab: comparable with a whole sweep)
            const float full = ms * (float)rows / (float)sub_rows;
            const size_t at = strlen(h->lean4_cands);
            if (cands[ci].wmax < V1)
                snprintf(h->lean4_cands + at, sizeof(h->lean4_cands) - at, ",%d/%d/%d:%.2f", cands[ci].cap, cands[ci].w, cands[ci].wmax, full);
            else
                snprintf(h->lean4_cands + at, sizeof(h->lean4_cands) - at, "%s%d/%d:%.2f", at ? "," : "", cands[ci].cap, cands[ci].w, full);
        }
        // (a narrower twin of a shape that is in the running: one more workgroup per CU may pay for the extra window halo)
        if (narrower && cands[ci].wmax == V1 && ms < 1.1f * best_ms && cands.size() < 48)
            cands.push_back({cands[ci].cap, cands[ci].w, narrower});
        if (ms < 0.95f * best_ms) {  // a later candidate must win by 5 %: within the timing noise the choice stays put, so the
            best_ms = ms;            // shape (and with it the committed counter passes) is the same from run to run
            best = (int)ci;
        }
    }
    if (best < 0) return give_up(h->lean_why[0] ? h->lean_why : "no tile shape fits the LDS budget");
    if (tune) {
        if ((rc = lean4_try(h, hpt0, cands[(size_t)best].cap, cands[(size_t)best].w, cands[(size_t)best].wmax, budget, nullptr)))
            return rc < 0 ? rc : give_up("tile shape lost");
        // the timed sweeps wrote into the second J buffer, pi and the control block
        HIPCHK(hipMemsetAsync(h->ctrl, 0, sizeof(Ctrl), h->stream));
        HIPCHK(hipMemsetAsync(h->pi, 0, (size_t)h->owned * h->pi_size, h->stream));
        HIPCHK(hipMemsetAsync(h->J[h->cur ^ 1], 0, (size_t)h->stored * 4, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    if (best >= 0 && !from_cache && tune) {
        std::lock_guard<std::mutex> lk(g_lean4_choice_mu);
        g_lean4_choice[key] = cands[(size_t)best];
    }
    if (from_cache) snprintf(h->lean4_cands, sizeof(h->lean4_cands), "cached");
    snprintf(h->lean4_choice, sizeof(h->lean4_choice), "%d/%d/%d", cands[(size_t)best].cap, cands[(size_t)best].w, cands[(size_t)best].wmax);
    h->lean_why[0] = 0;
    h->lean4_ok = true;
    return PVI_OK;
}

static int lean_setup(pvi_problem* h) {
    const DevP& P = h->P;
    LeanP& L = h->LP;
    const float* actc = L.actc;  // uploaded by pvi_create together with the float4 action table
    memset(&L, 0, sizeof(L));
    L.actc = actc;
    h->lean_ok = false;
    if (!h->fast_ok || ovr("NO_LEAN")) return PVI_OK;
    const int DOF = P.dof, M = P.m;
    int rc;
    if ((rc = lean4_setup(h))) return rc;
    if (h->lean4_ok) return PVI_OK;  // 4-D grids: the paired-window kernel (sweep_lean4.inc)
    if ((rc = dev_alloc(h, (size_t)DOF * h->owned, &L.ta))) return rc;
    if ((rc = dev_alloc(h, (size_t)DOF * M * h->owned, &L.tB))) return rc;
    if ((rc = dev_alloc(h, (size_t)h->owned, &L.gx))) return rc;
    if ((rc = dev_alloc(h, (size_t)h->owned, &L.flag))) return rc;
    if ((rc = dev_alloc(h, (size_t)P.dim[0] * P.dim[DOF], &L.pt0))) return rc;
    if (DOF == 2 && (rc = dev_alloc(h, (size_t)P.dim[1] * P.dim[3], &L.pt1))) return rc;
    if ((rc = dev_alloc(h, 8, &L.summary))) return rc;
    L.guard = h->F.guard;
    hipLaunchKernelGGL(k_lean_pairs, grid_for((long long)P.dim[0] * P.dim[DOF]), 256, 0, h->stream, P, 0, L.pt0);
    if (DOF == 2)
        hipLaunchKernelGGL(k_lean_pairs, grid_for((long long)P.dim[1] * P.dim[3]), 256, 0, h->stream, P, 1, L.pt1);
    HIPCHK(hipGetLastError());
    // lanes per node for small grids: as many as still fit ONE round of resident waves (1024 SIMDs x 8 waves x 64
    // lanes); a second round costs more than the extra parallelism brings (201x201x201: 27 -> 23 us)
    int ls = 0;
    // (measured: 401^2 x 101 is best at 4 lanes per node -- 1.2 rounds, 25 actions per lane; 201^2 x 201 loses at 16
    //  lanes with 12 actions per lane: allow a quarter round more while a lane keeps >= 16 actions)
    while (((h->owned << (ls + 1)) <= (1ll << 19) ||
            ((h->owned << (ls + 1)) <= 655360 && P.A / (2 << ls) >= 16)) &&
           (2 << ls) <= 16 && (4 << ls) <= P.A)
        ++ls;
    if (const char* e = ovr("LSPLIT")) ls = atoi(e);
    L.lsplit = ls;
    const int spb = std::max(16, 256 >> ls);  // nodes per workgroup
    int budget = DOF == 1 ? 8 * 1024 : 20 * 1024;  // floats: 32 KB (2-D), 80 KB (4-D: two workgroups per CU)
    if (const char* e = ovr("LDS_KB")) budget = atoi(e) * 256;
    budget = std::min(budget, 40000);
    int shapes[8][2];
    int ns = 0;
    if (ovr("TV0") && ovr("TV1")) {
        shapes[ns][0] = atoi(ovr("TV0"));
        shapes[ns++][1] = atoi(ovr("TV1"));
    } else if (DOF == 1) {
        // measured on 1001^2 x 51: 8x32 51.7 us, 4x63 52.3, 2x126 53.8, 1x251 55.0, 512-thread shapes 56-60
        shapes[ns][0] = std::max(1, spb / 32); shapes[ns++][1] = 32;
        shapes[ns][0] = std::max(1, spb / 64); shapes[ns++][1] = 64;
        shapes[ns][0] = 1; shapes[ns++][1] = spb;
        shapes[ns][0] = 1; shapes[ns++][1] = std::max(16, spb / 2);
    } else {
        // 4-D: long tiles along the last axis amortise the window best (measured on 101^4: 10x51 tile)
        const int V1 = P.dim[3];
        const int t1 = (V1 + (V1 + 63) / 64 - 1) / ((V1 + 63) / 64);
        shapes[ns][0] = std::max(1, 2 * spb / t1); shapes[ns++][1] = t1;
        shapes[ns][0] = std::max(1, spb / t1); shapes[ns++][1] = t1;
        shapes[ns][0] = std::max(1, 2 * spb / 32); shapes[ns++][1] = 32;
        shapes[ns][0] = std::max(1, spb / 32); shapes[ns++][1] = 32;
        shapes[ns][0] = std::max(1, spb / 16); shapes[ns++][1] = 16;
    }
    // two nodes per thread (2-D, uniform action walk; tiles twice as tall for the same workgroup size) halve the
    // per-wave fixed work and fit 1001^2 into one round of resident waves (A = 1: 21 -> 15 us), but the two
    // register-resident node contexts cost the action loop more than that (C2 43.7 -> 49 us at 71 VGPRs / 7 waves,
    // 55 us squeezed to 63 VGPRs): opt-in for experiments, PVI_NPT=2
    L.npt = 1;
    if (DOF == 1 && ls == 0 && ovr("NPT")) {
        const int want = atoi(ovr("NPT"));
        if (want == 2) L.npt = want;
    }
    // 4-D: the best tile shape depends on how the grid divides (101^4: 15x34 beats 10x51 by 8 %, 151^4: 19x26 beats
    // 16x31 by 7 %) -- time the candidates (widths V1/k, as many rows as fit 512 threads) with two real sweeps each and
    // keep the fastest.  Results do not depend on the shape (same arithmetic per node).  PVI_TUNE=0 switches it off.
    if (DOF == 2 && ls == 0 && !(ovr("TV0") && ovr("TV1")) && !(ovr("TUNE") && !atoi(ovr("TUNE")))) {
        const int V1 = P.dim[3];
        float best_ms = 1e30f;
        int best[2] = {0, 0};
        for (int k = 1; k <= 8; ++k) {
            const int w = (V1 + k - 1) / k;
            if (w > 64 && k < 8) continue;
            if (w < 16) break;
            const int t0 = std::max(1, std::min(L.V0 ? L.V0 : P.dim[2], 512 / w));
            rc = lean_try(h, t0, w, budget);
            if (rc < 0) return rc;
            if (rc == 2) break;
            if (rc != 0) continue;
            const int threads = L.TV0 * L.TV1;
            h->lean_block = ((threads + 63) / 64) * 64;
            if (h->lean_block > 512) continue;
            h->lean_ok = true;
            h->lean_lds_attr = false;
            float ms = 0.f;
            for (int rep = 0; rep < 3 && rc == 0; ++rep) {  // one warm-up, two timed
                if (rep == 1) HIPCHK(hipEventRecord(h->ev0, h->stream));
                hipLaunchKernelGGL(k_reset_stats, 1, STAT_WORDS, 0, h->stream, h->slots, STAT_WORDS);
                hipLaunchKernelGGL(k_begin_batch, 1, 1, 0, h->stream, h->ctrl);
                rc = launch_sweep(h, h->cur, 1.0, h->stream, 0, -1.0);
            }
            h->lean_ok = false;
            if (rc) return rc;
            HIPCHK(hipEventRecord(h->ev1, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
            if (ms < best_ms) {
                best_ms = ms;
                best[0] = L.TV0;
                best[1] = L.TV1;
            }
        }
        if (best[0]) {
            ns = 0;
            shapes[ns][0] = best[0];
            shapes[ns++][1] = best[1];
        }
        // the timed sweeps wrote garbage into the second J buffer, pi and the control block: clear what a caller
        // could observe before the first pvi_terminal_cost / pvi_set_J
        HIPCHK(hipMemsetAsync(h->ctrl, 0, sizeof(Ctrl), h->stream));
        HIPCHK(hipMemsetAsync(h->pi, 0, (size_t)h->owned * h->pi_size, h->stream));
        HIPCHK(hipMemsetAsync(h->J[h->cur ^ 1], 0, (size_t)h->stored * (h->d.dtype == PVI_F64 ? 8 : 4), h->stream));
    }
    // first candidate shape that fits the LDS budget (rc: 0 taken / lean_ok set, < 0 error)
    int rowmul = 1;  // 2-D: 2 = twice the rows per workgroup at one node per thread (512 threads)
    auto take_shape = [&]() -> int {
        for (int k = 0; k < ns; ++k) {
            int r = lean_try(h, shapes[k][0] * L.npt * rowmul, shapes[k][1], budget);
            if (r < 0) return r;
            if (r == 2) break;
            if (r == 0) {
                const int threads = L.npt > 1 ? L.half : ((L.TV0 * L.TV1) << L.lsplit);
                h->lean_block = ((threads + 63) / 64) * 64;
                if (h->lean_block > (L.npt > 1 ? 256 : 512)) continue;
                h->lean_ok = true;
                h->lean_lds_attr = false;
                return 0;
            }
        }
        return 0;
    };
    // 2-D grids walked uniformly: one or two nodes per thread?  Two halve the waves (dispatch, per-wave set-up, one
    // round of resident waves instead of two) but leave less to overlap; which wins depends on the action count
    // (2001^2 x 21: 67.7 -> 59.1 us with two, 1001^2 x 51: 35.4 -> 37.5 us), so both run a few timed sweeps here and
    // the faster stays.  Results do not depend on it (same arithmetic per node).  PVI_NPT fixes it, PVI_TUNE=0 keeps 1.
    if (DOF == 1 && ls == 0 && !ovr("NPT") && h->owned >= (1 << 17) && !(ovr("TUNE") && !atoi(ovr("TUNE")))) {
        // (clocks ramp up during the first sweeps after a create: the candidates alternate, two rounds of 40 timed
        //  sweeps behind 20 untimed ones each, and a candidate is judged by its faster round)
        // third candidate: one node per thread in 512-thread workgroups (half as many workgroups to dispatch, the window
        // shared by twice the rows): 1001^2 x 51 35.9 -> 33.1 us (768 threads: 39 us, 1024: 50 us)
        float best_of[4] = {0.f, 1e30f, 1e30f, 1e30f};
        for (int round = 0; round < 2; ++round)
            for (int cand = 1; cand <= 3; ++cand) {
                L.npt = cand == 2 ? 2 : 1;
                rowmul = cand == 3 ? 2 : 1;
                h->lean_ok = false;
                if ((rc = take_shape()) < 0) return rc;
                if (!h->lean_ok) continue;
                float ms = 0.f;
                for (int rep = 0; rep < 60 && rc == 0; ++rep) {
                    if (rep == 20) HIPCHK(hipEventRecord(h->ev0, h->stream));
                    hipLaunchKernelGGL(k_reset_stats, 1, STAT_WORDS, 0, h->stream, h->slots, STAT_WORDS);
                    hipLaunchKernelGGL(k_begin_batch, 1, 1, 0, h->stream, h->ctrl);
                    rc = launch_sweep(h, h->cur, 1.0, h->stream, 0, -1.0);
                }
                h->lean_ok = false;
                if (rc) return rc;
                HIPCHK(hipEventRecord(h->ev1, h->stream));
                HIPCHK(hipStreamSynchronize(h->stream));
                HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
                best_of[cand] = std::min(best_of[cand], ms);
            }
        int best_cand = 1;  // a candidate other than the plain one must win by 2 %
        if (best_of[2] < 0.98f * best_of[best_cand]) best_cand = 2;
        if (best_of[3] < (best_cand == 1 ? 0.98f : 1.f) * best_of[best_cand]) best_cand = 3;
        const int best_npt = best_cand == 2 ? 2 : 1;
        rowmul = best_cand == 3 ? 2 : 1;
        L.npt = best_npt;
        HIPCHK(hipMemsetAsync(h->ctrl, 0, sizeof(Ctrl), h->stream));
        HIPCHK(hipMemsetAsync(h->pi, 0, (size_t)h->owned * h->pi_size, h->stream));
        HIPCHK(hipMemsetAsync(h->J[h->cur ^ 1], 0, (size_t)h->stored * (h->d.dtype == PVI_F64 ? 8 : 4), h->stream));
    }
    if ((rc = take_shape()) < 0) return rc;
    if (h->lean_ok && L.tb_tile) {  // tB lives per tile: the per-node copy is not needed any more
        dev_release(h, L.tB);
        L.tB = nullptr;
    }
    // float32 accuracy guard: the displacement rel = ta + sum tB u is formed from float32 copies of ta and tB.  When
    // those operands are hundreds of cells and cancel (light links with strong actuators: the default two-link arm has
    // |ta| + |tB u| up to 3800 cells), their rounding alone moves the fraction by > 1e-5 cells and J by > 1e-5
    // relative (tools/tools_fuzz.py).  Such problems run the kernel with float64 dynamics and float32 storage instead.
    if (h->lean_opmag > 256) {
        snprintf(h->lean_why, sizeof(h->lean_why), "float32 displacement operands reach %d cells: float64 dynamics", h->lean_opmag);
        h->lean_ok = false;
        h->fast_ok = false;
    }
    if (!h->lean_ok) {  // release the per-node arrays: the fast / tiled kernels do not need them
        dev_release(h, L.ta); dev_release(h, L.tB); dev_release(h, L.gx); dev_release(h, L.flag);
        dev_release(h, L.win); dev_release(h, L.tbt);
        L.ta = L.tB = L.gx = nullptr; L.flag = nullptr; L.win = nullptr; L.tbt = nullptr;
    }
    return PVI_OK;
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
            const unsigned gm = grid_for(h->owned);
#define VM(DYN) hipLaunchKernelGGL((k_valid_mask<DYN>), gm, 256, 0, h->stream, h->P, h->act64, vm, cnt)
            switch (d->dynamics_id) {
                case PVI_DYN_CARTPOLE: VM(PVI_DYN_CARTPOLE); break;
                case PVI_DYN_TWOLINK: VM(PVI_DYN_TWOLINK); break;
                case PVI_DYN_NODE_2x1: VM(PVI_DYN_NODE_2x1); break;
                default: VM(PVI_DYN_NODE_2x2); break;
            }
#undef VM
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
                 "reach=0 opmag=0 sparse=0 win=1 tables=%d ptab=%d gx=%s stage=%d choice=%s tiles_per_plane=%d bands=%d cands=%s note=%s", h->L4.TV0, h->L4.TV1,
                 h->lean4_grid, h->lean4_block, h->L4.RS, h->lean4_lds, h->lean4_tables, h->lean4_ptab_inv, h->L4.gx ? "node" : "axes", h->lean4_stage,
                 h->lean4_choice, h->L4.ntr, h->lean4_bands, h->lean4_cands[0] ? h->lean4_cands : "-", h->lean_why);
        return PVI_OK;
    }
    snprintf(buf, (size_t)n, "path=%s tile=%dx%d grid=%ux%ux%u block=%d pw1=%d lds_bytes=%zu lsplit=%d tb_tile=%d dma16=%d npt=%d reach=%d opmag=%d sparse=%d win=0 tables=0 note=%s",
             path, h->LP.TV0, h->LP.TV1, h->lean_grid.x, h->lean_grid.y, h->lean_grid.z, h->lean_block, h->lean_pw1,
             h->lean_lds, h->lean_ok ? h->LP.lsplit : h->F.lsplit, h->LP.tb_tile, h->lean_ok ? h->LP.dma16 : 0,
             h->lean_ok ? h->LP.npt : 1, h->lean_reach, h->lean_opmag,
             (h->d.dtype == PVI_F32 && h->sparse64 && h->vmask) ? 1 : 0, h->lean_why);
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
    if constexpr (sizeof(REAL) == 4) {
        if (h->lean4_ok && !h->force_exact) return launch_lean4_t<PI_T>(h, Jin, Jout, (float)alpha, st, sc);
        if (h->lean_ok && !h->force_exact) {
            const float al = (float)alpha;
            sc.nblocks = h->lean_grid.x;
            if (sc.split_finish != 2)
                sc.split_finish = ((sc.nblocks >= 16384u || ovr_is("SPLIT_FINISH", 1)) && !ovr("NO_SPLIT_FINISH")) ? 1 : 0;
#define LEAN3(DYN, U, NP) LEAN4(DYN, U, NP, 0)
#define LEAN4(DYN, U, NP, RSK)                                                                                      \
    {                                                                                                               \
        auto kfn = k_sweep_lean<DYN, PI_T, U, NP, RSK>;                                                                  \
        set_kname(h, "k_sweep_lean", (int)DYN, tname<PI_T>(), (bool)U, (int)NP, (int)RSK);                              \
        if (!h->lean_lds_attr && h->lean_lds > 48 * 1024) {                                                         \
            HIPCHK(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, PVI_LDS_MAX)); \
            h->lean_lds_attr = true;                                                                                \
        }                                                                                                           \
        hipLaunchKernelGGL(kfn, h->lean_grid, h->lean_block, h->lean_lds, st, h->P, h->LP, h->F.act, h->LP.actc, Jin, Jout, pi, al, \
                           sc);                                                                                     \
        if (sc.split_finish == 1) hipLaunchKernelGGL(k_sweep_finish, 1, STAT_SHARDS, 0, st, sc);                    \
    }
#define LEAN(DYN)                                                  \
    if (h->LP.lsplit == 0) {                                       \
        if (h->LP.RS == 64)                          \
            LEAN4(DYN, true, 1, 64)                                \
        else if (Dyn<DYN>::DOF == 1 && h->LP.RS == 128) \
            LEAN4(DYN, true, 1, 128)                               \
        else                                                       \
            LEAN3(DYN, true, 1)                                    \
    } else                                                         \
        LEAN3(DYN, false, 1)
            switch (h->d.dynamics_id) {
                case PVI_DYN_PENDULUM:
                    if (h->LP.npt == 2 && h->LP.RS == 64)
                        LEAN4(PVI_DYN_PENDULUM, true, 2, 64)
                    else if (h->LP.npt == 2 && h->LP.RS == 128)
                        LEAN4(PVI_DYN_PENDULUM, true, 2, 128)
                    else if (h->LP.npt == 2)
                        LEAN3(PVI_DYN_PENDULUM, true, 2)
                    else
                        LEAN(PVI_DYN_PENDULUM)
                    break;
                case PVI_DYN_CARTPOLE: LEAN(PVI_DYN_CARTPOLE) break;
                case PVI_DYN_NODE_1x1: LEAN(PVI_DYN_NODE_1x1) break;
                case PVI_DYN_NODE_2x1: LEAN(PVI_DYN_NODE_2x1) break;
                case PVI_DYN_NODE_2x2: LEAN(PVI_DYN_NODE_2x2) break;
                default: LEAN(PVI_DYN_TWOLINK) break;
            }
#undef LEAN
#undef LEAN3
#undef LEAN4
            HIPCHK(hipGetLastError());
            return PVI_OK;
        }
        if (h->fast_ok && !is_node_dyn(h->d.dynamics_id) && !h->force_exact) {
            const unsigned gf = grid_for(h->owned << h->F.lsplit);
            const float al = (float)alpha;
            sc.nblocks = gf;
#define FAST(DYN)                                                                                                  \
    set_kname(h, "k_sweep_fast", (int)DYN, tname<PI_T>(), h->F.lsplit == 0);                                       \
    if (h->F.lsplit == 0)                                                                                          \
        hipLaunchKernelGGL((k_sweep_fast<DYN, PI_T, true>), gf, 256, 0, st, h->P, h->F, Jin, Jout, pi, al, sc, h->F.act);                                                                                  \
    else                                                                                                           \
        hipLaunchKernelGGL((k_sweep_fast<DYN, PI_T, false>), gf, 256, 0, st, h->P, h->F, Jin, Jout, pi, al, sc, h->F.act);
            switch (h->d.dynamics_id) {
                case PVI_DYN_PENDULUM: FAST(PVI_DYN_PENDULUM) break;
                case PVI_DYN_CARTPOLE: FAST(PVI_DYN_CARTPOLE) break;
                default: FAST(PVI_DYN_TWOLINK) break;
            }
#undef FAST
            HIPCHK(hipGetLastError());
            return PVI_OK;
        }
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
        if (h->use64 && !h->force_exact) {
            const bool off32 = (unsigned long long)h->stored * 8ull < (1ull << 32);
            // 4-D: 8 x 8 velocity patches per wave (PVI_PATCH=0: consecutive nodes)
            const bool patch = h->P.n == 4 && h->patch64 != 0;
            unsigned gp = g;
            if (patch) {
                const long long waves = (long long)(h->P.row_end - h->P.row_begin) * h->P.dim[1] *
                                        ((h->P.dim[2] + 7) / 8) * ((h->P.dim[3] + 7) / 8);
                gp = (unsigned)((waves + 3) / 4);
                sc.nblocks = gp;
            }
            const int sparse = h->sparse64;
            sc.xcd_remap = (gp >= 64u && !(ovr("XCD64") && !atoi(ovr("XCD64")))) ? 1 : 0;
            if (sc.xcd_remap && h->P.n == 4 && !(ovr("XCD_CHUNK") && atoi(ovr("XCD_CHUNK")) == 0)) {
                // 4-D: the blocks of ONE row of axis 0 per chunk, chunks dealt round-robin to the XCDs (round 4).  One contiguous
                // eighth of the rows per XCD left the XCDs with unequal work -- what a node costs depends on its position (rows
                // near the faces leave the box; the sparse walk's in-box share varies with the joint angles): C5 15.8 -> 14.65 ms
                // with chunks of one row, 14.9 with two, 15.7 with four (profiles/r04_c5_chunks.log)
                const int rows_c = ovr("XCD_CHUNK") ? atoi(ovr("XCD_CHUNK")) : 1;
                const long long per_row = (long long)gp / std::max(1, h->P.row_end - h->P.row_begin);
                const long long C = per_row * std::max(1, rows_c);
                if (C >= 2 && C * 16 <= (long long)gp) sc.xcd_remap = (int)C;
            }
            const size_t lds64 = h->levr_bytes + (sparse == 1 ? (size_t)h->P.A * sizeof(Act64) : 0);
#define S64Q(DYN, PT, SP)                                                                                             \
    set_kname(h, "k_sweep64", (int)DYN, tname<PI_T>(), off32, (bool)PT, (bool)SP);                                    \
    if (off32)                                                                                                        \
        hipLaunchKernelGGL((k_sweep64<DYN, PI_T, true, PT, SP>), gp, 256, lds64, st, h->P, Jin, Jout, pi, alpha, sc,  \
                           h->act64, h->levr, h->vmask);                                                              \
    else                                                                                                              \
        hipLaunchKernelGGL((k_sweep64<DYN, PI_T, false, PT, SP>), gp, 256, lds64, st, h->P, Jin, Jout, pi, alpha, sc, \
                           h->act64, h->levr, h->vmask);
#define S64P(DYN, PT)                              \
    if constexpr (Dyn<DYN>::DOF == 2) {            \
        if (sparse) {                              \
            S64Q(DYN, PT, true)                    \
        } else {                                   \
            S64Q(DYN, PT, false)                   \
        }                                          \
    } else {                                       \
        S64Q(DYN, PT, false)                       \
    }
#define S64(DYN)                                   \
    if constexpr (Dyn<DYN>::DOF == 2) {            \
        if (patch) {                               \
            S64P(DYN, true)                        \
        } else {                                   \
            S64P(DYN, false)                       \
        }                                          \
    } else {                                       \
        S64P(DYN, false)                           \
    }
            switch (h->d.dynamics_id) {
                case PVI_DYN_PENDULUM: S64(PVI_DYN_PENDULUM) break;
                case PVI_DYN_CARTPOLE: S64(PVI_DYN_CARTPOLE) break;
                case PVI_DYN_TWOLINK: S64(PVI_DYN_TWOLINK) break;
                case PVI_DYN_NODE_1x1: S64(PVI_DYN_NODE_1x1) break;
                case PVI_DYN_NODE_2x1: S64(PVI_DYN_NODE_2x1) break;
                default: S64(PVI_DYN_NODE_2x2) break;
            }
#undef S64
#undef S64P
#undef S64Q
            HIPCHK(hipGetLastError());
            return PVI_OK;
        }
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
    set_kname(h, "k_sweep3_fast", (int)DYN, tname<PI_T>());                                                         \
    hipLaunchKernelGGL((k_sweep3_fast<DYN, PI_T>), g3, 256, 0, st, h->P, Jin, Jout, pi, (float)alpha, sc, h->P.utab, h->P.gu, \
                       h->okmask3)
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

static int launch_sweep(pvi_problem* h, int src, double alpha, hipStream_t st, int k, double tol, int deferred) {
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
    if (h->d.dtype == PVI_F64)
        return h->pi_size == 1 ? launch_sweep_t<double, unsigned char>(h, src, alpha, st, sc)
                               : launch_sweep_t<double, unsigned short>(h, src, alpha, st, sc);
    return h->pi_size == 1 ? launch_sweep_t<float, unsigned char>(h, src, alpha, st, sc)
                           : launch_sweep_t<float, unsigned short>(h, src, alpha, st, sc);
}

// ---- multi-sweep launch (k_sweep64m): one cooperative launch for a whole batch of sweeps -------------------------------------
// Applies to float64 handles on the second-form kernel with the dense walk over consecutive nodes (2-D grids; 4-D ones when
// set-up kept neither patches nor validity masks), whole grid, every workgroup resident.  pvi_override("MULTI", "0") keeps
// one launch per sweep.
template <typename PI_T>
static const void* multi64_kernel(int dyn) {
    switch (dyn) {
        case PVI_DYN_PENDULUM: return (const void*)k_sweep64m<PVI_DYN_PENDULUM, PI_T>;
        case PVI_DYN_CARTPOLE: return (const void*)k_sweep64m<PVI_DYN_CARTPOLE, PI_T>;
        case PVI_DYN_TWOLINK: return (const void*)k_sweep64m<PVI_DYN_TWOLINK, PI_T>;
        case PVI_DYN_NODE_1x1: return (const void*)k_sweep64m<PVI_DYN_NODE_1x1, PI_T>;
        case PVI_DYN_NODE_2x1: return (const void*)k_sweep64m<PVI_DYN_NODE_2x1, PI_T>;
        case PVI_DYN_NODE_2x2: return (const void*)k_sweep64m<PVI_DYN_NODE_2x2, PI_T>;
        default: return nullptr;
    }
}
static bool multi64_applies(pvi_problem* h) {
    if (h->multi64 >= 0) return h->multi64 == 1;
    h->multi64 = 0;
    auto no = [&](const char* why) {
        snprintf(h->multi_why, sizeof(h->multi_why), "%s", why);
        return false;
    };
    if (ovr_is("MULTI", 0)) return no("MULTI=0");
    if (h->d.dtype != PVI_F64 || !h->use64 || h->spline || h->d.dynamics_id == PVI_DYN_TABLE) return no("not the float64 second-form sweep");
    if (h->P.n == 4 && (h->patch64 != 0 || h->sparse64 != 0)) return no("patch mapping / sparse walk");
    if ((unsigned long long)h->stored * 8ull >= (1ull << 32)) return no("64-bit offsets");
    const void* kfn = h->pi_size == 1 ? multi64_kernel<unsigned char>(h->d.dynamics_id) : multi64_kernel<unsigned short>(h->d.dynamics_id);
    if (!kfn) return no("dynamics");
    int coop = 0, per_cu = 0, ncu = 0;
    if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, h->device) != hipSuccess || !coop) return no("no cooperative launch");
    // (with the largest window the launch may ask for: launch_multi64)
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 256, h->levr_bytes + 5 * 512 * 8) != hipSuccess) return no("occupancy query");
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess) return no("device query");
    const unsigned g = grid_for(h->owned);
    if ((long long)g > (long long)per_cu * ncu) return no("more workgroups than are resident");
    // Small grids only: a sweep of 40 workgroups is a latency chain that one launch per sweep dominates (C1: 9.9 -> 7.6 us with
    // the first version of the kernel); with hundreds of workgroups every one of them runs the barrier's L2 write-back and
    // invalidate and the sweep gets SLOWER (401 x 401 x 51: 27 -> 62 us, profiles/r04_multi_first.log).
    if (g > 64u) return no("more than 64 workgroups: one launch per sweep is faster");
    h->multi64 = 1;
    return true;
}
static int launch_multi64(pvi_problem* h, int src, double alpha, double tol, int nsweeps) {
    SweepCtl sc;
    sc.ctrl = h->ctrl;
    sc.slot = h->slots;
    sc.result = h->results;
    sc.tol = tol;
    sc.k = 0;
    sc.nblocks = grid_for(h->owned);
    sc.split_finish = 0;
    sc.xcd_remap = (sc.nblocks >= 64u && !(ovr("XCD64") && !atoi(ovr("XCD64")))) ? 1 : 0;
    sc.regtab = (h->P.n == 2 && h->P.A <= 12 && !ovr_is("REGTAB", 0)) ? 1 : 0;  // (12 = RT of sweep64_body)
    h->regtab64 = sc.regtab;
    // LDS for the workgroup's window of J behind the level table: at most 5 x 512 doubles (WCH of sweep64_body), within 48 KB
    sc.win_bytes = 0;
    if (sc.regtab && !ovr_is("JWIN", 0) && h->levr_bytes + 4096 <= 48 * 1024)
        sc.win_bytes = (int)std::min<size_t>(5 * 512 * 8, 48 * 1024 - h->levr_bytes);
    const void* kfn = h->pi_size == 1 ? multi64_kernel<unsigned char>(h->d.dynamics_id) : multi64_kernel<unsigned short>(h->d.dynamics_id);
    DevP P = h->P;
    const double* Jin = (const double*)h->J[src];
    double* Jout = (double*)h->J[src ^ 1];
    void* pi = h->pi;
    const Act64* act64 = h->act64;
    const double2* levr = h->levr;
    void* args[] = {&P, &Jin, &Jout, &pi, &alpha, &sc, &act64, &levr, &nsweeps};
    set_kname(h, "k_sweep64m", (int)h->d.dynamics_id, h->pi_size == 1 ? tname<unsigned char>() : tname<unsigned short>());
    HIPCHK(hipLaunchCooperativeKernel(kfn, dim3(sc.nblocks), dim3(256), args, (unsigned)(h->levr_bytes + (size_t)sc.win_bytes), h->stream));
    return PVI_OK;
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
        // (1) the handle's production path: J_cur -> the other buffer, pi
        hipLaunchKernelGGL(k_reset_stats, 1, STAT_WORDS, 0, h->stream, h->slots, STAT_WORDS);
        hipLaunchKernelGGL(k_begin_batch, 1, 1, 0, h->stream, h->ctrl);
        int r = launch_sweep(h, h->cur, alpha, h->stream, 0, -1.0);
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
    if (dyn_shape(dyn, &en, &em) || is_node_dyn(dyn) || is_dyn3(dyn)) return fail(PVI_EINVAL, "no closed-form dynamics with id %d", dyn);
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
