// f64.hip -- float64 second form of the sweep (k_sweep64: bit-identical to the operation-for-operation k_sweep of pyrovi.hip),
// its multi-sweep launch (k_sweep64m) and the validity masks of the sparse walks (k_valid_mask), with their launchers.
// One of the three translation units of libpyrovi (pyrovi.hip, f64.hip, lean.hip); see core.h / host.h.
#include "core.h"
#include "host.h"

// (the Act64 record of this family is declared in host.h: the handle holds the table)
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

// fold of the per-workgroup statistics records of the register-table form: two records per thread -> per-wave maxima -> folded[]
__device__ __forceinline__ void fold_records(double a0, double a1, double a2, double b0, double b1, double b2, double (*wfold)[4],
                                             double* folded) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    a0 = wave_max(fmax(a0, b0));
    a1 = wave_max(fmax(a1, b1));
    a2 = wave_max(fmax(a2, b2));
    if (lane == 0) {
        wfold[0][wave] = a0;
        wfold[1][wave] = a1;
        wfold[2][wave] = a2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double m0 = fmax(fmax(wfold[0][0], wfold[0][1]), fmax(wfold[0][2], wfold[0][3]));
        const double m1 = fmax(fmax(wfold[1][0], wfold[1][1]), fmax(wfold[1][2], wfold[1][3]));
        const double m2 = fmax(fmax(wfold[2][0], wfold[2][1]), fmax(wfold[2][2], wfold[2][3]));
        folded[0] = m0;
        folded[1] = m1;
        folded[2] = -m2;
        folded[3] = fmax(fabs(m1), fabs(-m2));
    }
    __syncthreads();
}

// MULTI (round 4): the device form of the driver loops dynamicprogramming.py:265-314 for grids whose
// workgroups are all resident.  ONE launch runs up to `nsweeps` backups: everything of a node that does not change between
// sweeps (coordinates, position row, weights, dynamics prologue) stays in its thread's registers, J ping-pongs between the
// two buffers, the three statistics of sweep k go to slot k, a grid barrier separates the sweeps, and every workgroup folds
// the statistics itself and takes the same stop decision (delta <= tol).  Same arithmetic per cell as one launch per sweep:
// J, pi, the statistics and the stop sweep are bit-identical.
// RTM (multi-sweep launches of 2-D grids): 0 the general form with the fence-based barrier, 1 / 2 the register-table form
// (narrow: <= 12 actions, <= 64 workgroups; wide: <= 24, <= 512) -- a compile-time choice, so that each kernel carries only its
// own path (the combined kernel was 57 KB of code and 255 + 122 registers)
template <int DYN, typename PI_T, bool OFF32, bool PATCH, bool SPARSE, bool MULTI, int RTM = 0>
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
        constexpr bool WIDE = RTM == 2;
        constexpr int RT = WIDE ? 24 : 12;  // (WIDE: the form for up to 24 actions and 512 workgroups; its longer code costs C1 a microsecond per sweep)
        [[maybe_unused]] unsigned rt_off[RT];
        [[maybe_unused]] double rt_y[RT], rt_G[RT];
        [[maybe_unused]] unsigned rt_in = 0u;
        [[maybe_unused]] bool regtab = false;
        [[maybe_unused]] double jprev = 0.0;
        [[maybe_unused]] int win_r0 = 0, win_n = 0;
        [[maybe_unused]] bool win_ok = false;
        [[maybe_unused]] double* win = nullptr;
        if constexpr (MULTI && DOF == 1) {
            regtab = RTM > 0;   // (the host launches this instantiation only for grids of at most RT actions: launch_multi64)
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
      [[maybe_unused]] double pv0 = -INFINITY, pv1 = -INFINITY, pv2 = -INFINITY, pu0 = -INFINITY, pu1 = -INFINITY, pu2 = -INFINITY;
      __shared__ double wfold[3][4];
      [[maybe_unused]] bool pending = false;
      __shared__ double folded[4];
      for (int ks = 0;; ++ks) {  // (one trip unless MULTI)
        double best = P.INF;  // position row outside the box: every action costs INF + alpha*0, the first one wins
        int arg = 0;
        constexpr bool rt_done = MULTI && DOF == 1 && RTM > 0;
        if constexpr (MULTI && DOF == 1 && RTM > 0) {
            {
                if constexpr (WIDE) {
                    typedef unsigned v4u __attribute__((ext_vector_type(4)));
                    constexpr int RB = 12;  // actions whose 2 x 16-byte gathers are in flight together when they come from memory
                    v4u r0[RB], r1[RB];
                    constexpr int WCH = 8;  // 16-byte chunks of the window per thread
                    v4u wv[WCH];
                    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Jin, 0, 0xfffffff0u, 0x00020000);
                    const unsigned s0B = (unsigned)P.strd[0] * 8u;
                    constexpr unsigned OOB = 0xffffffffu;  // beyond num_records: the hardware returns zeros without an access
                    auto issue = [&](int b0) {  // the gathers of actions b0 .. b0 + RB - 1, straight from memory
    #pragma unroll
                        for (int k = 0; k < RB; ++k) {
                            if (b0 + k < P.A) {  // (uniform)
                                const unsigned vo = ((rt_in >> (b0 + k)) & 1u) ? rt_off[b0 + k] : OOB;
                                r0[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, 0, 16);    // aux 16 = sc1: bypasses the CU's L1
                                r1[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, s0B, 16);
                            }
                        }
                    };
                    const double a0 = 1.0 - y[0];
                    auto cell = [&](int a, double q00, double q01, double q10, double q11) {
                        const double ya = rt_y[a], a1 = 1.0 - ya;
                        const double Jt = q00 * a0 * a1 + q01 * a0 * ya + q10 * y[0] * a1 + q11 * y[0] * ya;
                        const double Jn = ((rt_in >> a) & 1u) ? Jt : 0.0;
                        const double q = rt_G[a] + alpha * Jn;
                        if (a == 0 || q < best) {
                            best = q;
                            arg = a;
                        }
                    };
                    if (win_ok) {  // (block-uniform) the window: chunk c of thread t = doubles 2 (t + 256 c), 2 (t + 256 c) + 1
                        const unsigned org = (unsigned)((long long)(win_r0 - P.store_begin) * P.strd[0]) * 8u;
    #pragma unroll
                        for (int c = 0; c < WCH; ++c) {
                            const int e = 2 * ((int)threadIdx.x + 256 * c);
                            wv[c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, e < win_n ? org + (unsigned)e * 8u : OOB, 0, 16);
                        }
                    } else if (pos_in) {
                        issue(0);
                    }
                    if (pending) {  // (uniform) the previous sweep's statistics: did it meet the tolerance?
                        pending = false;
                        fold_records(pv0, pv1, pv2, pu0, pu1, pu2, wfold, folded);
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
                                    const double* w0 = win + (rt_off[a] >> 3);  // (a cell outside the box points into the window too)
                                    cell(a, w0[0], w0[1], w0[s0], w0[s0 + 1]);
                                }
                            }
                        }
                    } else if (pos_in) {
    #pragma unroll
                        for (int b0 = 0; b0 < RT; b0 += RB) {
                            if (b0 < P.A) {  // (uniform)
                                if (b0 > 0) issue(b0);  // (a second round trip: only without the window, with more than RB actions)
    #pragma unroll
                                for (int k = 0; k < RB; ++k) {
                                    if (b0 + k < P.A)
                                        cell(b0 + k, __hiloint2double((int)r0[k].y, (int)r0[k].x), __hiloint2double((int)r0[k].w, (int)r0[k].z),
                                             __hiloint2double((int)r1[k].y, (int)r1[k].x), __hiloint2double((int)r1[k].w, (int)r1[k].z));
                                }
                            }
                        }
                    }
                } else {
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
            // (MULTI_MAX_WG workgroups per set: the register-table form runs up to that many, the fenced form at most 64)
            double* part = (double*)sc.slot + (size_t)(ks & 1) * MULTI_MAX_WG * 4;
            double v0 = -INFINITY, v1 = -INFINITY, v2 = -INFINITY;
            [[maybe_unused]] double u0 = -INFINITY, u1 = -INFINITY, u2 = -INFINITY;  // (second record of a thread, register-table form)
            if (rt_done) {
                // REGTAB: J and the statistics went out write-through and are read with sc1 loads: nothing to fence.
                // (Measured and not kept: arrival and statistics as ONE tagged 16-byte granule per value, polled by every
                //  workgroup -- no counter, one round trip less on paper, the same 5.0 us per sweep on C1.)
                if constexpr (WIDE) {
                    block_max3_store<true>(st_j, st_dmax, st_ndmin, part + 4 * blockIdx.x);
                    grid_barrier_wt(&sc.ctrl->ticket, (unsigned)(ks + 1) * gridDim.x);
                    // every thread takes the records of the workgroups t and t + 256 (MULTI_MAX_WG = 512): six sc1 loads (they bypass
                    // the CU's L1, which may hold these words from two sweeps ago), issued here, waited for where they are folded
                    {
                        auto ld = [&](int wg, int k) {
                            return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)(part + 4 * wg + k), __ATOMIC_RELAXED,
                                                                                      __HIP_MEMORY_SCOPE_AGENT));
                        };
                        const int w0 = (int)threadIdx.x, w1 = w0 + 256;
                        if (w0 < (int)gridDim.x) {
                            v0 = ld(w0, 0);
                            v1 = ld(w0, 1);
                            v2 = ld(w0, 2);
                        }
                        if (w1 < (int)gridDim.x) {
                            u0 = ld(w1, 0);
                            u1 = ld(w1, 1);
                            u2 = ld(w1, 2);
                        }
                    }
                } else {
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
                }
                if (ks + 1 < nsweeps) {  // folded behind the next sweep's J loads
                    pv0 = v0;
                    pv1 = v1;
                    pv2 = v2;
                    pu0 = u0;
                    pu1 = u1;
                    pu2 = u2;
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
            if (RTM == 2 && rt_done) {
                fold_records(v0, v1, v2, u0, u1, u2, wfold, folded);
            } else {
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
            }
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
// RTM 1 / 2 (2-D grids only): the register-table form, narrow / wide (sweep64_body)
template <int DYN, typename PI_T, int RTM = 0>
__global__ __launch_bounds__(256) void k_sweep64m(DevP P, const double* Jin, double* Jout, PI_T* __restrict__ pi, double alpha, SweepCtl sc,
                                                  const Act64* __restrict__ act64, const double2* __restrict__ levr, int nsweeps) {
    sweep64_body<DYN, PI_T, true, false, false, true, RTM>(P, Jin, Jout, pi, alpha, sc, act64, levr, nullptr, nsweeps);
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
// launchers
// =================================================================================================
template <typename PI_T>
static int launch_f64v2_t(pvi_problem* h, const double* Jin, double* Jout, double alpha, hipStream_t st, SweepCtl sc) {
    const unsigned g = grid_for(h->owned);
    sc.nblocks = g;
    PI_T* pi = (PI_T*)h->pi;
    {
        {
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
}
int launch_f64v2(pvi_problem* h, const double* Jin, double* Jout, double alpha, hipStream_t st, SweepCtl sc) {
    return h->pi_size == 1 ? launch_f64v2_t<unsigned char>(h, Jin, Jout, alpha, st, sc)
                           : launch_f64v2_t<unsigned short>(h, Jin, Jout, alpha, st, sc);
}

int launch_valid_mask(pvi_problem* h, uint4* vm, unsigned long long* cnt) {
    const unsigned gm = grid_for(h->owned);
#define VM(DYN) hipLaunchKernelGGL((k_valid_mask<DYN>), gm, 256, 0, h->stream, h->P, h->act64, vm, cnt)
    switch (h->d.dynamics_id) {
        case PVI_DYN_CARTPOLE: VM(PVI_DYN_CARTPOLE); break;
        case PVI_DYN_TWOLINK: VM(PVI_DYN_TWOLINK); break;
        case PVI_DYN_NODE_2x1: VM(PVI_DYN_NODE_2x1); break;
        default: VM(PVI_DYN_NODE_2x2); break;
    }
#undef VM
    return PVI_OK;
}

// ---- multi-sweep launch (k_sweep64m): one cooperative launch for a whole batch of sweeps -------------------------------------
// Applies to float64 handles on the second-form kernel with the dense walk over consecutive nodes (2-D grids; 4-D ones when
// set-up kept neither patches nor validity masks), whole grid, every workgroup resident.  pvi_override("MULTI", "0") keeps
// one launch per sweep.
template <typename PI_T>
static const void* multi64_kernel(int dyn, int rtm = 0) {
    if (rtm == 1) {
        switch (dyn) {
            case PVI_DYN_PENDULUM: return (const void*)k_sweep64m<PVI_DYN_PENDULUM, PI_T, 1>;
            case PVI_DYN_NODE_1x1: return (const void*)k_sweep64m<PVI_DYN_NODE_1x1, PI_T, 1>;
            default: return nullptr;
        }
    }
    if (rtm == 2) {
        switch (dyn) {
            case PVI_DYN_PENDULUM: return (const void*)k_sweep64m<PVI_DYN_PENDULUM, PI_T, 2>;
            case PVI_DYN_NODE_1x1: return (const void*)k_sweep64m<PVI_DYN_NODE_1x1, PI_T, 2>;
            default: return nullptr;
        }
    }
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
bool multi64_applies(pvi_problem* h) {
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
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 256, h->levr_bytes + 8 * 512 * 8) != hipSuccess) return no("occupancy query");
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess) return no("device query");
    const unsigned g = grid_for(h->owned);
    if ((long long)g > (long long)per_cu * ncu) return no("more workgroups than are resident");
    // Small grids only: a sweep of 40 workgroups is a latency chain that one launch per sweep dominates (C1: 9.9 -> 7.6 us with
    // the first version of the kernel); with hundreds of workgroups every one of them runs the barrier's L2 write-back and
    // invalidate and the sweep gets SLOWER (401 x 401 x 51: 27 -> 62 us, profiles/r04_multi_first.log).
    // The register-table form (2-D, few actions: write-through hand-off, no fences) keeps paying further: 201 x 201 x 21 float64,
    // 158 workgroups, 12.6 us per sweep as one launch per sweep.
    const bool regtab = h->P.n == 2 && h->P.A <= 24 && !ovr_is("REGTAB", 0);
    if (g > (regtab ? (unsigned)MULTI_MAX_WG : 64u)) return no(regtab ? "more than 512 workgroups" : "more than 64 workgroups: one launch per sweep is faster");
    h->multi_rtm = regtab ? ((h->P.A > 12 || g > 64u) ? 2 : 1) : 0;   // (the wide form's longer code costs C1 a microsecond per sweep)
    if (h->multi_rtm) {
        kfn = h->pi_size == 1 ? multi64_kernel<unsigned char>(h->d.dynamics_id, h->multi_rtm) : multi64_kernel<unsigned short>(h->d.dynamics_id, h->multi_rtm);
        if (!kfn) return no("dynamics");
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 256, h->levr_bytes + 8 * 512 * 8) != hipSuccess) return no("occupancy query");
        if ((long long)g > (long long)per_cu * ncu) return no("more workgroups than are resident");
    }
    h->multi64 = 1;
    return true;
}
int launch_multi64(pvi_problem* h, int src, double alpha, double tol, int nsweeps) {
    SweepCtl sc;
    sc.ctrl = h->ctrl;
    sc.slot = h->slots;
    sc.result = h->results;
    sc.tol = tol;
    sc.k = 0;
    sc.nblocks = grid_for(h->owned);
    sc.split_finish = 0;
    sc.xcd_remap = (sc.nblocks >= 64u && !(ovr("XCD64") && !atoi(ovr("XCD64")))) ? 1 : 0;
    const int rtm = h->multi_rtm;      // (multi64_applies: 0 general form, 1 / 2 register-table form narrow / wide)
    const bool wide = rtm == 2;
    sc.regtab = rtm > 0 ? 1 : 0;
    h->regtab64 = sc.regtab;
    // LDS for the workgroup's window of J behind the level table: at most WCH x 512 doubles (sweep64_body: 5, wide 8), within 48 KB
    sc.win_bytes = 0;
    if (sc.regtab && !ovr_is("JWIN", 0) && h->levr_bytes + 4096 <= 48 * 1024)
        sc.win_bytes = (int)std::min<size_t>((wide ? 8 : 5) * 512 * 8, 48 * 1024 - h->levr_bytes);
    const void* kfn = h->pi_size == 1 ? multi64_kernel<unsigned char>(h->d.dynamics_id, rtm) : multi64_kernel<unsigned short>(h->d.dynamics_id, rtm);
    DevP P = h->P;
    const double* Jin = (const double*)h->J[src];
    double* Jout = (double*)h->J[src ^ 1];
    void* pi = h->pi;
    const Act64* act64 = h->act64;
    const double2* levr = h->levr;
    void* args[] = {&P, &Jin, &Jout, &pi, &alpha, &sc, &act64, &levr, &nsweeps};
    set_kname(h, "k_sweep64m", (int)h->d.dynamics_id, h->pi_size == 1 ? tname<unsigned char>() : tname<unsigned short>(), rtm);
    HIPCHK(hipLaunchCooperativeKernel(kfn, dim3(sc.nblocks), dim3(256), args, (unsigned)(h->levr_bytes + (size_t)sc.win_bytes), h->stream));
    return PVI_OK;
}
