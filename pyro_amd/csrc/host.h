// Host-side common code of libpyrovi: the structures the kernels take by value (per family), the handle (pvi_problem), error
// and override plumbing, device-memory helpers, and the entry points the translation units call in each other.
#pragma once
#include "core.h"

#define PVI_INTERNAL __attribute__((visibility("hidden")))

#ifdef PVI_TRACE  // (round-5 fault hunt builds only: every device allocation and sweep launch on stderr)
static inline hipError_t pvi_traced_malloc(void** p, size_t n, const char* f, int l) {
    const hipError_t e = hipMalloc(p, n);
    fprintf(stderr, "PVI_TRACE malloc %p %p %zu %s:%d\n", *p, (void*)((char*)*p + n), n, f, l);
    return e;
}
static inline hipError_t pvi_traced_free(void* p, const char* f, int l) {
    fprintf(stderr, "PVI_TRACE free %p %s:%d\n", p, f, l);
    return hipFree(p);
}
#define hipMalloc(p, n) pvi_traced_malloc((void**)(p), (n), __FILE__, __LINE__)
#define hipFree(p) pvi_traced_free((void*)(p), __FILE__, __LINE__)
#endif

// ---- structures the kernel families take by value ----------------------------------------------------------------------
struct Act64 {
    double u0, u1, gu, aok;
};

struct FastP {
    const float4* act;  // [A] {u0, u1, gu*dt, isavalidinput}
    float guard;        // guard band in cells
    int lsplit;
};

struct LeanP {
    int V0, V1;          // velocity plane (DOF=1: V0 = 1, V1 = dim[1]; DOF=2: dim[2] x dim[3])
    int TV0, TV1;        // tile shape in the velocity plane
    int tv1_magic;       // ceil(2^20 / TV1): s / TV1 == (s * magic) >> 20 for s < 1024
    unsigned pd_magic;   // ceil(2^32 / posdim1): z / posdim1 == umulhi(z, magic) (0 when posdim1 == 1)
    unsigned ntx_magic, ntxy_magic;  // same for ntx and ntx*nty
    unsigned nblocks;    // ntx * nty * owned position nodes (1-D grid)
    unsigned xq, xrem;   // nblocks / 8, nblocks % 8: XCD-contiguous block remap
    int xcd_remap;       // 0: identity (experiments)
    int dof1;            // 2-D problems: the tile spans TV0 rows of axis 0 (V0 = owned rows), one position node
    int lsplit;          // log2 lanes per node
    int ntx, nty;        // tiles per plane
    int posdim1;         // dim[1] for DOF=2, 1 for DOF=1
    long long vplane;    // V0 * V1
    long long owned;
    float* ta;           // [DOF][owned]
    float* tB;           // [DOF*M][owned]
    float* gx;           // [owned]  g_x * dt
    unsigned char* flag; // [owned]  bit0 on target, bit1 guard band, bit2 position row in the box
    int2* pt0;           // [dim[0]][dim[DOF]]      {ci or -1, float bits of y}
    int2* pt1;           // [dim[1]][dim[DOF+1]]    (DOF=2)
    int* win;            // [tiles][8] {plo0, phi0, plo1, phi1, vlo0, vhi0, vlo1, vhi1}, inclusive
    float* tbt;          // [tiles][4] tB of the tile when it is the same for all of the tile's nodes
    int dma16;           // 4-D window fill with 16-byte DMA (pitch RS is then a multiple of 4 floats)
    unsigned rs_magic;   // magic32(RS) for the row / column split of a window element index
    int npt;             // nodes per thread of the sweep kernel (1, or 2 for 2-D grids walked uniformly)
    int half;            // nodes per thread pass: thread t owns tile nodes t and (two nodes per thread) t + half
    const float* actc;   // compact per-action constants for scalar loads: whole groups of 4 (+1 group of slack)
    int tb_tile;         // 1: every tile has a uniform tB -> the sweep reads tbt instead of the per-node array
    int* summary;        // [0] max window rows, [1] max row length, [2] some tile has a non-uniform tB, [3] error bits,
                         // [4] largest |displacement| of an in-box cell, in grid cells,
                         // [5] largest |ta| + sum |tB u| (cells): the float32 rounding of these operands is what limits
                         //     the accuracy of the fraction when they cancel
    float guard;
    int lds_floats;      // floats of one window buffer
    int RS;              // LDS row pitch in dwords, 32k+1
};

// PVI_FLAG_F32_FEEDBACK on a 2-D grid (k_sweep_leanfb): its own kernel argument -- LeanP keeps the layout of the kernels that do
// not read it (their code objects are the ones that have run: tools/kernel_manifest.py)
struct LeanFb {
    float* jlo;          // [owned] rounding residual of the stored J
    double alpha64;      // the discount factor of the launch unrounded
};

struct Lean4P {
    int V0, V1;             // velocity plane: dim[2] x dim[3]
    int TV0, TV1;           // rows / columns of the largest tile (diagnostics)
    int ntr;                // tiles per position node (the longest per-row list; shorter ones are padded with empty tiles)
    int posdim1;            // dim[1]
    unsigned ntr_magic, pd_magic;
    long long vplane, owned;
    const int4* tlist;      // [owned rows of axis 0][ntr] {first row, rows, first column, columns} of a tile of the velocity
                            // plane: row pieces cut where the axis-0 corner of THAT row steps, each piece split into columns
                            // so that rows x columns fills the workgroup ({.., 0, .., ..} = padding)
    const unsigned* sched;  // [grid] logical tile id of physical block b, 0xffffffff = padding
    const float2* tsp;      // [2][csize] {integer part, fraction} of the node displacement per velocity axis
    int cs[4];              // compact strides of tsp: 0 on the axes the displacement does not depend on
    long long csize;
    const float* gx;        // [owned] g_x dt, or NULL: summed from gt[]
    const double* gt[4];    // per-axis terms dx_d (Q_dd dx_d) over the levels of axis d (diagonal Q)
    unsigned char* flag;    // [owned] LEAN_FLAG_*
    const int2* pt0;        // [dim0][dim2] position row of axis 0: {lower corner | -1, fraction bits}
    const int2* pt1;        // [dim1][dim3]
    int* win;               // [tiles][8] {plo0, phi0, plo1, phi1, vlo0, vhi0, vlo1, vhi1}
    float* ptab;            // [owned position nodes][ngroups][24]: Pf0[4] Pf1[4] off[4] gudt[4] Pi0[4] Pi1[4]
    int ngroups;            // ceil(A / 4)
    int pcs[2];             // strides (in position nodes) of ptab over (owned row of axis 0, axis 1): 0 = does not depend on it
    int RS;                 // slots (8 bytes) per window row: the longest window row of the tiling rounded up to 4
    int* summary;           // [0] max window rows (pair planes x p1 x j2), [1] longest row, [3] error bits, [4] max pair planes,
                            // [7] some node's cell range does not fit `box`
    const char4* box;       // per owned node: the velocity cells its actions reach, relative to (iv0, iv1) -- set-up only
    const unsigned* vmask;  // [owned] bit a: the cell of (node, action a) lands in the box -- the reference's float64 test, decided at
                            // set-up (A <= 32; NULL: the sweep clamps and compares the cell index instead, sweep_lean4.inc L4_CLAMP)
    float* jlo;             // [owned] PVI_FLAG_F32_FEEDBACK: rounding residual of the stored J of every node (NULL: plain storage)
    double alpha64;         // ... and the discount factor of the launch unrounded (the sweep's own `alpha` is its float32 value)
};

// =================================================================================================
// Bicubic-spline value iteration, 2-D grids only
// (DynamicProgramming2DRectBivariateSpline, dynamicprogramming.py:578-614; the interpolant is
//  scipy RectBivariateSpline(kx=ky=3, s=0) built at discretizer.py:590-612 -- FITPACK regrid/bispev.)
//
// Every sweep: (1) refit the interpolating tensor-product cubic spline through J_k -- separable, one banded
// (2 sub-, 2 super-diagonals) solve per grid line along axis 0, then along axis 1, with LU factors of the
// B-spline collocation matrices computed once on the host (not-a-knot knot vector, as fpregr places the knots
// for s = 0); (2) the Bellman backup with S(x_next) in place of the n-linear interpolant.  x_next is CLAMPED
// to the grid box (fpbisp), there is no zero fill.  All arithmetic is float64 in the reference's order
// (fpbspl recursion; sp += c*hx*hy, i outer, j inner); J storage follows the handle's dtype.
// Restated for the tests in oracle/vi_oracle.py spline_*.
// =================================================================================================
struct SplineP {
    const double* tx;   // [n0+4] knots, axis 0
    const double* ty;   // [n1+4]
    const double* rtx;  // [n0+4][6] reciprocal knot differences used by the basis recursion at interval l
    const double* rty;
    const double* lu0;  // [n0][5] = {l2, l1, 1/d, u1, u2} of the axis-0 collocation matrix
    const double* lu1;  // [n1][5]
    double* work;       // [n0][n1] intermediate of the separable solve
    double* coef;       // [n0][n1] B-spline coefficients
    int n0, n1;
    int chunk0, warm0;  // axis-0 lines are cut into chunks of chunk0 rows; a chunk starts its recurrence warm0
    int chunk1, warm1;  //   rows early from a zero state (the recurrences forget their start like rho^k, rho ~ 0.27)
};

// =================================================================================================
// host side
// =================================================================================================
// (one copy for the library: defined in pyrovi.hip)
extern thread_local char g_err[512];
PVI_INTERNAL int fail(int code, const char* fmt, ...);

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
// (the key list, the table and ovr() itself: pyrovi.hip)
PVI_INTERNAL const char* ovr(const char* key);
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
    const void* lean_lds_attr = nullptr;  // the kernel function that last got hipFuncAttributeMaxDynamicSharedMemorySize (per function, not per handle)
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
    float* jlo = nullptr;     // [owned] PVI_FLAG_F32_FEEDBACK: rounding residual of the stored float32 J of every node
    bool fbcheck = false;     // ... with the corruption detector of the 4-D epilogue (pvi_override("FBCHECK", "1") at create: k_sweep_lean4fbc)
    LeanFb lean_fb = {nullptr, 0.0};  // ... as the 2-D window sweep's kernel argument (set at every launch: launch_sweep_t)
    const double* roll_params = nullptr;  // constants of the continuous closed form (pvi_set_rollout_params)
    SplineP SP;               // bicubic-spline interpolation mode (sweep_spline.inc)
    bool spline = false;
    int multi32 = -1;         // multi-sweep launch of the 2-D float32 sweep (k_sweep_leanm): -1 not decided, 0 no, 1 yes
    int multi64 = -1;         // multi-sweep launch of the float64 sweep (k_sweep64m): -1 not decided, 0 no, 1 yes
    char multi_why[96] = "";
    int multi_rtm = 0;            // ... 1 / 2: in its register-table form, narrow / wide (2-D grids, <= 12 / 24 actions, <= 64 / 512 workgroups)
    int regtab64 = -1;            // the multi-sweep launches keep the per-action cells in registers (2-D, <= 12 actions)
    char kname[128] = "";     // the sweep kernel of the last launch, as a kernel trace prints it (spaces removed): pvi_describe `kernel=`
};

// Name of a kernel template instantiation the way the demangler (rocprofv3's kernel trace) prints it, without spaces:
// "k_sweep64<3,unsignedchar,true,true,true>".  Recorded at every sweep launch, reported by pvi_describe, so that counter
// passes and trace summaries can be tied to the kernel a handle really runs (tools/make_counters_json.py, bench.py).
template <typename T> static inline const char* tname();
template <> inline const char* tname<float>() { return "float"; }
template <> inline const char* tname<double>() { return "double"; }
template <> inline const char* tname<unsigned char>() { return "unsignedchar"; }
template <> inline const char* tname<unsigned short>() { return "unsignedshort"; }
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
static inline double quad_form_host(const double* M, const double* dx, int n) {
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

// the residuals of error-feedback storage (PVI_FLAG_F32_FEEDBACK) belong to the handle (pvi_problem::jlo); the parameter blocks
// of the window sweeps get the pointer at every launch (launch_sweep_t)
static inline float*& jlo_of(pvi_problem* h) { return h->jlo; }

static inline void dev_release(pvi_problem* h, void* p) {
    if (!p) return;
    for (auto& q : h->dev_allocs)
        if (q == p) q = nullptr;
    (void)hipFree(p);
}

// ---- entry points across the translation units --------------------------------------------------------------------------
// pyrovi.hip
PVI_INTERNAL int launch_sweep(pvi_problem* h, int src, double alpha, hipStream_t st, int k, double tol, int deferred = 0);
// lean.hip: float32 LDS-window families (2-D k_sweep_lean, 4-D k_sweep_lean4) and the plain-gather k_sweep_fast
PVI_INTERNAL int lean_setup(pvi_problem* h);
PVI_INTERNAL int launch_lean4(pvi_problem* h, const float* Jin, float* Jout, float alpha, hipStream_t st, SweepCtl sc);
PVI_INTERNAL int launch_lean2(pvi_problem* h, const float* Jin, float* Jout, float alpha, hipStream_t st, SweepCtl sc);
PVI_INTERNAL int launch_fast(pvi_problem* h, const float* Jin, float* Jout, float alpha, hipStream_t st, SweepCtl sc);
// f64.hip: float64 second form (k_sweep64, k_sweep64m) and the validity masks of the sparse walks
PVI_INTERNAL int launch_f64v2(pvi_problem* h, const double* Jin, double* Jout, double alpha, hipStream_t st, SweepCtl sc);
PVI_INTERNAL int launch_valid_mask(pvi_problem* h, uint4* vm, unsigned long long* cnt);
PVI_INTERNAL bool multi32_applies(pvi_problem* h);   // lean.hip: k_sweep_leanm
PVI_INTERNAL int launch_multi32(pvi_problem* h, int src, double alpha, double tol, int nsweeps);
PVI_INTERNAL bool multi64_applies(pvi_problem* h);
PVI_INTERNAL int launch_multi64(pvi_problem* h, int src, double alpha, double tol, int nsweeps);
