// TEST INFRASTRUCTURE -- not part of the product, never loaded by it.
//
// A source-level emulation of the HIP programming model on the CPU, just large enough for pyro_amd/csrc/*: the kernels of
// libpyrovi are compiled as ordinary C++ (tests/emu/build_emu.py) against THIS header instead of <hip/hip_runtime.h>, and
// tests/test_emu_cpu.py runs them through the same C ABI on small grids against the oracle.  What that can show: that the
// kernel SOURCE and the host logic around it compute the reference's recursion (index arithmetic, tile schedules, LDS window
// addressing, reductions, stop test, refusals).  What it cannot show: anything about the gfx950 code objects (register
// allocation, hazards, occupancy, timing) -- that is what tests -m gpu and profiles/verified_kernels.json are for.
//
// Execution model (tests/emu/emu_runtime.cpp): a launch runs its workgroups one after the other (a few at a time on host
// threads); every work-item of a workgroup is a fiber with its own stack.  A fiber runs until it reaches __syncthreads() or a
// wave-collective operation (shuffle, DPP, ballot, readlane ...), where it parks; the scheduler completes the operation when
// every live lane of the wave (every live work-item of the workgroup) has arrived -- or, when nothing else can run, with the
// lanes that did arrive as the active set (divergent control flow).  Work-items between two such points run one after the
// other, NOT in lockstep: code that lets lanes of a wave communicate through LDS without a barrier is not modelled.
// Streams are synchronous, there is one device, cooperative launches are reported as unsupported.
#pragma once
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>
#include <utility>

#define PVI_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local   /* one host thread runs one workgroup at a time: per-thread = per-workgroup */
#define __constant__ static

// ---- vector types ------------------------------------------------------------------------------------------------------------
struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
#define EMU_VEC2(T, N) \
    struct N { T x, y; }; \
    static inline N make_##N(T x, T y) { return N{x, y}; }
#define EMU_VEC4(T, N) \
    struct alignas(sizeof(T) * 4 > 16 ? 16 : sizeof(T) * 4) N { T x, y, z, w; }; \
    static inline N make_##N(T x, T y, T z, T w) { return N{x, y, z, w}; }
struct alignas(8) float2 { float x, y; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
struct alignas(8) int2 { int x, y; };
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
struct alignas(8) uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
struct alignas(16) double2 { double x, y; };
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
struct alignas(16) double4 { double x, y, z, w; };
static inline double4 make_double4(double x, double y, double z, double w) { return double4{x, y, z, w}; }
EMU_VEC4(float, float4)
EMU_VEC4(int, int4)
EMU_VEC4(unsigned, uint4)
EMU_VEC4(char, char4)
EMU_VEC4(unsigned char, uchar4)

// ---- the engine's interface -----------------------------------------------------------------------------------------------------
namespace emu {
struct Idx3 { unsigned x, y, z; };
struct Fiber;
struct BlockCtx {
    Idx3 bid, bdim, gdim;
    char* dyn_lds;        // the workgroup's dynamic LDS, at a virtual address below 4 GiB (32-bit "LDS addresses" are host pointers)
    unsigned dyn_bytes;
};
struct FiberPub {          // what a work-item's code can see of itself
    Idx3 tid;
    unsigned flat, lane, wave;
};
extern thread_local FiberPub* g_me;
extern thread_local BlockCtx* g_blk;
enum Op { OP_SHFL, OP_SHFL_XOR, OP_BALLOT, OP_READLANE, OP_READFIRST, OP_DPP, OP_SYNC };
unsigned long long collective(Op op, unsigned long long val, unsigned long long aux, unsigned p0 = 0, unsigned p1 = 0, unsigned p2 = 0,
                              unsigned p3 = 0);
void barrier();
void launch(dim3 grid, dim3 block, size_t dyn_lds_bytes, const std::function<void()>& body, const char* name);
void yield_host();
// LDS access trace (build_emu.py --ldstrace rewrites the gathers of the 4-D window sweep into emu::lds_rd): bank-conflict counts
// of the production kernel's own addresses under the bank model of tools/lds_conflict_model4.py
void lds_note(unsigned byte_adr);
template <class T>
static inline T lds_rd(unsigned byte_adr) {
    lds_note(byte_adr);
    return *(const T*)(size_t)byte_adr;
}
// Cooperative launches go through a function POINTER and an array of argument addresses: build_emu.py rewrites every
// `(const void*)kernel<...>` of the sources into emu::coop_thunk(&kernel<...>), which records how to call that kernel.
typedef void (*Invoker)(const void* fn, void** args);
void register_kernel(const void* fn, Invoker inv);
template <class... A>
struct Thunk {
    template <size_t... I>
    static void go(void (*fn)(A...), void** args, std::index_sequence<I...>) { fn(*(typename std::remove_reference<A>::type*)args[I]...); }
    static void call(const void* fn, void** args) { go((void (*)(A...))fn, args, std::index_sequence_for<A...>{}); }
};
template <class... A>
static inline const void* coop_thunk(void (*fn)(A...)) {
    register_kernel((const void*)fn, &Thunk<A...>::call);
    return (const void*)fn;
}
}  // namespace emu

#define threadIdx (emu::g_me->tid)
#define blockIdx (emu::g_blk->bid)
#define blockDim (emu::g_blk->bdim)
#define gridDim (emu::g_blk->gdim)
#define warpSize 64

static inline void __syncthreads() { emu::barrier(); }
static inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }
static inline void __threadfence_block() { std::atomic_thread_fence(std::memory_order_seq_cst); }

// ---- bit casts, integer helpers ---------------------------------------------------------------------------------------------
template <class To, class From>
static inline To emu_bit(From f) {
    static_assert(sizeof(To) == sizeof(From), "bit cast");
    To t;
    memcpy(&t, &f, sizeof(t));
    return t;
}
static inline int __float_as_int(float f) { return emu_bit<int>(f); }
static inline unsigned __float_as_uint(float f) { return emu_bit<unsigned>(f); }
static inline float __int_as_float(int i) { return emu_bit<float>(i); }
static inline float __uint_as_float(unsigned i) { return emu_bit<float>(i); }
static inline double __longlong_as_double(long long v) { return emu_bit<double>(v); }
static inline long long __double_as_longlong(double v) { return emu_bit<long long>(v); }
static inline int __double2loint(double d) { return (int)(emu_bit<unsigned long long>(d) & 0xffffffffull); }
static inline int __double2hiint(double d) { return (int)(emu_bit<unsigned long long>(d) >> 32); }
static inline double __hiloint2double(int hi, int lo) {
    return emu_bit<double>(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
template <class T>
static inline T min(T a, T b) { return b < a ? b : a; }
template <class T>
static inline T max(T a, T b) { return a < b ? b : a; }
static inline double min(double a, double b) { return std::fmin(a, b); }
static inline double max(double a, double b) { return std::fmax(a, b); }
static inline float min(float a, float b) { return std::fmin(a, b); }
static inline float max(float a, float b) { return std::fmax(a, b); }
using std::floor;
using std::fabs;
using std::sqrt;
using std::sin;
using std::cos;
using std::tan;
using std::fma;
using std::isfinite;
using std::isnan;

// ---- AMD builtins the kernels spell out (renamed: the compiler for x86 does not know them) ------------------------------------
static inline float emu_fmed3f(float a, float b, float c) {  // v_med3_f32: the median of three
    return std::fmax(std::fmin(a, b), std::fmin(std::fmax(a, b), c));
}
static inline int emu_sbfe(int v, unsigned off, unsigned width) {  // v_bfe_i32
    off &= 31u;
    width &= 31u;
    if (!width) return 0;
    const unsigned u = ((unsigned)v >> off) & ((width >= 32 ? 0u : (1u << width)) - 1u);
    const unsigned sign = 1u << (width - 1);
    return (int)((u ^ sign) - sign);
}
#define __builtin_amdgcn_fmed3f emu_fmed3f
#define __builtin_amdgcn_sbfe emu_sbfe
#define __builtin_amdgcn_s_sleep(n) emu::yield_host()
#define __builtin_amdgcn_fence(order, scope) std::atomic_thread_fence(std::memory_order_seq_cst)

// wave collectives
template <class T>
static inline T emu_coll(emu::Op op, T v, unsigned p0 = 0, unsigned p1 = 0, unsigned p2 = 0, unsigned p3 = 0, unsigned long long aux = 0) {
    static_assert(sizeof(T) <= 8, "collective operand");
    unsigned long long bits = 0;
    memcpy(&bits, &v, sizeof(T));
    const unsigned long long r = emu::collective(op, bits, aux, p0, p1, p2, p3);
    T out;
    memcpy(&out, &r, sizeof(T));
    return out;
}
template <class T>
static inline T __shfl_xor(T v, int mask, int width = 64) { return emu_coll(emu::OP_SHFL_XOR, v, (unsigned)mask, (unsigned)width); }
template <class T>
static inline T __shfl(T v, int lane, int width = 64) { return emu_coll(emu::OP_SHFL, v, (unsigned)lane, (unsigned)width); }
static inline unsigned long long emu_ballot(bool p) { return emu::collective(emu::OP_BALLOT, p ? 1ull : 0ull, 0); }
static inline unsigned long long emu_active() { return emu::collective(emu::OP_BALLOT, 1ull, 0); }
static inline int emu_any(bool p) { return emu_ballot(p) != 0ull; }
static inline int emu_all(bool p) { return emu_ballot(!p) == 0ull; }
#define __any(p) emu_any((bool)(p))
#define __all(p) emu_all((bool)(p))
#define __builtin_amdgcn_ballot_w64(p) emu_ballot((bool)(p))
#define __ballot(p) emu_ballot((bool)(p))
static inline int emu_readlane(int v, int lane) { return emu_coll(emu::OP_READLANE, v, (unsigned)lane); }
static inline int emu_readfirstlane(int v) { return emu_coll(emu::OP_READFIRST, v); }
#define __builtin_amdgcn_readlane emu_readlane
#define __builtin_amdgcn_readfirstlane emu_readfirstlane
static inline int emu_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    unsigned long long o = (unsigned)old;
    return emu_coll(emu::OP_DPP, src, (unsigned)ctrl, (unsigned)row_mask, (unsigned)bank_mask, bound_ctrl ? 1u : 0u, o);
}
#define __builtin_amdgcn_update_dpp emu_update_dpp

// buffer resources: raw buffers with the range check of the hardware (an offset at or beyond num_records reads zeros)
struct __amdgpu_buffer_rsrc_t {
    const char* base;
    unsigned num_records;
};
static inline __amdgpu_buffer_rsrc_t emu_make_rsrc(void* p, short /*stride*/, unsigned num, int /*flags*/) {
    return __amdgpu_buffer_rsrc_t{(const char*)p, num};
}
#define __builtin_amdgcn_make_buffer_rsrc emu_make_rsrc
typedef unsigned emu_u4 __attribute__((ext_vector_type(4)));
static inline unsigned emu_buf_b32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, int /*aux*/) {
    if ((unsigned long long)voff + 4ull > (unsigned long long)r.num_records) return 0u;
    unsigned v;
    memcpy(&v, r.base + (size_t)voff + (size_t)soff, 4);
    return v;
}
static inline emu_u4 emu_buf_b128(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, int /*aux*/) {
    emu_u4 v = {0u, 0u, 0u, 0u};
    if ((unsigned long long)voff + 16ull > (unsigned long long)r.num_records) return v;
    memcpy(&v, r.base + (size_t)voff + (size_t)soff, 16);
    return v;
}
#define __builtin_amdgcn_raw_buffer_load_b32 emu_buf_b32
#define __builtin_amdgcn_raw_buffer_load_b128 emu_buf_b128
// global -> LDS DMA: every lane moves `size` bytes from ITS global pointer to (wave-uniform LDS base) + offset + lane * size
static inline void emu_global_load_lds(const void* g, void* lds, unsigned size, unsigned offset, unsigned /*aux*/) {
    memcpy((char*)lds + offset + (size_t)emu::g_me->lane * size, g, size);
}
#define __builtin_amdgcn_global_load_lds emu_global_load_lds

// ---- atomics -------------------------------------------------------------------------------------------------------------------
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __HIP_MEMORY_SCOPE_SYSTEM 0
#define __HIP_MEMORY_SCOPE_WORKGROUP 0
#define __hip_atomic_load(p, order, scope) __atomic_load_n(p, __ATOMIC_SEQ_CST)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n(p, v, __ATOMIC_SEQ_CST)
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST)
template <class T>
static inline T emu_fetch_max(T* p, T v) {
    T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
    }
    return old;
}
template <class T>
static inline T emu_fetch_min(T* p, T v) {
    T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
    }
    return old;
}
#define __hip_atomic_fetch_max(p, v, order, scope) emu_fetch_max(p, v)
// The lanes of a wave issue an atomic TOGETHER: when lanes 0..2 publish three statistics and lane 0 then takes a ticket, the three
// atomics have been issued before the ticket (the kernels rely on it: block_stats + sweep_finish).  Work-items run one after the
// other here, so every atomic is followed by a rendezvous of the lanes that execute it.
static inline void emu_wave_sync() { (void)emu::collective(emu::OP_SYNC, 0ull, 0ull); }
template <class T, class U>
static inline T atomicMax(T* p, U v) { T r = emu_fetch_max(p, (T)v); emu_wave_sync(); return r; }
template <class T, class U>
static inline T atomicMin(T* p, U v) { T r = emu_fetch_min(p, (T)v); emu_wave_sync(); return r; }
template <class T, class U>
static inline T atomicOr(T* p, U v) { T r = __atomic_fetch_or(p, (T)v, __ATOMIC_SEQ_CST); emu_wave_sync(); return r; }
template <class T, class U>
static inline T atomicAnd(T* p, U v) { T r = __atomic_fetch_and(p, (T)v, __ATOMIC_SEQ_CST); emu_wave_sync(); return r; }
template <class T, class U>
static inline T atomicAdd(T* p, U v) { T r = __atomic_fetch_add(p, (T)v, __ATOMIC_SEQ_CST); emu_wave_sync(); return r; }

// ---- runtime API (one device, synchronous streams) -----------------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
typedef struct emu_stream* hipStream_t;
typedef struct emu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 16, hipDeviceAttributeCooperativeLaunch = 95 };
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipMalloc(void** p, size_t n);
template <class T>
static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
hipError_t hipFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st = nullptr);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st = nullptr);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int dev);
hipError_t hipFuncSetAttribute(const void* fn, hipFuncAttribute a, int v);
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void* fn, int block, size_t lds);
hipError_t hipLaunchCooperativeKernel(const void* fn, dim3 grid, dim3 block, void** args, unsigned lds, hipStream_t st);

#define EMU_STR2(x) #x
#define EMU_STR(x) EMU_STR2(x)
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    emu::launch(dim3(grid), dim3(block), (size_t)(lds), [=]() { kernel(__VA_ARGS__); }, EMU_STR(kernel))
