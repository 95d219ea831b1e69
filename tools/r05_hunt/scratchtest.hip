// Round-5 fault hunt: is a wave's private (scratch) memory really private at full occupancy?  Every lane writes tags derived
// from its global thread id into NB bytes of scratch, works for a while (LDS, barriers, global loads), reads the tags back.
// A mismatch names the thread whose tag was found.  k<NB>: NB bytes of scratch per lane, 80 VGPRs, 512 threads, LDS_BYTES of
// dynamic LDS -- the resources of k_sweep_lean4fb (three workgroups per CU, six waves per SIMD).
// hipcc --offload-arch=gfx950 -O3 -o scratchtest tools/r05_hunt/scratchtest.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int NDW>
__global__ __launch_bounds__(512, 6) void k(unsigned long long* out, const float* __restrict__ in, float* __restrict__ sink, int n, int iters) {
    extern __shared__ float lds[];
    // the space is reserved by a private array the compiler sees; the accesses are the spill instructions themselves
    // (scratch_store_dword / scratch_load_dword off, v, off offset:N), as inline assembly
    volatile unsigned priv[NDW];
    const unsigned gid = blockIdx.x * 512u + threadIdx.x;
#pragma unroll
    for (int j = 0; j < NDW; ++j) priv[j] = 0u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define PUT(J) if (J < NDW) asm volatile("scratch_store_dword off, %0, off offset:" #J "*4" ::"v"(gid * 8u + (unsigned)J) : "memory");
    PUT(0) PUT(1) PUT(2) PUT(3) PUT(4) PUT(5) PUT(6) PUT(7)
    asm volatile("" ::: "v79", "s99");   // (the register budget of the sweep kernel)
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const unsigned i = (blockIdx.x * 977u + threadIdx.x * 13u + it * 7919u) % (unsigned)n;
        const float v = in[i];
        lds[(threadIdx.x + (it & 7) * 512) & 4095] = v;
        __syncthreads();
        acc += lds[(threadIdx.x * 7 + it) & 4095];
        __syncthreads();
    }
    unsigned bad = 0, got = 0, slot = 0;
#define GET(J)                                                                                           \
    if (J < NDW) {                                                                                       \
        unsigned v;                                                                                      \
        asm volatile("scratch_load_dword %0, off, off offset:" #J "*4\n\ts_waitcnt vmcnt(0)" : "=v"(v)::"memory"); \
        if (v != gid * 8u + (unsigned)J && !bad) {                                                       \
            bad = 1;                                                                                     \
            got = v;                                                                                     \
            slot = J;                                                                                    \
        }                                                                                                \
    }
    GET(0) GET(1) GET(2) GET(3) GET(4) GET(5) GET(6) GET(7)
    if (bad) {
        const unsigned long long s = atomicAdd(out, 1ull);
        if (s < 6) {
            out[1 + 2 * s] = ((unsigned long long)gid << 32) | slot;
            out[2 + 2 * s] = got;
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int NDW>
static void run(const float* in, float* sink, int n, unsigned long long* out, int lds_bytes, int blocks, int iters) {
    (void)hipMemset(out, 0, 13 * 8);
    (void)hipFuncSetAttribute((const void*)k<NDW>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(k<NDW>, dim3(blocks), dim3(512), lds_bytes, 0, out, in, sink, n, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[13];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("scratch %2d B/lane, LDS %d B, %d blocks x 10: %llu lanes of %lld read back another value", NDW * 4, lds_bytes, blocks, h[0], 10LL * blocks * 512);
    for (int j = 0; j < 4 && (unsigned long long)j < h[0]; ++j)
        printf("  [thread %llu slot %llu: found %llu = thread %llu slot %llu]", h[1 + 2 * j] >> 32, h[1 + 2 * j] & 0xffffffffu, h[2 + 2 * j], h[2 + 2 * j] / 8, h[2 + 2 * j] % 8);
    printf("  (%s)\n", hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
    const int n = 1 << 24;
    const int lds = argc > 1 ? atoi(argv[1]) : 24544, blocks = argc > 2 ? atoi(argv[2]) : 8528, iters = argc > 3 ? atoi(argv[3]) : 40;
    float *in, *sink;
    unsigned long long* out;
    (void)hipMalloc(&in, n * 4);
    (void)hipMemset(in, 0, n * 4);
    (void)hipMalloc(&sink, 4);
    (void)hipMalloc(&out, 13 * 8);
    run<4>(in, sink, n, out, lds, blocks, iters);
    run<5>(in, sink, n, out, lds, blocks, iters);
    run<4>(in, sink, n, out, lds, blocks, iters);
    run<8>(in, sink, n, out, lds, blocks, iters);
    run<4>(in, sink, n, out, 102400 > 65536 ? 65536 : lds, blocks, iters);   // two workgroups per CU
    return 0;
}
