// Round-5 fault hunt: does a value parked in a HIGH scalar register pair survive a busy kernel?  (The e1 build of the
// error-feedback sweep kept a wave mask in s[98:99] across its action loops and computed garbage; the ISA never touches the pair
// in between.)  Every wave writes a tag into s[LO:LO+1] by inline assembly, runs loads / LDS traffic / barriers / scratch,
// reads the pair back and counts mismatches.  hipcc --offload-arch=gfx950 -O3 -o sgprtest tools/r05_hunt/sgprtest.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define STR2(x) #x
#define STR(x) STR2(x)
template <int LO>
struct Reg;
#define REGPAIR(LO, HI)                                                                                               \
    template <> struct Reg<LO> {                                                                                      \
        static __device__ __forceinline__ void put(unsigned long long v) {                                            \
            asm volatile("s_mov_b64 s[" STR(LO) ":" STR(HI) "], %0" ::"s"(v) : "s" STR(LO), "s" STR(HI));               \
        }                                                                                                             \
        static __device__ __forceinline__ unsigned long long get() {                                                  \
            unsigned long long r;                                                                                     \
            asm volatile("s_mov_b64 %0, s[" STR(LO) ":" STR(HI) "]" : "=s"(r));                                         \
            return r;                                                                                                 \
        }                                                                                                             \
    };
REGPAIR(40, 41) REGPAIR(88, 89) REGPAIR(90, 91) REGPAIR(94, 95) REGPAIR(96, 97) REGPAIR(98, 99) REGPAIR(100, 101)

template <int LO>
__global__ __launch_bounds__(512, 6) void k(unsigned long long* out, const float* __restrict__ in, float* __restrict__ sink, int n, int iters) {
    extern __shared__ float lds[];
    const unsigned long long tag = 0x5a5a000000000000ull | ((unsigned long long)blockIdx.x << 16) | threadIdx.x >> 6;
    const unsigned long long utag = __builtin_amdgcn_readfirstlane((unsigned)tag) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(tag >> 32)) << 32);
    Reg<LO>::put(utag);
    volatile float priv[24];   // scratch
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const unsigned i = (blockIdx.x * 977u + threadIdx.x * 13u + it * 7919u) % (unsigned)n;
        const float v = in[i];
        lds[threadIdx.x + (it & 7) * 512] = v;
        priv[(it + threadIdx.x) % 24] = v;
        __syncthreads();
        acc += lds[(threadIdx.x * 7 + it) & 4095] + priv[(it * 5 + 3) % 24];
        __syncthreads();
    }
    const unsigned long long back = Reg<LO>::get();
    if (back != utag && (threadIdx.x & 63) == 0) {
        const unsigned long long slot = atomicAdd(out, 1ull);
        if (slot < 8) {
            out[1 + 2 * slot] = utag;
            out[2 + 2 * slot] = back;
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int LO>
static void run(const float* in, float* sink, int n, unsigned long long* out) {
    hipMemset(out, 0, 17 * 8);
    hipFuncSetAttribute((const void*)k<LO>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(k<LO>, dim3(20000), dim3(512), 43264, 0, out, in, sink, n, 60);
    hipDeviceSynchronize();
    unsigned long long h[17];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("s[%d:%d]: %llu waves of %d lost the value", LO, LO + 1, h[0], 20 * 20000 * 8);
    for (int j = 0; j < 3 && (unsigned long long)j < h[0]; ++j) printf("  [put %016llx got %016llx]", h[1 + 2 * j], h[2 + 2 * j]);
    printf("  (%s)\n", hipGetErrorString(hipGetLastError()));
}

int main() {
    const int n = 1 << 24;
    float *in, *sink;
    unsigned long long* out;
    hipMalloc(&in, n * 4);
    hipMemset(in, 0, n * 4);
    hipMalloc(&sink, 4);
    hipMalloc(&out, 17 * 8);
    run<40>(in, sink, n, out);
    run<88>(in, sink, n, out);
    run<90>(in, sink, n, out);
    run<94>(in, sink, n, out);
    run<96>(in, sink, n, out);
    run<98>(in, sink, n, out);
    run<100>(in, sink, n, out);
    return 0;
}
