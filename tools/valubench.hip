// VALU issue-rate microbenchmark (gfx950): clocks per wave64 instruction per SIMD for the op kinds of the sweep loop.
// hipcc --offload-arch=gfx950 -O3 tools/valubench.hip -o /tmp/vb && /tmp/vb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    const int lane = threadIdx.x & 63;
    float a[8];
    v2f p[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (float)(threadIdx.x + i) * 0.001f;
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = (v2f){a[2 * i], a[2 * i + 1]};
    int ia = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {  // 8 independent chains x 8 = 64 instrs of the kind under test per iteration
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) a[i] = __builtin_fmaf(a[i], s, 0.5f);                      // v_fma_f32
                if (MODE == 1 && i < 4) p[i] = __builtin_elementwise_fma(p[i], (v2f){s, s}, (v2f){0.5f, 0.5f});  // v_pk_fma_f32 (4 per u)
                if (MODE == 2) a[i] = __builtin_amdgcn_fmed3f(a[i], s, 0.75f) + 0.0f;         // v_med3 (+ may fold)
                if (MODE == 3) a[i] = __builtin_floorf(a[i] * s);                              // v_mul + v_floor
                if (MODE == 4) a[i] = (float)(unsigned)(a[i]) + s;                             // v_cvt_u32 + v_cvt_f32 + add
                if (MODE == 5) a[i] = (a[i] < s) ? a[(i + 1) & 7] : a[i];                      // v_cmp + v_cndmask
                if (MODE == 6) a[i] += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[(i + 1) & 7]), (it + i) & 63));  // readlane + add
            }
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += a[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + (float)ia + (float)lane;
}

template <int MODE>
void run(const char* name, double instr_per_iter, int wpb) {
    float* d;
    (void)hipMalloc(&d, 2048 * 1024 * 4);
    const int iters = 4000;
    for (int blocks_per_cu = 1; blocks_per_cu <= 8; blocks_per_cu *= 2) {
        const int blocks = 256 * blocks_per_cu;
        hipEvent_t a, b;
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        k<MODE><<<blocks, 64 * wpb>>>(d, 10, 0.999f);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        k<MODE><<<blocks, 64 * wpb>>>(d, iters, 0.999f);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        const double waves_per_simd = blocks_per_cu * wpb / 4.0;
        const double instr_per_simd = waves_per_simd * iters * instr_per_iter;
        printf("%-28s waves/SIMD %4.1f  %.3f ms  -> %.2f clk @2.4GHz per counted wave-instr per SIMD\n", name, waves_per_simd, ms,
               ms * 1e-3 * 2.4e9 / instr_per_simd);
    }
    (void)hipFree(d);
}

// VALU and LDS work in one loop: does the time add up or overlap?  NF fmas (8 chains) + NL ds_read2_b32 per iteration
template <int NF, int NL>
__global__ __launch_bounds__(256) void kmix(float* out, int iters, float s) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i * 1e-3f;
    __syncthreads();
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (float)(threadIdx.x + i) * 0.001f;
    float acc = 0.f;
    unsigned adr = (threadIdx.x & 63) * 4u;
    for (int it = 0; it < iters; ++it) {
        const float* p = (const float*)((const char*)lds + ((adr + (unsigned)it * 4u) & 0x1fffu));
        float l[NL > 0 ? 2 * NL : 1];
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            l[2 * u] = p[u * 67];
            l[2 * u + 1] = p[u * 67 + 1];
        }
#pragma unroll
        for (int u = 0; u < NF / 8; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], s, 0.5f);
#pragma unroll
        for (int u = 0; u < NL; ++u) acc += l[2 * u] * l[2 * u + 1];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int NF, int NL>
void runmix() {
    float* d;
    (void)hipMalloc(&d, 2048 * 1024 * 4);
    const int iters = 4000, blocks = 256 * 8;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    kmix<NF, NL><<<blocks, 256>>>(d, 10, 0.999f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    kmix<NF, NL><<<blocks, 256>>>(d, iters, 0.999f);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("mix: %3d v_fma + %2d ds_read2_b32 (+2 VALU each) per iteration, 8 waves/SIMD: %.1f clk @2.4GHz per iteration per SIMD-wave\n",
           NF, NL, ms * 1e-3 * 2.4e9 / (8.0 * iters));
    (void)hipFree(d);
}

// scalar-ALU issue cost: 64 independent-ish s_add_i32 per iteration next to NF v_fma (does scalar work add as well?)
template <int NF>
__global__ __launch_bounds__(256) void ksalu(float* out, int iters, float s, int seed) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (float)(threadIdx.x + i) * 0.001f;
    int r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            asm volatile("s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 3\n s_add_i32 %2, %2, 5\n s_add_i32 %3, %3, 7"
                         : "+s"(r0), "+s"(r1), "+s"(r2), "+s"(r3));
#pragma unroll
        for (int u = 0; u < NF / 8; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], s, 0.5f);
    }
    float acc = (float)(r0 + r1 + r2 + r3);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int NF>
void runsalu() {
    float* d;
    (void)hipMalloc(&d, 2048 * 1024 * 4);
    const int iters = 4000, blocks = 256 * 8;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    ksalu<NF><<<blocks, 256>>>(d, 10, 0.999f, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    ksalu<NF><<<blocks, 256>>>(d, iters, 0.999f, 1);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("salu: 64 s_add_i32 + %3d v_fma per iteration, 8 waves/SIMD: %.1f clk @2.4GHz per iteration per SIMD-wave\n", NF,
           ms * 1e-3 * 2.4e9 / (8.0 * iters));
    (void)hipFree(d);
}

int main() {
    runsalu<0>();
    runsalu<64>();
    runsalu<128>();
    runmix<64, 0>();
    runmix<0, 8>();
    runmix<64, 8>();
    runmix<128, 8>();
    runmix<64, 16>();
    runmix<8, 16>();
    run<0>("v_fma_f32", 64, 4);
    run<1>("v_pk_fma_f32", 32, 4);
    run<2>("v_med3_f32(+add?)", 64, 4);
    run<3>("v_mul+v_floor (2 ops)", 64, 4);
    run<4>("cvt_u32+cvt_f32+add (3 ops)", 64, 4);
    run<5>("v_cmp+v_cndmask (2 ops)", 64, 4);
    run<6>("v_readlane+v_add (2 ops)", 64, 4);
    return 0;
}
