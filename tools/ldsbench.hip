// LDS read-pattern microbenchmark (gfx950): cycles per wave-instruction for the gather shapes the sweep can use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, int stride) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 9216; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    unsigned base = (unsigned)(lane * stride) * 4u;
    for (int it = 0; it < iters; ++it) {
        // ONE address per iteration, 8 LDS instructions at immediate offsets (rows 520 B apart), 8-16 adds
        const char* p = (const char*)lds + ((base + (unsigned)it * 64u) & 0x1fffu);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {  // ds_read2_b32 (j, j+1), unaligned pair
                const float* q = (const float*)(p + 4 + u * 520);
                acc += q[0] * q[1];
            } else if (MODE == 1) {  // ds_read2_b32 with far-apart halves (two rows)
                const float* q = (const float*)(p + 4 + u * 520);
                acc += q[0] * q[33];
            } else if (MODE == 2) {  // ds_read_b64 aligned
                const v2f q = *(const v2f*)(p + u * 520);
                acc += q.x * q.y;
            } else if (MODE == 3) {  // ds_read_b128 aligned
                const v4f q = *(const v4f*)(p + u * 528);
                acc += q.x * q.y + q.z * q.w;
            } else if (MODE == 4) {  // single ds_read_b32
                acc += *(const float*)(p + 4 + u * 520);
            } else if (MODE == 5) {  // two ds_read_b32 (volatile: not merged into ds_read2_b32)
                const volatile float* q = (const volatile float*)(p + 4 + u * 520);
                acc += q[0] * q[1];
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static int LDSB = 65536;
template <int MODE>
void run(const char* name, int stride) {
    float* d;
    hipMalloc(&d, 256 * 1024 * 4);
    const int iters = 2000, blocks = 1024;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256, LDSB>>>(d, 10, stride);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<blocks, 256, LDSB>>>(d, iters, stride);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per CU: blocks/256 CUs * 4 waves * iters*8 instrs
    const double instr_per_cu = (double)blocks / 256 * 4 * iters * 8;
    const double clk = ms * 1e-3 * 2.4e9;
    printf("%-34s stride %2d dwords: %.2f clk per wave-instruction per CU (%.3f ms)\n", name, stride, clk / instr_per_cu, ms);
    hipFree(d);
}

int main() {
    for (int stride : {1, 2}) {
        run<0>("ds_read2_b32 (j,j+1) unaligned", stride);
        run<1>("ds_read2_b32 (j, j+33 dwords)", stride);
        run<2>("ds_read_b64 aligned", stride * 2);
        run<4>("ds_read_b32", stride);
    }
    run<5>("2 x ds_read_b32 volatile (j,j+1)", 1);
    run<3>("ds_read_b128 aligned", 4);
    LDSB = 36864;  // 4 workgroups per CU -> 4 waves per SIMD
    printf("--- 4 waves per SIMD\n");
    run<0>("ds_read2_b32 (j,j+1) unaligned", 1);
    run<5>("2 x ds_read_b32 volatile (j,j+1)", 1);
    run<2>("ds_read_b64 aligned", 2);
    run<4>("ds_read_b32", 1);
    run<3>("ds_read_b128 aligned", 4);
    run<0>("ds_read2_b32 stride 65", 65);
    run<4>("ds_read_b32 stride 65", 65);
    return 0;
}
