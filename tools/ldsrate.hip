// LDS read-rate microbenchmark (gfx950), LDS-bound (one VALU instruction per read): ds_read_b64, ds_read_b128 at 16-byte
// aligned addresses and ds_read_b128 at addresses that are only 8-byte aligned (does the hardware take them, at what rate?).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ldsrate tools/ldsrate.hip && /tmp/ldsrate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// MODE 0: 16 x ds_read_b64 of consecutive 8-byte slots; 1: 8 x ds_read_b128, 16-byte aligned; 2: 8 x ds_read_b128 at
// 8-byte-aligned-only addresses (lane i reads slots i+1, i+2 of its row: overlapping its neighbours' like the sweep's pairs);
// 3: 8 x ds_read_b128 lane i reads slots i, i+1 (16-byte aligned for even lanes only)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, int check) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    unsigned base;
    if (MODE == 0) base = (unsigned)(wave * 1024 + lane * 2) * 4u;            // slot = lane (8 B each)
    if (MODE == 1) base = (unsigned)(wave * 1024 + lane * 4) * 4u;            // 16 B per lane, aligned
    if (MODE == 2) base = (unsigned)(wave * 1024 + lane * 2 + 2) * 4u + 0u;   // slot lane + 1: 8-byte aligned, odd/even mix
    if (MODE == 3) base = (unsigned)(wave * 1024 + lane * 2) * 4u;
    if (MODE == 2) base += 8u * (lane & 0);                                  // (kept simple: slots lane+1 .. lane+2)
    for (int it = 0; it < iters; ++it) {
        const unsigned a = base + (unsigned)(it & 3) * 512u * 0u;
        if (MODE == 0) {
            v2f r[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[u]) : "v"(a), "n"(u * 512));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                         "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]));
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += r[u].x;
        } else {
            v4f r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[u]) : "v"(a), "n"(u * 1024 + (MODE == 1 ? 0 : 0)));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += r[u].x + r[u].z;
            if (check && it == 0) {
                // expected: floats at (base/4 + u*256) + {0,1,2,3}
                bool ok = true;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float e = (float)(base / 4 + u * 256);
                    ok = ok && r[u].x == e && r[u].y == e + 1 && r[u].z == e + 2 && r[u].w == e + 3;
                }
                if (!ok) acc = -1e30f;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
static void run(const char* name, int reads_per_iter, int bytes_per_read) {
    float* d;
    hipMalloc(&d, 4096 * 256 * 4);
    float* hst = (float*)malloc(256 * 4);
    const int iters = 4000;
    for (int bpc : {1, 2, 4}) {
        const int blocks = 256 * bpc;
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        k<MODE><<<blocks, 256, 36864>>>(d, 1, 1);
        hipMemcpy(hst, d, 256 * 4, hipMemcpyDeviceToHost);
        bool ok = true;
        for (int i = 0; i < 256; ++i) ok = ok && hst[i] > -1e29f;
        hipEventRecord(a);
        k<MODE><<<blocks, 256, 36864>>>(d, iters, 0);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double instr_per_cu = (double)bpc * 4 * iters * reads_per_iter;
        const double clk = ms * 1e-3 * 2.4e9 / instr_per_cu;
        printf("%-34s waves/SIMD %d  values %s  %.2f clk per wave-instruction per CU = %.0f B/clk/CU\n", name, bpc, ok ? "ok" : "WRONG", clk,
               64.0 * bytes_per_read / clk);
    }
    hipFree(d);
}

int main() {
    run<0>("ds_read_b64, consecutive slots", 16, 8);
    run<1>("ds_read_b128, 16-byte aligned", 8, 16);
    run<2>("ds_read_b128, 8-byte aligned (odd)", 8, 16);
    run<3>("ds_read_b128, stride 8 (pairs)", 8, 16);
    return 0;
}
