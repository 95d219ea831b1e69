// LDS bank-conflict microbenchmark for ds_read_b64 / ds_write_b64 on gfx950: 16 operations in flight per wave and wait, 16 waves
// per CU, so that the conflict-free rate of the LDS (2 cycles per ds_read_b64) is reached and a second cycle shows.
// (tools/ldsgather.hip issues 8 volatile reads per wait and stays at 4.4 cycles per read whatever the pattern.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ldsbank tools/ldsbank.hip && /tmp/ldsbank
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

// every lane reads (WRITE = 0) or writes (1) the 8-byte slot slot[lane] + 0 .. 15 rows of pitch RS
template <int WRITE>
__global__ __launch_bounds__(256) void k(const int* __restrict__ slot, float* out, int iters, int RS) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 12288; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const unsigned base = (unsigned)slot[threadIdx.x & 63] * 8u;
    const unsigned rowb = (unsigned)RS * 8u;
    v2f acc = {0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        v2f q[16];
        if (WRITE) {
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("ds_write_b64 %0, %1" ::"v"(base + (unsigned)u * rowb), "v"(acc));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("ds_read_b64 %0, %1" : "=v"(q[u]) : "v"(base + (unsigned)u * rowb));
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]), "+v"(q[8]), "+v"(q[9]),
                           "+v"(q[10]), "+v"(q[11]), "+v"(q[12]), "+v"(q[13]), "+v"(q[14]), "+v"(q[15]));
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += q[u];
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y;
}

static int* d_slot;
static float* d_out;
template <int WRITE>
static void run(const char* name, const std::vector<int>& s, int RS) {
    (void)hipMemcpy(d_slot, s.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
    const int iters = 1000, blocks = 2048;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    k<WRITE><<<blocks, 256, 49152>>>(d_slot, d_out, 10, RS);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    k<WRITE><<<blocks, 256, 49152>>>(d_slot, d_out, iters, RS);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    const double instr_per_cu = (double)blocks / 256 * 4 * iters * 16;
    printf("%-6s %-58s : %.2f clk per wave-instruction per CU (at 2.4 GHz)\n", WRITE ? "write" : "read", name, ms * 1e-3 * 2.4e9 / instr_per_cu);
}

// lanes = nodes of a tile in row-major order, ncols per row, window pitch rs; `step` >= 0: lanes from `step` on are displaced by
// dcol columns / drow rows (the floor of the gather displacement steps inside the wave)
static std::vector<int> tile(int ncols, int rs, int step = -1, int dcol = 0, int drow = 0) {
    std::vector<int> s(64);
    for (int l = 0; l < 64; ++l) {
        s[l] = (l / ncols) * rs + l % ncols + 4;
        if (step >= 0 && l >= step) s[l] += dcol + drow * rs;
    }
    return s;
}

int main() {
    (void)hipMalloc(&d_slot, 64 * sizeof(int));
    (void)hipMalloc(&d_out, 2048 * 256 * 4);
    char nm[128];
    run<0>("64 consecutive slots", tile(64, 64), 64);
    for (int ncols : {51, 26, 34, 32}) {
        for (int rs : {64, 52, 32 + ncols, 64 + ncols % 32}) {
            snprintf(nm, sizeof nm, "tile rows of %d lanes, pitch %d", ncols, rs);
            run<0>(nm, tile(ncols, rs), rs);
        }
    }
    run<0>("51 lanes, pitch 64, column step +1 at lane 9", tile(51, 64, 9, 1, 0), 64);
    run<0>("51 lanes, pitch 64, column step -1 at lane 9", tile(51, 64, 9, -1, 0), 64);
    run<0>("51 lanes, pitch 64, row step +1 at lane 9", tile(51, 64, 9, 0, 1), 64);
    run<0>("51 lanes, pitch 83, column step +1 at lane 9", tile(51, 83, 9, 1, 0), 83);
    run<0>("51 lanes, pitch 83, row step +1 at lane 9", tile(51, 83, 9, 0, 1), 83);
    // the window fill: 16 lanes per row, lane i stores the slots 4 i + k (k = 0 .. 3 in four instructions), four rows per wave
    {
        std::vector<int> s(64);
        for (int l = 0; l < 64; ++l) s[l] = (l >> 4) * 64 + (l & 15) * 4;
        run<1>("fill: lane i -> slot 4 i (stride 32 bytes), 4 rows per wave", s, 64 * 4);
        for (int l = 0; l < 64; ++l) s[l] = (l >> 4) * 64 + (l & 15);
        run<1>("fill: lane i -> slot i (consecutive), 4 rows per wave", s, 64 * 4);
        for (int l = 0; l < 64; ++l) s[l] = (l >> 4) * 64 + (l & 15) * 2;
        run<1>("fill: lane i -> slot 2 i (stride 16 bytes), 4 rows per wave", s, 64 * 4);
        for (int l = 0; l < 64; ++l) s[l] = l;
        run<1>("fill: 64 consecutive slots", s, 64);
    }
    return 0;
}
