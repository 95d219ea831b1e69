// LDS gather microbenchmark (gfx950) for the 4-D lean sweep: what a wave's pair / quad reads cost when its lanes are the
// consecutive nodes of a TV1-wide tile (row pitch RS slots) and the gather displacement steps inside the wave.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ldsgather tools/ldsgather.hip && /tmp/ldsgather
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef volatile const __attribute__((address_space(3))) v2f lds_v2;
typedef volatile const __attribute__((address_space(3))) v4f lds_v4;
typedef volatile const __attribute__((address_space(3))) float lds_f;

// SLOT: bytes per slot (8: pairs read with ds_read_b64, 16: quads read with ds_read_b128, 4: single floats ds_read_b32)
template <int SLOT>
__global__ __launch_bounds__(256) void k(float* out, int iters, int TV1, int RS, int step1_at, int step0_at) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int row = lane / TV1, col = lane - row * TV1;
    int slot = row * RS + col + 8;
    if (step1_at >= 0 && lane >= step1_at) slot += 1;        // the column displacement floor steps inside the wave
    if (step0_at >= 0 && lane >= step0_at) slot += RS;       // the row displacement floor steps inside the wave
    const unsigned base = (unsigned)slot * SLOT;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const unsigned a = base + (unsigned)(it & 7) * SLOT;
        if (SLOT == 10) {  // pair layout, (j3, j3 + 1) with ONE ds_read_b128 at an address that is only 8-byte aligned
            v4f q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned ad = (a + (unsigned)u * (unsigned)(RS * SLOT)) / 10 * 8;
                asm volatile("ds_read_b128 %0, %1" : "=v"(q[u]) : "v"(ad));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]));
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += q[u].x * q[u].y + q[u].z * q[u].w;
            continue;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned ad = a + (unsigned)u * (unsigned)(RS * SLOT);
            if (SLOT == 8) {
                const v2f q = *(lds_v2*)(size_t)ad;
                acc += q.x * q.y;
            } else if (SLOT == 9) {  // pair layout, (j3, j3 + 1) with ONE ds_read2_b64 (offset0:0 offset1:1)
                const __attribute__((address_space(3))) v2f* q = (const __attribute__((address_space(3))) v2f*)(size_t)(ad / 9 * 8);
                const v2f q0 = q[0], q1 = q[1];
                acc += q0.x * q0.y + q1.x * q1.y;
            } else if (SLOT == 16) {
                const v4f q = *(lds_v4*)(size_t)ad;
                acc += q.x * q.y + q.z * q.w;
            } else {
                acc += *(lds_f*)(size_t)ad;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int SLOT>
static void run(const char* name, int TV1, int RS, int s1, int s0) {
    float* d;
    hipMalloc(&d, 1024 * 256 * 4);
    const int iters = 2000, blocks = 1024;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<SLOT><<<blocks, 256, 36864>>>(d, 10, TV1, RS, s1, s0);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<SLOT><<<blocks, 256, 36864>>>(d, iters, TV1, RS, s1, s0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr_per_cu = (double)blocks / 256 * 4 * iters * 8;
    printf("%-10s TV1 %2d RS %2d step1@%2d step0@%2d : %.2f clk per wave-instruction per CU\n", name, TV1, RS, s1, s0,
           ms * 1e-3 * 2.4e9 / instr_per_cu);
    hipFree(d);
}

int main() {
    for (int tv1 : {64, 34, 17}) {
        const int rs_cong = tv1 == 64 ? 64 : (tv1 == 34 ? 66 : 49), rs_64 = 64;
        for (int rs : {rs_cong, rs_64}) {
            run<8>("b64 pair", tv1, rs, -1, -1);
            run<8>("b64 pair", tv1, rs, 9, -1);
            run<8>("b64 pair", tv1, rs, -1, 9);
            run<8>("b64 pair", tv1, rs, 9, 41);
            run<8>("b64 pair", tv1, rs, 20, 20);
        }
    }
    for (int tv1 : {64, 34, 17}) {
        const int rs = tv1 == 64 ? 64 : (tv1 == 34 ? 34 + 16 : 17 + 16);  // quads: 16 bank groups of 16 bytes
        run<16>("b128 quad", tv1, rs, -1, -1);
        run<16>("b128 quad", tv1, rs, 9, -1);
        run<16>("b128 quad", tv1, rs, -1, 9);
        run<16>("b128 quad", tv1, rs, 9, 41);
        run<16>("b128 quad", tv1, rs, 20, 20);
    }
    for (int tv1 : {64, 34, 17}) {
        const int rs = tv1 == 64 ? 64 : (tv1 == 34 ? 66 : 49);
        run<9>("read2_b64", tv1, rs, -1, -1);
        run<9>("read2_b64", tv1, rs, 9, 41);
    }
    for (int tv1 : {64, 34, 17}) {
        const int rs = tv1 == 64 ? 64 : (tv1 == 34 ? 66 : 49);
        run<10>("b128@8", tv1, rs, -1, -1);
        run<10>("b128@8", tv1, rs, 9, 41);
        run<10>("b128@8", tv1, rs, 20, 20);
    }
    run<4>("b32", 64, 64, -1, -1);
    run<4>("b32", 34, 66, 9, 41);
    return 0;
}
