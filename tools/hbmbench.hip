// HBM read-stream calibration: what a pure 16-byte-per-lane read of N bytes achieves on this box (the ceiling the packed
// table-tier sweep is measured against).   hipcc --offload-arch=gfx950 -O3 tools/hbmbench.hip -o /tmp/hb && /tmp/hb
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ in, float* __restrict__ out, long long n, int per) {
    long long i = (long long)blockIdx.x * blockDim.x * per + threadIdx.x;
    float acc = 0.f;
    for (int k = 0; k < per; ++k, i += blockDim.x)
        if (i < n) {
            const float4 v = in[i];
            acc += v.x + v.y + v.z + v.w;
        }
    if (acc == 12345.678f) out[0] = acc;  // never true: keeps the loads alive
}
int main() {
    const long long bytes = 818ll << 20, n = bytes / 16;
    float4* d; float* o;
    (void)hipMalloc(&d, bytes); (void)hipMalloc(&o, 4); (void)hipMemset(d, 0, bytes);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int per : {1, 4, 8, 16}) {
        const unsigned grid = (unsigned)((n + 256ll * per - 1) / (256ll * per));
        k_read<<<grid, 256>>>(d, o, n, per);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        for (int r = 0; r < 10; ++r) k_read<<<grid, 256>>>(d, o, n, per);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("read %lld MB, %2d x 16 B per thread: %.3f ms -> %.0f GB/s\n", bytes >> 20, per, ms / 10, bytes / (ms / 10 * 1e-3) / 1e9);
    }
    return 0;
}
