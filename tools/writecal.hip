// writecal.hip -- what does WRITE_SIZE say for the stores of the 4-D window sweep?  (development tool; VERDICT r4 next #4a)
//
// The counter passes of C4 (cart-pole 151^4) report 5.98 GB written per sweep for 2.60 GB of J and pi (profiles/r04_counters_c4.json):
// tiles of 19 x 26 nodes store 104-byte row pieces of J and 26-byte row pieces of pi at a row pitch of 604 bytes, so nearly every piece
// ends inside a 64-byte sector that the column-neighbour tile completes.  The guide calibrates the counter on a 16 B / lane stream only.
// This program writes a known number of bytes in several patterns; `rocprofv3 --pmc WRITE_SIZE` over it gives the counter per kernel:
//
//   k_stream16     16 bytes per lane, contiguous                                   (the guide's calibration stream)
//   k_stream4      4 bytes per lane, contiguous                                    (a float store per lane, whole rows)
//   k_stream1      1 byte per lane, contiguous                                     (a pi store per lane, whole rows)
//   k_tiles<0>     J (4 B) and pi (1 B) of R x C tiles of a V x V plane, tiles in plane order, block b = tile b
//   k_tiles<1>     ... the tiles of a plane dealt to the XCDs as lean4_schedule does (block b -> XCD b % 8 takes the next tile of ITS
//                  eighth of the (plane, tile) list): column neighbours run back to back on one XCD
//   k_tiles<2>     ... block b -> tile b, but the column chunks of one row piece on DIFFERENT XCDs (b % 8 = chunk): the worst case
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/writecal tools/writecal.hip
//   cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/wc -o p -- /tmp/writecal 151 19 26 400
//   python tools/writecal_summary.py /tmp/wc        (bytes each kernel wrote / counter value -> bytes per count, amplification)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                       \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

__global__ void k_stream16(float4* __restrict__ out, long long n16, float v) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) out[i] = make_float4(v, v, v, v);
}
__global__ void k_stream4(float* __restrict__ out, long long n, float v) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v;
}
__global__ void k_stream1(unsigned char* __restrict__ out, long long n, int v) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (unsigned char)v;
}

struct Tile {
    int plane, row0, nrows, col0, ncols;
};
// one workgroup per tile, thread s -> (row s / ncols, column s % ncols): the lane map of k_sweep_lean4
template <int ORDER>   // (the order only names the kernel: one row per order in the counter output)
__global__ __launch_bounds__(512) void k_tiles(const Tile* __restrict__ tiles, float* __restrict__ J, unsigned char* __restrict__ pi, int V, float v) {
    const Tile t = tiles[blockIdx.x];
    if (t.nrows <= 0) return;
    const int r = (int)threadIdx.x / t.ncols, c = (int)threadIdx.x - r * t.ncols;
    if (r >= t.nrows) return;
    const long long o = ((long long)t.plane * V + t.row0 + r) * V + t.col0 + c;
    J[o] = v + (float)threadIdx.x;
    pi[o] = (unsigned char)(threadIdx.x & 31);
}

int main(int argc, char** argv) {
    const int V = argc > 1 ? atoi(argv[1]) : 151, R = argc > 2 ? atoi(argv[2]) : 19, C = argc > 3 ? atoi(argv[3]) : 26;
    const int planes = argc > 4 ? atoi(argv[4]) : 400, reps = argc > 5 ? atoi(argv[5]) : 5;
    if (R * C > 512 || R < 1 || C < 1) {
        fprintf(stderr, "R x C <= 512\n");
        return 1;
    }
    const long long n = (long long)planes * V * V;
    float* J;
    unsigned char* pi;
    CHK(hipMalloc(&J, ((n + 3) / 4) * 16));
    CHK(hipMalloc(&pi, n + 16));
    // the tiles of a plane: row pieces of at most R rows, each cut into near-equal column chunks of at most C columns (lean4_row_tiles)
    std::vector<Tile> plane_tiles;
    const int nr = (V + R - 1) / R, nc = (V + C - 1) / C;
    for (int a = 0; a < nr; ++a)
        for (int b = 0; b < nc; ++b) {
            const int r0 = (int)((long long)V * a / nr), r1 = (int)((long long)V * (a + 1) / nr);
            const int c0 = (int)((long long)V * b / nc), c1 = (int)((long long)V * (b + 1) / nc);
            plane_tiles.push_back(Tile{0, r0, r1 - r0, c0, c1 - c0});
        }
    const int tpp = (int)plane_tiles.size();
    std::vector<Tile> order[3];
    for (int p = 0; p < planes; ++p)
        for (const Tile& t : plane_tiles) order[0].push_back(Tile{p, t.row0, t.nrows, t.col0, t.ncols});
    {   // XCD x takes the x-th contiguous eighth of the (plane, tile) list; its entries go to blocks 8 j + x
        const long long E = (long long)order[0].size();
        size_t mx = 0;
        std::vector<std::vector<Tile>> lists(8);
        for (int x = 0; x < 8; ++x) {
            for (long long e = E * x / 8; e < E * (x + 1) / 8; ++e) lists[x].push_back(order[0][(size_t)e]);
            mx = std::max(mx, lists[x].size());
        }
        order[1].assign(8 * mx, Tile{0, 0, 0, 0, 0});
        for (int x = 0; x < 8; ++x)
            for (size_t j = 0; j < lists[x].size(); ++j) order[1][8 * j + x] = lists[x][j];
    }
    {   // column chunk b of every row piece on XCD b % 8: neighbours never share an L2
        std::vector<std::vector<Tile>> lists(8);
        for (int p = 0; p < planes; ++p)
            for (int a = 0; a < nr; ++a)
                for (int b = 0; b < nc; ++b) {
                    const Tile& t = plane_tiles[(size_t)a * nc + b];
                    lists[b % 8].push_back(Tile{p, t.row0, t.nrows, t.col0, t.ncols});
                }
        size_t mx = 0;
        for (auto& l : lists) mx = std::max(mx, l.size());
        order[2].assign(8 * mx, Tile{0, 0, 0, 0, 0});
        for (int x = 0; x < 8; ++x)
            for (size_t j = 0; j < lists[x].size(); ++j) order[2][8 * j + x] = lists[x][j];
    }
    Tile* dt[3];
    for (int k = 0; k < 3; ++k) {
        CHK(hipMalloc(&dt[k], order[k].size() * sizeof(Tile)));
        CHK(hipMemcpy(dt[k], order[k].data(), order[k].size() * sizeof(Tile), hipMemcpyHostToDevice));
    }
    printf("V %d tile %dx%d planes %d: %d tiles per plane, nodes %lld, J bytes %lld, pi bytes %lld\n", V, R, C, planes, tpp, n, 4 * n, n);
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    auto timed = [&](const char* name, long long bytes, auto&& launch) {
        launch();
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CHK(hipEventRecord(e1));
        CHK(hipDeviceSynchronize());
        float ms = 0.f;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("WROTE %-12s %lld bytes per launch, %.3f ms, %.0f GB/s\n", name, bytes, ms / reps, (double)bytes / (ms / reps) * 1e-6);
    };
    const long long n16 = (n + 3) / 4;
    timed("k_stream16", n16 * 16, [&] { hipLaunchKernelGGL(k_stream16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (float4*)J, n16, 1.f); });
    timed("k_stream4", n * 4, [&] { hipLaunchKernelGGL(k_stream4, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, J, n, 2.f); });
    timed("k_stream1", n, [&] { hipLaunchKernelGGL(k_stream1, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, pi, n, 3); });
    const char* names[3] = {"k_tiles/plane", "k_tiles/xcd", "k_tiles/split"};
    timed(names[0], n * 5, [&] { hipLaunchKernelGGL(k_tiles<0>, dim3((unsigned)order[0].size()), dim3(512), 0, 0, dt[0], J, pi, V, 0.f); });
    timed(names[1], n * 5, [&] { hipLaunchKernelGGL(k_tiles<1>, dim3((unsigned)order[1].size()), dim3(512), 0, 0, dt[1], J, pi, V, 1.f); });
    timed(names[2], n * 5, [&] { hipLaunchKernelGGL(k_tiles<2>, dim3((unsigned)order[2].size()), dim3(512), 0, 0, dt[2], J, pi, V, 2.f); });
    CHK(hipGetLastError());
    return 0;
}
