// Development micro-benchmark: HBM read bandwidth of 1 GiB under the access patterns the tile kernels could use.
//   A : every wave owns a contiguous span, lane l reads 2 x 16 B at l*32 (+16)       (what k_ac_tile does)
//   A2: contiguous span, lane l reads 16 B at l*16 and at l*16 + 1024                 (fully coalesced instructions)
//   B : tiles dealt round robin to waves (grid stride), lane layout of A
//   B2: tile groups of 8 KiB dealt round robin to waves
// usage: stream_patterns [log2 bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int PAT, int DEPTH>
__global__ __launch_bounds__(1024) void k_read(const uint4 *__restrict__ p, size_t n_vec, unsigned *out) {
    const unsigned lane = threadIdx.x & 63, wave = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const unsigned n_waves = gridDim.x * (blockDim.x / 64);
    const size_t tiles = n_vec / 128;            // 2 KiB tiles (128 vectors)
    const size_t per_wave = tiles / n_waves;     // (sizes are powers of two)
    unsigned x = 0;
    uint4 buf[DEPTH][2];
    for (size_t t = 0; t < per_wave; t += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            size_t tile;
            if (PAT == 0 || PAT == 1) tile = (size_t)wave * per_wave + t + d;
            else if (PAT == 2) tile = (t + d) * n_waves + wave;
            else tile = (t / DEPTH * n_waves + wave) * DEPTH + d;
            const uint4 *b = p + tile * 128;
            if (PAT == 1) { buf[d][0] = b[lane]; buf[d][1] = b[lane + 64]; }
            else { buf[d][0] = b[lane * 2]; buf[d][1] = b[lane * 2 + 1]; }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) x ^= buf[d][0].x ^ buf[d][0].y ^ buf[d][0].z ^ buf[d][0].w ^ buf[d][1].x ^ buf[d][1].y ^ buf[d][1].z ^ buf[d][1].w;
    }
    if (x == 0x12345678u) out[0] = x;
}

template <int PAT, int DEPTH>
static void run(const char *name, const uint4 *d, size_t n_vec, unsigned *d_out, int grid, int block) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int r = 0; r < 8; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_read<PAT, DEPTH>), dim3(grid), dim3(block), 0, 0, d, n_vec, d_out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); if (r) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("%-28s grid %4d x %4d  median %.4f ms  %.0f GB/s\n", name, grid, block, ms[ms.size() / 2], n_vec * 16.0 / ms[ms.size() / 2] * 1e-6);
}

int main(int argc, char **argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 30;
    const size_t bytes = (size_t)1 << lg, n_vec = bytes / 16;
    uint4 *d; unsigned *d_out;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&d_out, 64)); CK(hipMemset(d, 1, bytes));
    run<0, 4>("A  span, 32B/lane, depth 4", d, n_vec, d_out, 256, 1024);
    run<1, 4>("A2 span, coalesced, depth 4", d, n_vec, d_out, 256, 1024);
    run<2, 4>("B  tile round robin", d, n_vec, d_out, 256, 1024);
    run<3, 4>("B2 group round robin", d, n_vec, d_out, 256, 1024);
    run<0, 8>("A  depth 8", d, n_vec, d_out, 256, 1024);
    run<3, 8>("B2 depth 8", d, n_vec, d_out, 256, 1024);
    run<0, 4>("A  512 blocks x 512", d, n_vec, d_out, 512, 512);
    run<3, 4>("B2 512 blocks x 512", d, n_vec, d_out, 512, 512);
    run<0, 4>("A  1024 blocks x 256", d, n_vec, d_out, 1024, 256);
    run<3, 4>("B2 1024 blocks x 256", d, n_vec, d_out, 1024, 256);
    run<3, 4>("B2 2048 blocks x 256", d, n_vec, d_out, 2048, 256);
    run<1, 4>("A2 1024 blocks x 256", d, n_vec, d_out, 1024, 256);
    return 0;
}
