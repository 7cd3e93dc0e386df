// micro-benchmark: whole-chip HBM read bandwidth of a streaming kernel as a function of independent 16-byte loads in flight per thread
// (U) and of the grid shape.  hipcc --offload-arch=gfx950 -O3 read_bw.hip -o read_bw && ./read_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int U>
__global__ __launch_bounds__(256) void rd(const uint4* __restrict__ x, size_t n16, unsigned* out) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned acc = 0;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const size_t bytes = 1ull << 30;                       // 1 GiB: beyond the 256 MB Infinity Cache
    uint4* x; unsigned* out;
    (void)hipMalloc(&x, bytes); (void)hipMalloc(&out, 64);
    (void)hipMemset(x, 1, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grids[] = {256 * 4, 256 * 8, 256 * 16, 256 * 32};
    for (int gi = 0; gi < 4; ++gi)
        for (int u = 1; u <= 8; u *= 2) {
            float best = 1e9f;
            for (int r = 0; r < 4; ++r) {
                (void)hipEventRecord(e0);
                const size_t n16 = bytes / 16;
                switch (u) {
                    case 1: rd<1><<<grids[gi], 256>>>(x, n16, out); break;
                    case 2: rd<2><<<grids[gi], 256>>>(x, n16, out); break;
                    case 4: rd<4><<<grids[gi], 256>>>(x, n16, out); break;
                    case 8: rd<8><<<grids[gi], 256>>>(x, n16, out); break;
                }
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("grid %5d blocks (%2d per CU)  U = %d loads in flight: %6.1f us  %.2f TB/s\n", grids[gi], grids[gi] / 256, u, best * 1e3, bytes / best / 1e9);
        }
    return 0;
}
