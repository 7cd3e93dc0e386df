// micro-benchmark: how fast can a grid stream a 364 MB tensor of [pixel][32 bf16 channels] rows to HBM with the store shapes the
// kernels' epilogues use?  (hipcc --offload-arch=gfx950 -O3 store_patterns.hip -o store_patterns && ./store_patterns)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

// every wave writes "items" of 32 pixels x 64 bytes = 2 KB; pattern P decides which lane writes which bytes
template <int P>
__global__ __launch_bounds__(256) void k(uint8_t* y, unsigned nitems_total, unsigned seed) {
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned j = lane & 31, h = lane >> 5;
    const unsigned per_block = 40;                      // items per block, 10 per wave
    for (unsigned it = wv; it < per_block; it += 4) {
        const unsigned item = blockIdx.x * per_block + it;
        if (item >= nitems_total) break;
        uint8_t* base = y + (size_t)item * 2048;
        const unsigned v = seed + lane + it;
        if (P == 0) {            // 16 B per lane, 1 KB contiguous per instruction
#pragma unroll
            for (int i = 0; i < 2; ++i) *reinterpret_cast<uint4*>(base + i * 1024 + lane * 16) = make_uint4(v, v + 1, v + 2, v + 3);
        } else if (P == 1) {     // 8 B per lane: pixel j, channel run 4 * h + 8 * gq  (32x32 MFMA output, rows = channels)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) *reinterpret_cast<uint2*>(base + j * 64 + gq * 16 + h * 8) = make_uint2(v, v + gq);
        } else if (P == 2) {     // 16 B per lane: pixel j, bytes 32 * i + 16 * h  (after one permlane32 swap)
#pragma unroll
            for (int i = 0; i < 2; ++i) *reinterpret_cast<uint4*>(base + j * 64 + i * 32 + h * 16) = make_uint4(v, v + 1, v + 2, v + i);
        } else if (P == 3) {     // 2 B per lane: 32 lanes = one pixel's 64 B  (32x32 MFMA output, rows = pixels)
#pragma unroll
            for (int r = 0; r < 16; ++r) *reinterpret_cast<unsigned short*>(base + (8 * (r >> 2) + 4 * h + (r & 3)) * 64 + j * 2) = (unsigned short)(v + r);
        } else if (P == 4) {     // 4 B per lane: 16 lanes = one pixel's 64 B (16x16-style: lane & 15 = channel pair)
#pragma unroll
            for (int r = 0; r < 8; ++r) *reinterpret_cast<unsigned*>(base + (4 * r + (lane >> 4)) * 64 + (lane & 15) * 4) = v + r;
        } else if (P == 5) {     // 8 B per lane, 8 lanes = one pixel's 64 B, 512 B contiguous per instruction
#pragma unroll
            for (int r = 0; r < 4; ++r) *reinterpret_cast<uint2*>(base + r * 512 + lane * 8) = make_uint2(v, v + r);
        } else if (P == 6) {     // 16x16 MFMA output, rows = pixels (lane & 15), 4 channels per lane: 8 B/lane, 32-byte runs at 128 B stride
#pragma unroll
            for (int r = 0; r < 4; ++r) *reinterpret_cast<uint2*>(base + (lane & 15) * 128 + r * 32 + (lane >> 4) * 8) = make_uint2(v, v + r);
        } else if (P == 7) {     // the same after exchanging halves between lane pairs: 16 B/lane, 64-byte runs at 128 B stride
#pragma unroll
            for (int r = 0; r < 2; ++r) *reinterpret_cast<uint4*>(base + (lane & 15) * 128 + r * 64 + (lane >> 4) * 16) = make_uint4(v, v + 1, v + 2, v + r);
        }
    }
}

int main() {
    const size_t bytes = 256ull * 149 * 149 * 64;
    const unsigned nitems = (unsigned)(bytes / 2048);
    uint8_t* y;
    hipMalloc(&y, bytes + 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned grid = (nitems + 39) / 40;
    const char* names[] = {"16 B/lane, 1 KB contiguous", "8 B/lane, 16 B runs at 64 B stride", "16 B/lane, 32 B runs at 64 B stride",
                           "2 B/lane, 64 B rows", "4 B/lane, 64 B rows", "8 B/lane, 512 B contiguous",
                           "8 B/lane, 32 B runs at 128 B stride", "16 B/lane, 64 B runs at 128 B stride"};
    for (int rep = 0; rep < 2; ++rep)
    for (int p = 0; p < 8; ++p) {
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0);
            switch (p) {
                case 0: k<0><<<grid, 256>>>(y, nitems, r); break;
                case 1: k<1><<<grid, 256>>>(y, nitems, r); break;
                case 2: k<2><<<grid, 256>>>(y, nitems, r); break;
                case 3: k<3><<<grid, 256>>>(y, nitems, r); break;
                case 4: k<4><<<grid, 256>>>(y, nitems, r); break;
                case 5: k<5><<<grid, 256>>>(y, nitems, r); break;
                case 6: k<6><<<grid, 256>>>(y, nitems, r); break;
                case 7: k<7><<<grid, 256>>>(y, nitems, r); break;
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        if (rep) printf("P%d %-40s %7.1f us  %.2f TB/s\n", p, names[p], best * 1e3, bytes / best / 1e9);
    }
    return 0;
}
