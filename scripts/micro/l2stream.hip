// micro-benchmark: how fast can a 256-thread block pull L2-resident bytes with the wgrad kernel's loop structure?
// variants: contiguous 1 KiB pieces vs 4 rows x 256 B pieces (row stride 384 B), registers vs LDS-DMA, steps with barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE, int NLOAD, int DEPTH>
__global__ __launch_bounds__(256) void stream(const unsigned char* buf, unsigned bytes, int nsteps, unsigned span, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[65536];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, bytes, 0x00020000);
    unsigned bsel = (MODE & 8) ? blockIdx.x / 22u : blockIdx.x;           // MODE&8: 22 consecutive blocks share a window
    unsigned base = (bsel * 40503u % (span / 32768u)) * 32768u;     // block's window start
    u32x4_t acc = {0, 0, 0, 0};
    for (int st = 0; st < nsteps; ++st) {
        unsigned win = (base + (unsigned)st * 32768u) % span;
#pragma unroll
        for (int j = 0; j < NLOAD; ++j) {
            unsigned off;
            if (MODE & 1) {   // 4 rows x 256 B, row stride 384 B (pixel-major gather)
                unsigned row = (wave * NLOAD + j) * 4 + (lane >> 4);
                off = win + row * 384u + (lane & 15) * 16u;
                if ((MODE & 4) && (lane & 15) >= 12) off = 0x80000000u;      // MODE&4: 12 of 16 chunks valid
            } else {
                off = win + (wave * NLOAD + j) * 1024u + lane * 16u;
            }
            if (MODE & 2) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + (st & 1) * 32768 + (wave * NLOAD + j) * 1024), 16, off, 0, 0, 0);
            } else {
                u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                acc += v;
            }
        }
        if (DEPTH == 1) __syncthreads();
        else { if ((st % DEPTH) == DEPTH - 1) __syncthreads(); }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345u) sink[0] = 1;
}

template <int MODE, int NLOAD, int DEPTH>
void run(const char* name, unsigned char* buf, unsigned bytes, unsigned span, unsigned* sink) {
    int nsteps = 200, blocks = 512 * 2;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((stream<MODE, NLOAD, DEPTH>), dim3(blocks), dim3(256), 0, 0, buf, bytes, nsteps, span, sink);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((stream<MODE, NLOAD, DEPTH>), dim3(blocks), dim3(256), 0, 0, buf, bytes, nsteps, span, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double gb = (double)blocks * nsteps * 4 * NLOAD * 1024.0 / 1e9;
    printf("%-44s span %4u MB: %7.3f ms  %7.1f GB/s  (%5.1f GB/s/CU)\n", name, span >> 20, ms, gb / ms * 1e3, gb / ms * 1e3 / 256);
}

int main() {
    unsigned bytes = 1u << 30;
    unsigned char* buf; unsigned* sink;
    hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes); hipMalloc(&sink, 4);
    for (unsigned span : {2u << 20, 64u << 20}) {
        run<0, 8, 1>("regs contiguous, 8 loads/wave/step", buf, bytes, span, sink);
        run<1, 8, 1>("regs 4x256B rows, 8 loads/wave/step", buf, bytes, span, sink);
        run<2, 8, 1>("lds-dma contiguous, 8 loads/wave/step", buf, bytes, span, sink);
        run<3, 8, 1>("lds-dma 4x256B rows, 8 loads/wave/step", buf, bytes, span, sink);
        run<7, 8, 1>("lds-dma rows, 12/16 valid", buf, bytes, span, sink);
        run<11, 8, 1>("lds-dma rows, 22 blocks share a window", buf, bytes, span, sink);
        run<15, 8, 1>("lds-dma rows, 12/16 valid + shared window", buf, bytes, span, sink);
    }
    return 0;
}
