// Conv2d_1a_3x3 (3x3 / stride 2 / no padding, 32 output channels) straight from the RESIZED u8 ROI plane.
//
// The reference feeds the stem with  PIL 'L' -> convert('RGB') -> Resize -> ToTensor -> Normalize  (neuston_data.py:342-371,
// 456-464): three copies of one grey plane g, each under its own affine x_c = a_c * g + b_c (ToTensor's /255, Normalize,
// [TV] transform_input folded into a_c, b_c).  The conv of those three planes is a ONE-plane conv plus a constant:
//     y[k] = sum_{r,s} g[r,s] * (sum_c a_c w[k,r,s,c])  +  sum_{r,s,c} b_c w[k,r,s,c]            (no padding: the constant is exact)
// and its weight gradient needs only  A[k,r,s] = sum dy[k] g[r,s]  and  B[k] = sum dy[k]:
//     dw[k,r,s,c] = a_c A[k,r,s] + b_c B[k].
// So the [B,S,S,8] input tensor (366 MB per batch of 256 at 299 px, written by the resize and read back 2.25 times by the GEMM
// kernels) never exists: the resize writes the u8 plane (23 MB), these kernels read it through L1/L2.  The floor of both is the one
// tensor they must move (the raw output / its gradient, 364 MB: 62 us as a plain fill); measured per batch of 256: forward 124 us,
// weight gradient 125 us on the matrix cores (below), 152 / 213 us as vector FMAs -- against 317 / 180 us for the GEMM kernels.
// Arithmetic: fp32 on the fp32 MASTER weights and exact u8 pixels (the GEMM path rounds x_c and w to bf16 first), outputs rounded
// to the storage type; BatchNorm partial sums over the rounded outputs like every other conv epilogue.
// Replaces aten::conv2d fwd / weight-grad of [TV] Inception3.Conv2d_1a_3x3 (reference call site neuston_models.py:66-68, 81-86).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int K1 = 32;         // output channels
constexpr int PIXB = 2048;     // output pixels per block: 64 per trip (4 threads x 8 channels each), one partial row per block

struct StemArgs {
    const uint8_t* g;
    const float* w;            // fp32 master [32][3][3][3]
    const float* ab;           // a_0..2, b_0..2 (device)
    void* y;
    const void* dy;
    float* part;
    const float* scale;
    const float* shift;
    int relu, H, W, P, Q, ld;
    unsigned M, PQ;
    fastdiv_t dPQ, dQ;
    unsigned rows, gpr;        // MFMA kernels: output rows N*P, 32-pixel groups per output row
    fastdiv_t dP, dG;
#ifdef IFCBK_EXPERIMENT_STEM
    int dbg;                   // timing-only builds: 1 no pixel loads, 2 no stores, 4 no statistics
#endif
};
#ifdef IFCBK_EXPERIMENT_STEM
#define STEM_DBG(a, bit) ((a).dbg & (bit))
#else
#define STEM_DBG(a, bit) 0
#endif

template <class T> __device__ __forceinline__ void store8(T* p, const float* f);
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float* f) { *reinterpret_cast<uint4*>(p) = pack8(f); }
template <> __device__ __forceinline__ void store8<float>(float* p, const float* f) {
    *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
}
template <class T> __device__ __forceinline__ void load8(const T* p, float* f);
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float* f) { unpack8(*reinterpret_cast<const uint4*>(p), f); }
template <> __device__ __forceinline__ void load8<float>(const float* p, float* f) {
    *reinterpret_cast<float4*>(f) = *reinterpret_cast<const float4*>(p);
    *reinterpret_cast<float4*>(f + 4) = *reinterpret_cast<const float4*>(p + 4);
}

// the nine taps of output pixel m (bytes of three source rows, two apart), as loaded
__device__ __forceinline__ void taps_raw(const StemArgs& a, unsigned m, uint8_t* gb) {
    const unsigned n = fdiv(m, a.dPQ), rem = m - n * a.PQ;
    const unsigned p = fdiv(rem, a.dQ), q = rem - p * (unsigned)a.Q;
    const uint8_t* s = a.g + ((size_t)n * a.H + 2 * p) * a.W + 2 * q;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) gb[r * 3 + c] = s[(size_t)r * a.W + c];
}

constexpr int UNR = 4;         // trips whose loads are issued together: one memory round trip per UNR trips instead of per trip

// AFFINE: eval form, y = act(conv * scale + shift) (folded BatchNorm), no statistics
template <class T, bool AFFINE>
__global__ __launch_bounds__(256) void stem_u8_fwd_kernel(StemArgs a) {
    const int c8 = threadIdx.x & 3, pl = threadIdx.x >> 2;
    float we[8][9], bias[8];
    {
        float av[3], bv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { av[c] = a.ab[c]; bv[c] = a.ab[3 + c]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float* wk = a.w + (size_t)(c8 * 8 + j) * 27;
            float b = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float w0 = wk[t * 3], w1 = wk[t * 3 + 1], w2 = wk[t * 3 + 2];
                we[j][t] = w0 * av[0] + w1 * av[1] + w2 * av[2];
                b += w0 * bv[0] + w1 * bv[1] + w2 * bv[2];
            }
            bias[j] = b;
        }
    }
    float sc[8], sh[8];
    if (AFFINE) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc[j] = a.scale[c8 * 8 + j]; sh[j] = a.shift[c8 * 8 + j]; }
    }
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
    const unsigned m0 = blockIdx.x * (unsigned)PIXB + pl;
    T* y = (T*)a.y;
    for (int it = 0; it < PIXB / 64; it += UNR) {
        if (m0 + it * 64 >= a.M) break;
        // (the loads of UNR trips go out together; out-of-range trips load pixel M-1 and skip the store.  This vector form serves fp32
        // storage and rows narrower than 32 pixels; it is VALU-bound: 125 instructions per 16 pixels and wave)
        uint8_t gb[UNR][9];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const unsigned m = m0 + (it + u) * 64;
            taps_raw(a, m < a.M ? m : a.M - 1, gb[u]);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const unsigned m = m0 + (it + u) * 64;
            if (m >= a.M) break;
            float gt[9], o[8];
#pragma unroll
            for (int t = 0; t < 9; ++t) gt[t] = (float)gb[u][t];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = bias[j];
#pragma unroll
                for (int t = 0; t < 9; ++t) v = fmaf(gt[t], we[j][t], v);
                if (AFFINE) {
                    v = fmaf(v, sc[j], sh[j]);
                    if (a.relu) v = v > 0.f ? v : 0.f;
                } else {
                    v = Chunk<T>::round(v);
                    s1[j] += v;
                    s2[j] = fmaf(v, v, s2[j]);
                }
                o[j] = v;
            }
            store8<T>(y + (size_t)m * a.ld + c8 * 8, o);
        }
    }
    if (AFFINE || !a.part) return;
    // fixed-order block sums: lanes of equal channel group (lane & 3) by xor shuffles, then the four waves through LDS
    __shared__ float red[4][2][K1];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int d = 4; d < 64; d <<= 1) {
            s1[j] += __shfl_xor(s1[j], d);
            s2[j] += __shfl_xor(s2[j], d);
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane < 4) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[wv][0][lane * 8 + j] = s1[j]; red[wv][1][lane * 8 + j] = s2[j]; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * K1) {
        const int st = threadIdx.x >> 5, ch = threadIdx.x & 31;
        a.part[((size_t)blockIdx.x * 2 + st) * K1 + ch] = (red[0][st][ch] + red[1][st][ch]) + (red[2][st][ch] + red[3][st][ch]);
    }
}

// per-block partial sums part[block][k][0..8] = sum dy[k] * g[tap], [9] = sum dy[k]
template <class T>
__global__ __launch_bounds__(256) void stem_u8_wgrad_kernel(StemArgs a) {
    const int c8 = threadIdx.x & 3, pl = threadIdx.x >> 2;
    float acc[8][10];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int t = 0; t < 10; ++t) acc[j][t] = 0.f;
    const unsigned m0 = blockIdx.x * (unsigned)PIXB + pl;
    const T* dy = (const T*)a.dy;
    for (int it = 0; it < PIXB / 64; it += UNR) {
        if (m0 + it * 64 >= a.M) break;
        uint8_t gb[UNR][9];
        float d[UNR][8];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const unsigned m = m0 + (it + u) * 64, mc = m < a.M ? m : a.M - 1;
            load8<T>(dy + (size_t)mc * a.ld + c8 * 8, d[u]);
            taps_raw(a, mc, gb[u]);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (m0 + (it + u) * 64 >= a.M) break;
            float gt[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) gt[t] = (float)gb[u][t];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[j][t] = fmaf(d[u][j], gt[t], acc[j][t]);
                acc[j][9] += d[u][j];
            }
        }
    }
    __shared__ float red[4][K1 * 10];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            float v = acc[j][t];
#pragma unroll
            for (int d = 4; d < 64; d <<= 1) v += __shfl_xor(v, d);
            if (lane < 4) red[wv][(lane * 8 + j) * 10 + t] = v;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < K1 * 10; i += 256)
        a.part[(size_t)blockIdx.x * (K1 * 10) + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// ---------------------------------------------------------------- bf16 storage: the same two passes on the matrix cores
// As vector FMAs the forward was VALU-bound (125 instructions per 16 pixels and wave: 152 us for a tensor a fill writes in 62 us), the
// weight gradient likewise (213 us).  Both are small GEMMs whose bf16 operands are EXACT: pixels 0..255 and dy are bf16 values, and the
// fp32 effective taps go in as three bf16 pieces (hi + mid + lo = the fp32 value to 2^-24), each piece against its own copy of the pixel:
//   forward   D[pixel][ch] = sum_k G[pixel][k] * Wsplit[k][ch],  k = 27 (tap, piece) slots + 3 bias pieces against a constant 1 + 2 zeros
//             = two v_mfma_f32_32x32x16_bf16 per 32 pixels; a lane ends up with ONE channel of 16 pixels -> 32 lanes store a pixel's 64 bytes
//   wgrad     D[ch][tap] = sum_pixels dy[pixel][ch] * G[pixel][tap], taps 0..8 + a constant-1 column (the sum of dy) = two
//             v_mfma_f32_16x16x32_bf16 per 32 pixels (channels 0..15 / 16..31)
// Work unit of a wave: 32 consecutive pixels of one output row (the last group of a row is partly masked); a block owns RB output rows.
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
constexpr int RB = 8;

__device__ __forceinline__ unsigned bfbits(float f) { return __float_as_uint(f) >> 16; }           // exact for the values used here
__device__ __forceinline__ unsigned bfpair(unsigned lo, unsigned hi) { return lo | (hi << 16); }
// w = hi + mid + lo with bf16 pieces (round-to-nearest each; the remainders are exact in fp32)
__device__ __forceinline__ void split3(float w, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = f2bf(w);
    const float r1 = w - bf2f((bf16_t)hi);
    mid = f2bf(r1);
    lo = f2bf(r1 - bf2f((bf16_t)mid));
}
__device__ __forceinline__ bf16x8_t frag(unsigned a, unsigned b, unsigned c, unsigned d) {
    u32x4_t v = {a, b, c, d};
    return __builtin_bit_cast(bf16x8_t, v);
}

template <bool AFFINE>
__global__ __launch_bounds__(256) void stem_u8_fwd_mfma_kernel(StemArgs a) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    // D[pixel][channel] = G[pixel][k] * Wsplit[k][channel]: a lane owns CHANNEL j of 16 pixels (accumulator v <-> pixel
    // 8 * (v / 4) + 4 * h + v % 4 of the group), so one store instruction writes the 64 contiguous bytes of a pixel from 32 lanes.
    // (With channels as rows a lane held 4-channel runs of one pixel: 8-byte stores in 16-byte runs at 64-byte stride, which stream at
    // 3.3 TB/s where every other shape reaches 5.3 -- scripts/micro/store_patterns.hip.)
    bf16x8_t W1, W2;                                  // k slots of channel j: see above
    {
        const float* wk = a.w + (size_t)j * 27;
        unsigned hi[9], mid[9], lo[9], bh, bm, bl;
        float b = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float w0 = wk[t * 3], w1 = wk[t * 3 + 1], w2 = wk[t * 3 + 2];
            split3(w0 * a.ab[0] + w1 * a.ab[1] + w2 * a.ab[2], hi[t], mid[t], lo[t]);
            b += w0 * a.ab[3] + w1 * a.ab[4] + w2 * a.ab[5];
        }
        split3(b, bh, bm, bl);
        if (h == 0) {
            W1 = frag(bfpair(hi[0], hi[1]), bfpair(hi[2], hi[3]), bfpair(hi[4], hi[5]), bfpair(hi[6], hi[7]));
            W2 = frag(bfpair(mid[0], mid[1]), bfpair(mid[2], mid[3]), bfpair(mid[4], mid[5]), bfpair(mid[6], mid[7]));
        } else {
            W1 = frag(bfpair(hi[8], mid[8]), bfpair(lo[8], bh), bfpair(bm, bl), 0u);
            W2 = frag(bfpair(lo[0], lo[1]), bfpair(lo[2], lo[3]), bfpair(lo[4], lo[5]), bfpair(lo[6], lo[7]));
        }
    }
    float sc = 1.f, sh = 0.f;
    if (AFFINE) { sc = a.scale[j]; sh = a.shift[j]; }
    float s1 = 0.f, s2 = 0.f;
    // byte offset of accumulator v's element inside a full group (pixels all inside the row)
    unsigned voff[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) voff[v] = ((unsigned)(8 * (v >> 2) + 4 * h + (v & 3)) * (unsigned)a.ld + (unsigned)j) * 2u;
    const unsigned row0 = blockIdx.x * (unsigned)RB, nitem = (unsigned)RB * a.gpr;
    const unsigned ONE = 0x3f80u;
    // The LAST group of a row starts at pixel Q - 32: it overlaps its neighbour instead of hanging over the row end, so every lane
    // always has a real pixel, the stores need no clamping (the overlap is written twice with the same values) and only the
    // statistics have to skip the `ovl` pixels that the neighbour already counted.  (Q >= 32; narrower outputs use the vector kernel.)
    const unsigned ovl = a.gpr * 32 - (unsigned)a.Q;
    // The nine bytes of the NEXT work unit are requested before this unit's stores are issued: memory operations retire in order,
    // so a wait for loads issued behind the stores would also wait for the stores (measured: loads 40 us + stores 40 us + arithmetic
    // 63 us added up to the kernel's 152 us).  The loop body has no memory operation under a branch for the same reason: at a join
    // the compiler assumes the path that issued fewest and waits for everything.
    // raw loads only (three unaligned 16-bit + three 8-bit loads): the bytes are taken apart at the top of the NEXT trip, behind a
    // scheduling barrier -- left to itself the scheduler pulls those cheap ALU ops up between the stores, and their wait for the loads
    // (issued after the previous trip's stores) then waits for those stores as well
    auto fetch = [&](unsigned item, unsigned& row, unsigned& grp, unsigned short* r01, uint8_t* r2) {
        const unsigned rr = fdiv(item, a.dG);      // wave-uniform
        grp = item - rr * a.gpr;
        row = row0 + rr;
        const unsigned rowc = row < a.rows ? row : a.rows - 1;
        const unsigned n = fdiv(rowc, a.dP), p = rowc - n * (unsigned)a.P;
        const unsigned q = grp * 32 - (grp == a.gpr - 1 ? ovl : 0u) + j;
        const uint8_t* s = a.g + ((size_t)n * a.H + 2 * p) * a.W + 2 * q;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            __builtin_memcpy(&r01[r], s + (size_t)r * a.W, 2);
            r2[r] = s[(size_t)r * a.W + 2];
        }
    };
    unsigned row, grp, row_n, grp_n;
    unsigned short r01[3], r01_n[3];
    uint8_t r2[3], r2_n[3];
    fetch(wv, row_n, grp_n, r01_n, r2_n);
    // (nothing pending at loop entry: otherwise the compiler's single wait at the loop head must cover this edge too -- vmcnt(0) -- and on
    // the back edge that waits for the stores after all)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (unsigned item = wv; item < nitem; item += 4) {
        row = row_n; grp = grp_n;
#pragma unroll
        for (int r = 0; r < 3; ++r) { r01[r] = r01_n[r]; r2[r] = r2_n[r]; }
        if (row >= a.rows) break;
        __builtin_amdgcn_sched_barrier(0);
        unsigned t[9];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            t[3 * r] = bfbits((float)(r01[r] & 0xffu));
            t[3 * r + 1] = bfbits((float)(r01[r] >> 8));
            t[3 * r + 2] = bfbits((float)r2[r]);
        }
        if (STEM_DBG(a, 1)) {
#pragma unroll
            for (int i = 0; i < 9; ++i) t[i] = bfbits((float)((lane + i) & 255));
        }
        fetch(item + 4 < nitem ? item + 4 : item, row_n, grp_n, r01_n, r2_n);
        const bf16x8_t G2 = frag(bfpair(t[0], t[1]), bfpair(t[2], t[3]), bfpair(t[4], t[5]), bfpair(t[6], t[7]));
        const bf16x8_t G1 = h ? frag(bfpair(t[8], t[8]), bfpair(t[8], ONE), bfpair(ONE, ONE), 0u) : G2;
        f32x16_t acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G1, W1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G2, W2, acc, 0, 0, 0);
        const unsigned skip = grp == a.gpr - 1 ? ovl : 0u;        // pixels of this group that the previous group already covered
        // group base (wave-uniform -> scalar registers): pixel row * Q + first pixel of the group, channel 0
        const size_t gbase = ((size_t)row * a.Q + grp * 32 - skip) * a.ld * 2;
        // (readfirstlane returns int: without the unsigned casts the low half sign-extends once the offset passes 2 GiB -- 1,511
        //  images of 149 x 149 x 32 -- and the rows of the later images land 4 GiB in front of the tensor)
        uint8_t* yb = (uint8_t*)a.y + (((size_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(gbase >> 32)) << 32) |
                                       (size_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)gbase));
        unsigned short ob[16];
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            float x = acc[v];
            if (AFFINE) {
                x = fmaf(x, sc, sh);
                if (a.relu) x = x > 0.f ? x : 0.f;
            }
            ob[v] = f2bf(x);
        }
        if (!AFFINE && !STEM_DBG(a, 4)) {
            if (skip == 0) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float x = bf2f(ob[v]);
                    s1 += x;
                    s2 = fmaf(x, x, s2);
                }
            } else {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float x = (unsigned)(8 * (v >> 2) + 4 * h + (v & 3)) >= skip ? bf2f(ob[v]) : 0.f;
                    s1 += x;
                    s2 = fmaf(x, x, s2);
                }
            }
        }
#pragma unroll
        for (int v = 0; v < 16; ++v)
            if (!STEM_DBG(a, 2) || ob[v] == 0x1234) *reinterpret_cast<bf16_t*>(yb + voff[v]) = ob[v];
        __builtin_amdgcn_sched_barrier(0);
    }
    if (AFFINE || !a.part) return;
    __shared__ float red[4][2][K1];
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    if (h == 0) { red[wv][0][j] = s1; red[wv][1][j] = s2; }
    __syncthreads();
    if (threadIdx.x < 2 * K1) {
        const int st = threadIdx.x >> 5, ch = threadIdx.x & 31;
        a.part[((size_t)blockIdx.x * 2 + st) * K1 + ch] = (red[0][st][ch] + red[1][st][ch]) + (red[2][st][ch] + red[3][st][ch]);
    }
}

__global__ __launch_bounds__(256) void stem_u8_wgrad_mfma_kernel(StemArgs a) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, kg = lane >> 4;         // A: row = channel c (c + 16 in the second product); B: column = tap c; k = 8 pixels from 8 * kg
    const int tr = c < 9 ? c / 3 : 0, tc = c < 9 ? c % 3 : 0;
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const unsigned row0 = blockIdx.x * (unsigned)RB, nitem = (unsigned)RB * a.gpr;
    const bf16_t* dy = (const bf16_t*)a.dy;
    for (unsigned item = wv; item < nitem; item += 4) {
        const unsigned rr = fdiv(item, a.dG), grp = item - rr * a.gpr, row = row0 + rr;      // wave-uniform
        if (row >= a.rows) break;
        const unsigned n = fdiv(row, a.dP), p = row - n * (unsigned)a.P;
        const unsigned q0 = grp * 32 + 8 * kg;
        const bf16_t* dp = dy + (size_t)row * a.Q * a.ld + c;
        const uint8_t* s = a.g + ((size_t)n * a.H + 2 * p + tr) * a.W + tc;
        unsigned d0[8], d1[8], gb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned q = q0 + e;
            const bool ok = q < (unsigned)a.Q;
            const unsigned qc = ok ? q : (unsigned)a.Q - 1;
            const unsigned v0 = dp[(size_t)qc * a.ld], v1 = dp[(size_t)qc * a.ld + 16];
            d0[e] = ok ? v0 : 0u;                     // (pixels beyond the row end contribute nothing)
            d1[e] = ok ? v1 : 0u;
            gb[e] = s[2 * qc];
        }
        unsigned g16[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) g16[e] = c < 9 ? bfbits((float)gb[e]) : (c == 9 ? 0x3f80u : 0u);
        const bf16x8_t A0 = frag(bfpair(d0[0], d0[1]), bfpair(d0[2], d0[3]), bfpair(d0[4], d0[5]), bfpair(d0[6], d0[7]));
        const bf16x8_t A1 = frag(bfpair(d1[0], d1[1]), bfpair(d1[2], d1[3]), bfpair(d1[4], d1[5]), bfpair(d1[6], d1[7]));
        const bf16x8_t B = frag(bfpair(g16[0], g16[1]), bfpair(g16[2], g16[3]), bfpair(g16[4], g16[5]), bfpair(g16[6], g16[7]));
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0, B, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B, acc1, 0, 0, 0);
    }
    // lane (tap c, kg): acc0[v] = channel 4 * kg + v, acc1[v] = channel 16 + 4 * kg + v
    __shared__ float red[4][K1][16];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        red[wv][4 * kg + v][c] = acc0[v];
        red[wv][16 + 4 * kg + v][c] = acc1[v];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K1 * 10; i += 256) {
        const int k = i / 10, t = i - k * 10;
        a.part[(size_t)blockIdx.x * (K1 * 10) + i] = (red[0][k][t] + red[1][k][t]) + (red[2][k][t] + red[3][k][t]);
    }
}

// one block per output channel k: 64 row groups x 10 sums over the block partials (fp64, fixed order), then the 27 master-gradient
// entries of that channel
__global__ __launch_bounds__(640) void stem_u8_wgrad_reduce_kernel(const float* part, int nblk, const float* ab, float* dw, int accumulate) {
    __shared__ double s[64][10];
    __shared__ double tot[10];
    const int k = blockIdx.x, t = threadIdx.x % 10, rg = threadIdx.x / 10;
    double v = 0.0;
    int i = rg;
    for (; i + 192 < nblk; i += 256) {             // four rows per trip: independent loads, fixed-order adds
        float f[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) f[u] = part[((size_t)(i + 64 * u) * K1 + k) * 10 + t];
#pragma unroll
        for (int u = 0; u < 4; ++u) v += (double)f[u];
    }
    for (; i < nblk; i += 64) v += (double)part[((size_t)i * K1 + k) * 10 + t];
    s[rg][t] = v;
    __syncthreads();
    if (threadIdx.x < 10) {
        double z = 0.0;
        for (int r = 0; r < 64; ++r) z += s[r][threadIdx.x];
        tot[threadIdx.x] = z;
    }
    __syncthreads();
    if (threadIdx.x < 27) {
        const int tap = threadIdx.x / 3, c = threadIdx.x % 3;
        const float val = (float)((double)ab[c] * tot[tap] + (double)ab[3 + c] * tot[9]);
        float* o = dw + (size_t)k * 27 + threadIdx.x;
        *o = accumulate ? *o + val : val;
    }
}

bool stem_ok(const ifcbk_conv_desc* d) {
    return d && (d->dtype == IFCBK_BF16 || d->dtype == IFCBK_F32) && d->K == K1 && d->R == 3 && d->S == 3 && d->stride_h == 2 &&
           d->stride_w == 2 && d->pad_h == 0 && d->pad_w == 0 && d->Cw == 3 && d->N > 0 && d->H >= 3 && d->W >= 3 &&
           d->P == (d->H - 3) / 2 + 1 && d->Q == (d->W - 3) / 2 + 1 && d->ldy >= K1 && d->ldy % 8 == 0 &&
           (int64_t)d->N * d->P * d->Q < (1ll << 31) - PIXB;
}

StemArgs stem_args(const ifcbk_conv_desc* d, const uint8_t* g, const float* w, const float* ab) {
    StemArgs a = {};
    a.g = g; a.w = w; a.ab = ab;
    a.H = d->H; a.W = d->W; a.P = d->P; a.Q = d->Q; a.ld = d->ldy;
    a.PQ = (unsigned)(d->P * d->Q);
    a.M = (unsigned)((int64_t)d->N * d->P * d->Q);
    a.dPQ = make_fastdiv(a.PQ);
    a.dQ = make_fastdiv((unsigned)d->Q);
    a.rows = (unsigned)(d->N * d->P);
    a.gpr = (unsigned)cdiv(d->Q, 32);
    a.dP = make_fastdiv((unsigned)d->P);
    a.dG = make_fastdiv(a.gpr);
#ifdef IFCBK_EXPERIMENT_STEM
    a.dbg = getenv("IFCBK_STEM_DBG") ? atoi(getenv("IFCBK_STEM_DBG")) : 0;
#endif
    return a;
}

}  // namespace

// rows of the BatchNorm partial sums ifcbk_stem_u8_fwd writes (= its grid); 0: this descriptor is not served
// (bf16: the MFMA kernels, one block per RB output rows; fp32 parity mode: the vector-FMA kernels, one block per PIXB pixels)
static bool stem_mfma(const ifcbk_conv_desc* d) { return d->dtype == IFCBK_BF16 && d->Q >= 32; }
extern "C" int ifcbk_stem_u8_rows(const ifcbk_conv_desc* d) {
    if (!stem_ok(d)) return 0;
    return stem_mfma(d) ? cdiv((int64_t)d->N * d->P, RB) : cdiv((int64_t)d->N * d->P * d->Q, PIXB);
}

extern "C" size_t ifcbk_stem_u8_wgrad_workspace(const ifcbk_conv_desc* d) {
    return stem_ok(d) ? (size_t)ifcbk_stem_u8_rows(d) * K1 * 10 * sizeof(float) : 0;
}

extern "C" int ifcbk_stem_u8_fwd(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const uint8_t* g, const float* w_master, const float* ab,
                                 void* y, float* bn_part, const float* scale, const float* shift, int relu, void* stream) {
    if (!stem_ok(d)) IFCBK_FAIL(ctx, IFCBK_EUNSUPPORTED, "stem_u8_fwd: 3x3 / stride 2 / unpadded / 32 output channels / Cw = 3 only");
    if (!g || !w_master || !ab || !y || ((scale == nullptr) != (shift == nullptr))) IFCBK_FAIL(ctx, IFCBK_EINVAL, "stem_u8_fwd: null operand");
    StemArgs a = stem_args(d, g, w_master, ab);
    a.y = y; a.part = bn_part; a.scale = scale; a.shift = shift; a.relu = relu;
    const dim3 grid((unsigned)ifcbk_stem_u8_rows(d)), blk(256);
    hipStream_t st = (hipStream_t)stream;
    const bool f32 = d->dtype == IFCBK_F32;
    if (scale) {
        if (f32) hipLaunchKernelGGL((stem_u8_fwd_kernel<float, true>), grid, blk, 0, st, a);
        else if (stem_mfma(d)) hipLaunchKernelGGL((stem_u8_fwd_mfma_kernel<true>), grid, blk, 0, st, a);
        else hipLaunchKernelGGL((stem_u8_fwd_kernel<bf16_t, true>), grid, blk, 0, st, a);
    } else {
        if (f32) hipLaunchKernelGGL((stem_u8_fwd_kernel<float, false>), grid, blk, 0, st, a);
        else if (stem_mfma(d)) hipLaunchKernelGGL((stem_u8_fwd_mfma_kernel<false>), grid, blk, 0, st, a);
        else hipLaunchKernelGGL((stem_u8_fwd_kernel<bf16_t, false>), grid, blk, 0, st, a);
    }
    IFCBK_LAUNCH_CHECK(ctx, "stem_u8_fwd");
    return IFCBK_OK;
}

extern "C" int ifcbk_stem_u8_wgrad(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const uint8_t* g, const void* dy, const float* ab,
                                   float* dw, int accumulate, void* stream) {
    if (!stem_ok(d)) IFCBK_FAIL(ctx, IFCBK_EUNSUPPORTED, "stem_u8_wgrad: 3x3 / stride 2 / unpadded / 32 output channels / Cw = 3 only");
    if (!g || !dy || !ab || !dw) IFCBK_FAIL(ctx, IFCBK_EINVAL, "stem_u8_wgrad: null operand");
    const size_t need = ifcbk_stem_u8_wgrad_workspace(d);
    if (need > ctx->ws_bytes) IFCBK_FAIL(ctx, IFCBK_ENOMEM, "stem_u8_wgrad: workspace %zu > reserved %zu", need, ctx->ws_bytes);
    StemArgs a = stem_args(d, g, nullptr, ab);
    a.dy = dy; a.part = (float*)ctx->ws;
    const int nblk = ifcbk_stem_u8_rows(d);
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == IFCBK_F32) hipLaunchKernelGGL(stem_u8_wgrad_kernel<float>, dim3(nblk), dim3(256), 0, st, a);
    else if (stem_mfma(d)) hipLaunchKernelGGL(stem_u8_wgrad_mfma_kernel, dim3(nblk), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(stem_u8_wgrad_kernel<bf16_t>, dim3(nblk), dim3(256), 0, st, a);
    IFCBK_LAUNCH_CHECK(ctx, "stem_u8_wgrad");
    hipLaunchKernelGGL(stem_u8_wgrad_reduce_kernel, dim3(K1), dim3(640), 0, st, (const float*)ctx->ws, nblk, ab, dw, accumulate);
    IFCBK_LAUNCH_CHECK(ctx, "stem_u8_wgrad_reduce");
    return IFCBK_OK;
}
