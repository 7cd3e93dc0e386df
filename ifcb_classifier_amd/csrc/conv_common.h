// Shared by the implicit-GEMM convolution kernels (conv_igemm.hip, conv_big.hip): argument block, MFMA wrappers, LDS-DMA
// helpers and the common epilogue (C tile in LDS -> coalesced 16-byte row stores + fused statistics / affine / accumulate).
#pragma once
#include "common.h"

namespace {

struct ConvArgs {
    const void* x;
    const void* w;
    void* y;
    float* part;       // [mblocks][2][K] or null
    const float* ep_scale;   // optional fused epilogue: y = act(acc*scale[n] + shift[n] (+ residual))
    const float* ep_shift;
    const void* ep_res;
    int ep_ldr, ep_relu;
    // dgrad whose output is the gradient of ONE BatchNorm+ReLU activation: the epilogue also reduces that BN's backward
    // sums (sum dz, sum dz*xhat, dz = dy where the activation was positive) into `part` -- the BN backward then needs no
    // reduction pass of its own.  bs_raw: the BN's input (raw conv output of the producing layer), pixel stride bs_ld
    const void* bs_raw;
    const float *bs_mean, *bs_invstd, *bs_scale, *bs_shift;
    int bs_ld;
    // ... or, when the output gradient belongs to a CONCATENATION of several producers (an Inception block output consumed by the
    // next block's sibling GEMM): one entry per 8 output channels with that chunk's producer (raw tensor, statistics); raw == null:
    // the chunk has no BatchNorm producer (a pooled slice).  Sums land in part[mblock][2][K] at the chunk's own columns.
    const ifcbk_bs_chunk* bs_tab;
    unsigned xbytes, wbytes;   // buffer-descriptor extents of x and w
    int H, W, C, ldx;
    int K, R, S;
    int P, Q, ldy;
    int ostr_h, ostr_w, base_h, base_w, ish, isw;
    int M, Kg;
    int accumulate;
    // MODE 4 (eval-mode sibling GEMM): the output channels are cut into up to 4 segments with their own destination tensors
    int seg_n, seg_end[4], seg_ld[4], seg_aff[4];     // channel end (exclusive), pixel stride, 1 = affine + ReLU / 0 = raw
    void* seg_y[4];
    int PQ;
    int tilesN;
    // MODE 2 (one parity class of a stride-2 dgrad): sub-filter taps inside the full flipped filter, scattered output
    int wKg, wSfull, w_rbase, w_sbase;      // filter row length (elements), full S, first tap row / column (step 2)
    int oH, oW, o_a, o_b;                   // dx dims and the class parity: output pixel (n,i,j) -> (n, 2i+o_a, 2j+o_b)
    fastdiv_t fPQ, fQ;
    // conv_slab.hip, column-major tiles (7x1 filters): tile row m counts the output pixels in (n, q, p) order; its NHWC row is
    // n * PQ + p * Q + q
    int tr;
    fastdiv_t fP;
    int dbg;                                // timing-only diagnostics of conv_big.hip (0 in production)
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS image of a tile: [rows][8 chunks of 16 B] (128-byte rows: 64 bf16 or 32 fp32 of k); chunk c of row r lives at
// physical chunk c ^ (r & 7), which makes every ds_read_b128 fragment read conflict-free.  The image is filled by
// LDS-DMA (buffer_load_dwordx4 ... lds: destination = wave-uniform base + lane*16), so the swizzle is applied on the
// SOURCE side: the lane that lands on (row, phys) fetches logical chunk phys ^ (row & 7).
template <class T> struct Mma;
template <> struct Mma<bf16_t> {
    typedef bf16x8_t frag_t;                        // 8 consecutive k of one row
    __device__ static __forceinline__ void run(const frag_t& a, const frag_t& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ void pack4(bf16_t* p, const f32x4_t& v) {
        uint2 u;
        u.x = pack2bf(v[0], v[1]);
        u.y = pack2bf(v[2], v[3]);
        *reinterpret_cast<uint2*>(p) = u;
    }
};
// fp32 parity mode: v_mfma_f32_16x16x4_f32 (exact fp32 fma chain).  A lane's 16-byte fragment holds 4 k values; the
// four MFMAs of a fragment pair use element j of both operands, i.e. a permuted but CONSISTENT k order.
template <> struct Mma<float> {
    typedef f32x4_t frag_t;
    __device__ static __forceinline__ void run(const frag_t& a, const frag_t& b, f32x4_t& c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
    }
    __device__ static __forceinline__ void pack4(float* p, const f32x4_t& v) { *reinterpret_cast<f32x4_t*>(p) = v; }
};

// LDS-DMA with a run-time scalar offset (a __device__ helper: used directly in a __global__ template the host pass
// silently drops the kernel's stub)
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, lptr_t dst, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, voff, soff, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


// ---- epilogue, second half: the block's C tile sits in LDS as [BM][LDC] storage-type elements (LDC = BN + one chunk);
// every thread owns one 16-byte chunk column and walks down the rows: whole-row coalesced stores, and in the same pass the
// BatchNorm batch statistics of the ROUNDED outputs (forward), the BN-backward sums of the producing layer (MODE 3), the
// eval-mode affine (+residual) (+ReLU), the read-modify-write of an accumulating input gradient, the per-segment
// destinations of the eval sibling GEMM (MODE 4), the scatter of a stride-2 parity class (MODE 2).
template <class T, int BM, int BN, int NTHREADS, int MODE>
__device__ __forceinline__ void conv_epilogue_store(const ConvArgs& a, T* sC, float* sRed, const int t, const int lane, const int wave,
                                                    const int m0, const int n0, const int mtile) {
    constexpr bool BSTAT = MODE == 3 || MODE == 5;        // 3: one producer (a.bs_raw ...), 5: per-chunk producer table (a.bs_tab)
    constexpr bool BTAB = MODE == 5;
    constexpr int ES = (int)sizeof(T);
    constexpr int CE = 16 / ES;
    constexpr int NW = NTHREADS / 64;
    constexpr int CPR = BN / CE;
    constexpr int CPRP = CPR <= 4 ? 4 : CPR <= 8 ? 8 : CPR <= 16 ? 16 : 32;
    constexpr int LDC = BN + CE;
    {
        constexpr int RPP = NTHREADS / CPRP;
        const int cc = t & (CPRP - 1);
        const int r0 = t / CPRP;
        const bool cvalid = (cc < CPR) && (n0 + cc * CE < a.K);
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        f32x2_t s1p[CE / 2], s2p[CE / 2];
#pragma unroll
        for (int j = 0; j < CE / 2; ++j) s1p[j] = s2p[j] = f32x2_t{0.f, 0.f};
        if (cvalid) {
            const int nn = n0 + cc * CE;
            float bmu[CE], bis[CE], bsc[CE], bsh[CE];
            const T* braw = nullptr;                 // this thread's chunk of the producing BatchNorm's input, pixel 0
            int bld = 0;
            if (BSTAT) {
                if (BTAB) {
                    const ifcbk_bs_chunk e = a.bs_tab[nn >> 3];
                    const int off = nn & 7;
                    braw = e.raw ? (const T*)e.raw + off : nullptr;
                    bld = e.raw_ld;
#pragma unroll
                    for (int j = 0; j < CE; ++j) {
                        bmu[j] = braw ? e.stat[off + j] : 0.f;
                        bis[j] = braw ? e.stat[e.stat_ld + off + j] : 0.f;
                        bsc[j] = braw ? e.stat[2 * e.stat_ld + off + j] : 0.f;
                        bsh[j] = braw ? e.stat[3 * e.stat_ld + off + j] : 0.f;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < CE; ++j) {
                        bmu[j] = a.bs_mean[nn + j];
                        bis[j] = a.bs_invstd[nn + j];
                        bsc[j] = a.bs_scale[nn + j];
                        bsh[j] = a.bs_shift[nn + j];
                    }
                }
            }
            float sc[CE], sh[CE];
            if (a.ep_scale) {
#pragma unroll
                for (int j = 0; j < CE; ++j) {
                    sc[j] = a.ep_scale[nn + j];
                    sh[j] = a.ep_shift[nn + j];
                }
            }
            // MODE 4: this thread's chunk column belongs to one output segment (its own tensor, stride, affine or raw)
            T* segbase = (T*)a.y + nn;
            int segld = a.ldy;
            bool seg_aff = true;
            if (MODE == 4) {
                int si = 0;
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (q + 1 < a.seg_n && nn >= a.seg_end[q]) si = q + 1;
                segbase = (T*)a.seg_y[si] + (nn - (si ? a.seg_end[si - 1] : 0));
                segld = a.seg_ld[si];
                seg_aff = a.seg_aff[si] != 0;
            }
            // rows of this thread in batches of UB: the read-modify-write operands (accumulating dgrad, residual) of a
            // whole batch are requested before the first is used -- one exposed memory latency per batch instead of per row
            // (an accumulating 1x1 dgrad into a 288-channel block input ran at 1.5 TB/s with a load -> wait -> store loop)
            constexpr int RT = BM / RPP;
            // MODE 3 keeps batches of four raw rows (eight cost conv_igemm<5,..,3> its second resident block: 71 -> 125 us on the 8x8
            // layers); the table form (one block per CU on the wide tiles, nothing to lose) takes eight where the rows divide
            // (narrow tiles, <= 96 channels, gain from eight: 35x35 dgrads 0.113 -> 0.099 ms)
            constexpr int UB = BTAB ? (RT % 8 == 0 ? 8 : 4) : BSTAT ? ((RT % 8 == 0 && BN <= 96) ? 8 : 4) : (RT < 8 ? RT : 8);
            for (int b = 0; b < RT; b += UB) {
                typename Chunk<T>::raw_t pre[UB], prer[UB], prb[UB];
                size_t opx[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int r = r0 + (b + u) * RPP;
                    const int m = (RT % UB == 0 || b + u < RT) ? m0 + r : a.M;      // rows past the tile (RT not a multiple of UB)
                    opx[u] = (size_t)m;
                    if (MODE != 2 && a.tr && m < a.M) {
                        const uint32_t on = fdiv((uint32_t)m, a.fPQ);
                        const uint32_t orem = (uint32_t)m - on * a.fPQ.d;
                        const uint32_t oq = fdiv(orem, a.fP);
                        const uint32_t op = orem - oq * a.fP.d;
                        opx[u] = (size_t)on * a.fPQ.d + (size_t)op * a.Q + oq;
                    }
                    if (MODE == 2 && m < a.M) {
                        const uint32_t on = fdiv((uint32_t)m, a.fPQ);
                        const uint32_t orem = (uint32_t)m - on * a.fPQ.d;
                        const uint32_t oi = fdiv(orem, a.fQ);
                        const uint32_t oj = orem - oi * a.fQ.d;
                        opx[u] = ((size_t)on * a.oH + 2 * oi + a.o_a) * a.oW + 2 * oj + a.o_b;
                    }
                    if (m < a.M) {
                        if (a.accumulate) pre[u] = Chunk<T>::load_raw((const T*)a.y + opx[u] * a.ldy + nn);
                        if (a.ep_scale && a.ep_res) prer[u] = Chunk<T>::load_raw((const T*)a.ep_res + (MODE == 2 ? (size_t)m : opx[u]) * a.ep_ldr + nn);
                        if (BTAB) {
                            if (braw) prb[u] = Chunk<T>::load_raw(braw + opx[u] * bld);
                        } else if (BSTAT) {
                            prb[u] = Chunk<T>::load_raw((const T*)a.bs_raw + opx[u] * a.bs_ld + nn);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int r = r0 + (b + u) * RPP;
                    const int m = m0 + r;
                    if (m >= a.M || (RT % UB != 0 && b + u >= RT)) break;
                    // the chunk as it is stored (rounded for bf16): when nothing modifies it (training forward, first-writer
                    // dgrad) the raw bits go straight to memory; the statistics use packed fp32 math (v_pk_add / v_pk_fma)
                    const typename Chunk<T>::raw_t rawc = Chunk<T>::load_raw(sC + r * LDC + cc * CE);
                    T* dst = MODE == 4 ? segbase + opx[u] * segld : (T*)a.y + opx[u] * a.ldy + nn;
                    float fv[CE];
                    if (a.part || a.accumulate || a.ep_scale) Chunk<T>::widen(rawc, fv);
                    if (BSTAT && (!BTAB || braw)) {
                        float fx[CE];
                        Chunk<T>::widen(prb[u], fx);
#pragma unroll
                        for (int j = 0; j < CE; j += 2) {
                            const float d0 = (fx[j] * bsc[j] + bsh[j]) > 0.f ? fv[j] : 0.f;
                            const float d1 = (fx[j + 1] * bsc[j + 1] + bsh[j + 1]) > 0.f ? fv[j + 1] : 0.f;
                            const f32x2_t dz = {d0, d1};
                            const f32x2_t xh = {(fx[j] - bmu[j]) * bis[j], (fx[j + 1] - bmu[j + 1]) * bis[j + 1]};
                            s1p[j / 2] += dz;
                            s2p[j / 2] += dz * xh;
                        }
                    } else if (!BSTAT && a.part) {
#pragma unroll
                        for (int j = 0; j < CE; j += 2) {
                            const f32x2_t v = {fv[j], fv[j + 1]};
                            s1p[j / 2] += v;
                            s2p[j / 2] += v * v;
                        }
                    }
                    if (!(a.accumulate || a.ep_scale) || (MODE == 4 && !seg_aff)) {
                        *reinterpret_cast<typename Chunk<T>::raw_t*>(dst) = rawc;
                        continue;
                    }
                    if (a.accumulate) {
                        float fo[CE];
                        Chunk<T>::widen(pre[u], fo);
#pragma unroll
                        for (int j = 0; j < CE; ++j) fv[j] += fo[j];
                    }
                    if (a.ep_scale) {
                        if (a.ep_res) {
                            float fr[CE];
                            Chunk<T>::widen(prer[u], fr);
#pragma unroll
                            for (int j = 0; j < CE; ++j) fv[j] = fv[j] * sc[j] + sh[j] + fr[j];
                        } else {
#pragma unroll
                            for (int j = 0; j < CE; ++j) fv[j] = fv[j] * sc[j] + sh[j];
                        }
                        if (a.ep_relu) {
#pragma unroll
                            for (int j = 0; j < CE; ++j) fv[j] = fmaxf(fv[j], 0.f);
                        }
                    }
                    Chunk<T>::store(dst, fv);
                }
            }
        }
        if (a.part) {
            float s1[CE], s2[CE];
#pragma unroll
            for (int j = 0; j < CE; ++j) {
                s1[j] = s1p[j / 2][j & 1];
                s2[j] = s2p[j / 2][j & 1];
            }
#pragma unroll
            for (int off = CPRP; off < 64; off <<= 1)
#pragma unroll
                for (int j = 0; j < CE; ++j) {
                    s1[j] += __shfl_xor(s1[j], off);
                    s2[j] += __shfl_xor(s2[j], off);
                }
            if (lane < CPRP && cc < CPR) {
#pragma unroll
                for (int j = 0; j < CE; ++j) {
                    sRed[(wave * 2 + 0) * BN + cc * CE + j] = s1[j];
                    sRed[(wave * 2 + 1) * BN + cc * CE + j] = s2[j];
                }
            }
            __syncthreads();
            for (int i = t; i < 2 * BN; i += NTHREADS) {
                int which = i / BN, n = i - which * BN;
                if (n0 + n < a.K) {
                    float s = 0.f;
#pragma unroll
                    for (int w = 0; w < NW; ++w) s += sRed[(w * 2 + which) * BN + n];
                    a.part[((size_t)mtile * 2 + which) * a.K + n0 + n] = s;
                }
            }
        }
    }
}

}  // namespace
