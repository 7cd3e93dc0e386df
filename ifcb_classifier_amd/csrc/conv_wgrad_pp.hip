// Convolution weight gradient, wide tile, ping-pong (bf16, gfx950) -- the structure of conv_big.hip applied to
//   dW[k][j] = sum_pix dy[pix][k] * X[pix][j]        k = output channel, j = (r,s,c) filter column, X = gathered input
// Block tile: (32*KH) output channels x 256 filter columns, 64 pixels per K-step, ONE 512-thread block per CU, split-K over pixel
// ranges into fp32 slabs (summed in a fixed order by wgrad_reduce of conv_wgrad.hip).  conv_wgrad_rows moves 32 LDS-DMA pieces
// per 0.79-1.05 M MACs (96..128 x 128 tiles, two blocks per CU) and sits at ~20 % MFMA utilisation; this tile moves 48-56
// pieces per 2.1-3.1 M MACs.
//
// Eight waves = two groups (wave >> 2).  Group G owns the output-channel half [G*16*KH, (G+1)*16*KH) and column quarter
// wc = wave & 3 (64 columns = 4 tiles): KH x 4 accumulator tiles per wave.  A K-step has two PHASES kh = 0, 1: the 32-pixel
// halves of the step (one 16x16x32 MFMA per tile and phase).  The groups run half a phase apart (see conv_big.hip): while one
// multiplies, the other reads fragments and issues LDS-DMA.
//
// LDS: two K-step buffers (step parity) of NSA + 4 sub-tiles [64 pixels][64 channels] (128-byte rows, 16-byte chunk c of row r
// at physical chunk c ^ (r & 7), swizzle applied on the DMA source side).  Both operands are pixel-major, the reduction index is
// the ROW: fragments come from ds_read_b64_tr_b16 (transposing read): lane (g, lq, lp) reads rows 4g+lq and 16+4g+lq of a
// 32-pixel half, 4 channels at 4*lp -- a half-wave then touches 8 consecutive rows x 32 bytes = 16 distinct 16-byte slots.
//
// Staging.  Wave w loads pixel rows 8w..8w+7 of every sub-tile, i.e. rows of half G = w >> 2, read only in phases kh = G.
// Its NP = NSA + 4 pieces of a step go out in two batches (P1 = ceil(NP/2), P2 = NP - P1) in consecutive phases:
//     group 0:  (t,1) -> P1 of step t+2,   (t+1,0) -> P2 of step t+2        group 1:  (t,0) -> P1 of step t+1,  (t,1) -> P2 of step t+1
// and the wave waits vmcnt(P1) at the end of the load part in which it issued a P1 batch: the step issued before is complete one
// phase before its first reader.  WAR: a wave's fragment reads are COMPLETE (lgkmcnt(0)) before it meets the barrier that ends
// its load part, so rows read in phase g may be re-filled from phase g+1 on (conv_big needs g+2: it waits behind the barrier).
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace {

struct WppArgs {
    const void* x;
    const void* dy;
    float* slab;      // [nsplit][K][RSC]
    unsigned xbytes, dybytes;
    int H, W, C, ldx;
    int K, S;
    int ldy;
    int sh, sw, ph, pw;
    int M, RSC;
    int split_len;    // pixels per split (multiple of 64)
    int tilesJ, tiles;
    fastdiv_t fPQ, fQ;
};

typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void wpp_dma16(__amdgpu_buffer_rsrc_t rs, lptr_t dst, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, voff, 0, 0, 0);
}
template <int N>
__device__ __forceinline__ void wpp_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#define WPP_TR(lo, hi, addr, OFF)                                                                                   \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                       \
                 : "=&v"(lo), "=&v"(hi)                                                                             \
                 : "v"(addr), "n"(OFF), "n"((OFF) + 16 * 128));

// DM = 1: the LDS-DMA pieces are issued from INSIDE the MFMA cluster (one piece after each row of KH MFMAs) instead of the load
// part: the matrix pipe leaves the wave's scalar / vector-memory issue ports idle half of the time, while the load part is what
// bounds the ping-pong (conv_big.hip's measurements)
template <int KH, int DM>
__device__ __forceinline__ void wpp_run(const WppArgs& a, const int lin) {
    constexpr int BMK = 32 * KH;                     // output channels per block
    constexpr int NSA = (BMK + 63) / 64;             // dy sub-tiles
    constexpr int NP = NSA + 4;                      // LDS-DMA pieces per wave and K-step
    constexpr int P1 = (NP + 1) / 2, P2 = NP - P1;
    constexpr int SUB = 64 * 128;                    // bytes per sub-tile
    constexpr int PARB = NP * SUB;                   // bytes per K-step buffer
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PARB];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int grp = wave >> 2, wc = wave & 3;
    const int split = lin / a.tiles;
    const int tile = lin - split * a.tiles;
    const int ktile = tile / a.tilesJ, jtile = tile - ktile * a.tilesJ;
    const int k0 = ktile * BMK, j0 = jtile * 256;
    const int pix_begin = split * a.split_len;
    const int pix_end = min(pix_begin + a.split_len, a.M);
    const int nsteps = (pix_end - pix_begin + 63) / 64;

    // ---- LDS-DMA roles: lane -> (row 8*wave + (l>>3), physical chunk l&7), logical chunk (l&7) ^ (l>>3)
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    constexpr int FAR = 1 << 24;
    const int lrow8 = lane >> 3;
    const int csrc = (lane & 7) ^ lrow8;
    unsigned acol[NSA];                              // byte offset of this lane's dy chunk inside a pixel row, or OOB
#pragma unroll
    for (int s = 0; s < NSA; ++s) {
        const int ch = s * 64 + csrc * 8;
        acol[s] = (ch < BMK && k0 + ch < a.K) ? (unsigned)(k0 + ch) * 2u : OOB;
    }
    int bcol_r[4], bcol_s[4], btap[4];               // filter tap (r, s) and byte offset of this lane's x chunk, per sub-tile
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int jcol = j0 + s * 64 + csrc * 8;
        const bool bv = jcol < a.RSC;
        const int jj = bv ? jcol : 0;
        const int rs = jj / a.C;
        const int c = jj - rs * a.C;
        const int r = rs / a.S;
        const int sx = rs - r * a.S;
        bcol_r[s] = bv ? r : FAR;
        bcol_s[s] = sx;
        btap[s] = ((r * a.W + sx) * a.ldx + c) * 2;
    }
    // the pixel this lane gathers for the step its wave is staging
    int ph0 = 0, pw0 = 0, pxoff = 0;
    unsigned pdyoff = OOB;
    int tgt = 0;                                     // step being staged
#define WPP_DECODE()                                                                                                \
    {                                                                                                               \
        const int pix = pix_begin + tgt * 64 + wave * 8 + lrow8;                                                    \
        const bool pv = tgt < nsteps && pix < pix_end;                                                              \
        const uint32_t pp = pv ? (uint32_t)pix : 0u;                                                                \
        const uint32_t n = fdiv(pp, a.fPQ);                                                                         \
        const uint32_t rem = pp - n * a.fPQ.d;                                                                      \
        const uint32_t p = fdiv(rem, a.fQ);                                                                         \
        const uint32_t q = rem - p * a.fQ.d;                                                                        \
        ph0 = pv ? (int)p * a.sh - a.ph : -FAR;                                                                     \
        pw0 = (int)q * a.sw - a.pw;                                                                                 \
        pxoff = (((int)n * a.H + ((int)p * a.sh - a.ph)) * a.W + pw0) * a.ldx * 2;                                  \
        pdyoff = pv ? pp * (uint32_t)a.ldy * 2u : OOB;                                                              \
    }
    // piece I of the step `tgt`: I < NSA -> dy sub-tile I, else x sub-tile I - NSA
#define WPP_PIECE(I)                                                                                                \
    {                                                                                                               \
        unsigned char* dst = smem + (tgt & 1) * PARB + (I) * SUB + wave * 1024;                                     \
        if ((I) < NSA) {                                                                                            \
            const unsigned vo = (pdyoff != OOB && acol[(I) < NSA ? (I) : 0] != OOB) ? pdyoff + acol[(I) < NSA ? (I) : 0] : OOB; \
            wpp_dma16(rsA, (lptr_t)dst, vo);                                                                        \
        } else {                                                                                                    \
            constexpr int sb = (I) < NSA ? 0 : (I) - NSA;                                                           \
            const bool v = (unsigned)(ph0 + bcol_r[sb]) < (unsigned)a.H && (unsigned)(pw0 + bcol_s[sb]) < (unsigned)a.W; \
            wpp_dma16(rsB, (lptr_t)dst, v ? (unsigned)(pxoff + btap[sb]) : OOB);                                    \
        }                                                                                                           \
    }
#define WPP_BATCH1()                                                                                                \
    {                                                                                                               \
        WPP_DECODE()                                                                                                \
        WPP_PIECE(0) WPP_PIECE(1) WPP_PIECE(2)                                                                      \
        if constexpr (P1 > 3) WPP_PIECE(3)                                                                          \
    }
#define WPP_BATCH2()                                                                                                \
    {                                                                                                               \
        WPP_PIECE(P1) WPP_PIECE(P1 + 1) WPP_PIECE(P1 + 2)                                                           \
        ++tgt;                                                                                                      \
    }
    static_assert(P1 >= 3 && P1 <= 4 && P2 == 3, "batch macros");

    // ---- prologue: step 0 completely; group 0 is one batch ahead (its first batch of step 1)
    WPP_BATCH1() WPP_BATCH2()
    if (grp == 0) WPP_BATCH1()

    // ---- fragment addresses: tile-in-sub-tile ti (16 channels = chunks 2ti, 2ti+1); lane (g, lq, lp)
    const int g = lane >> 4, lq = (lane >> 2) & 3, lp = lane & 3;
    const int row0 = 4 * g + lq, rk = row0 & 7;
    unsigned tiaddr[4];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) tiaddr[ti] = (unsigned)(row0 * 128 + (((2 * ti + (lp >> 1)) ^ rk) * 16) + (lp & 1) * 8);
    const unsigned sbase = (unsigned)(size_t)(lptr_t)smem;
    unsigned xaddr[4], daddr[KH];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) xaddr[jt] = sbase + (NSA + wc) * SUB + tiaddr[jt];
#pragma unroll
    for (int kt = 0; kt < KH; ++kt) {
        const int ch = grp * 16 * KH + kt * 16;
        const int ti = (ch >> 4) & 3;
        // (runtime select of the per-lane tile address: four candidates, wave-uniform ti)
        const unsigned ta = ti == 0 ? tiaddr[0] : ti == 1 ? tiaddr[1] : ti == 2 ? tiaddr[2] : tiaddr[3];
        daddr[kt] = sbase + (ch >> 6) * SUB + ta;
    }

    f32x4_t acc[4][KH];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < KH; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (grp == 0) wpp_wait_vmcnt<P1>(); else wpp_wait_vmcnt<0>();     // step 0 has landed (group 0: its step-1 batch may fly)
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();        // group 1 runs one barrier behind group 0

// batch piece for MFMA-row slot SL of the phase (DM = 1): batch 1 = pieces [0, P1), batch 2 = pieces [P1, NP)
#define WPP_SLOT(FIRST, SL)                                                                                         \
    {                                                                                                               \
        if (FIRST) {                                                                                                \
            if ((SL) == 0) { WPP_DECODE() }                                                                         \
            if ((SL) == 0) WPP_PIECE(0)                                                                             \
            if ((SL) == 1) WPP_PIECE(1)                                                                             \
            if ((SL) == 2) WPP_PIECE(2)                                                                             \
            if constexpr (P1 > 3) { if ((SL) == 3) WPP_PIECE(3) }                                                   \
        } else {                                                                                                    \
            if ((SL) == 0) WPP_PIECE(P1)                                                                            \
            if ((SL) == 1) WPP_PIECE(P1 + 1)                                                                        \
            if ((SL) == 2) { WPP_PIECE(P1 + 2) ++tgt; }                                                             \
        }                                                                                                           \
    }
#define WPP_PHASE(KHV)                                                                                              \
    {                                                                                                               \
        s16x4_t xlo[4], xhi[4], dlo[KH], dhi[KH];                                                                   \
        _Pragma("unroll") for (int jt = 0; jt < 4; ++jt) WPP_TR(xlo[jt], xhi[jt], xaddr[jt] + paroff, (KHV) * 32 * 128) \
        _Pragma("unroll") for (int kt = 0; kt < KH; ++kt) WPP_TR(dlo[kt], dhi[kt], daddr[kt] + paroff, (KHV) * 32 * 128) \
        if (DM) {                                                                                                   \
            /* the step whose last pieces went out in the previous MFMA cluster must be complete before the next phase */ \
            if (grp == ((KHV) == 1 ? 0 : 1)) wpp_wait_vmcnt<0>();                                                   \
        } else if (grp == 0) {                                                                                      \
            if ((KHV) == 1) { WPP_BATCH1() wpp_wait_vmcnt<P1>(); } else { WPP_BATCH2() }                            \
        } else {                                                                                                    \
            if ((KHV) == 0) { WPP_BATCH1() wpp_wait_vmcnt<P1>(); } else { WPP_BATCH2() }                            \
        }                                                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        /* reads complete BEFORE the barrier (WAR, see header) */ \
        _Pragma("unroll") for (int jt = 0; jt < 4; ++jt) asm volatile("" : "+v"(xlo[jt]), "+v"(xhi[jt]));          \
        _Pragma("unroll") for (int kt = 0; kt < KH; ++kt) asm volatile("" : "+v"(dlo[kt]), "+v"(dhi[kt]));         \
        __builtin_amdgcn_s_barrier();                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                              \
        {                                                                                                           \
            bf16x8_t fx[4], fd[KH];                                                                                 \
            _Pragma("unroll") for (int jt = 0; jt < 4; ++jt)                                                        \
                fx[jt] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(xlo[jt], xhi[jt], 0, 1, 2, 3, 4, 5, 6, 7)); \
            _Pragma("unroll") for (int kt = 0; kt < KH; ++kt)                                                       \
                fd[kt] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(dlo[kt], dhi[kt], 0, 1, 2, 3, 4, 5, 6, 7)); \
            const bool first = grp == 0 ? (KHV) == 1 : (KHV) == 0;      /* which batch this group issues in this phase */ \
            _Pragma("unroll") for (int jt = 0; jt < 4; ++jt) {                                                      \
                _Pragma("unroll") for (int kt = 0; kt < KH; ++kt)                                                   \
                    acc[jt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[jt], fd[kt], acc[jt][kt], 0, 0, 0);    \
                if (DM) {                                                                                           \
                    __builtin_amdgcn_sched_barrier(0);                                                              \
                    if (jt == 0) WPP_SLOT(first, 0)                                                                 \
                    if (jt == 1) WPP_SLOT(first, 1)                                                                 \
                    if (jt == 2) WPP_SLOT(first, 2)                                                                 \
                    if (jt == 3) WPP_SLOT(first, 3)                                                                 \
                    __builtin_amdgcn_sched_barrier(0);                                                              \
                }                                                                                                   \
            }                                                                                                       \
        }                                                                                                           \
        __builtin_amdgcn_s_setprio(0);                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        __builtin_amdgcn_s_barrier();                                                                               \
        asm volatile("" ::: "memory");                                                                              \
    }

    for (int st = 0; st < nsteps; ++st) {
        const unsigned paroff = (unsigned)(st & 1) * PARB;
        WPP_PHASE(0)
        WPP_PHASE(1)
    }
#undef WPP_PHASE
#undef WPP_SLOT
#undef WPP_BATCH1
#undef WPP_BATCH2
#undef WPP_PIECE
#undef WPP_DECODE
    if (grp == 0) __builtin_amdgcn_s_barrier();
    wpp_wait_vmcnt<0>();                               // the tail's dummy pieces

    // ---- slab store: lane holds 4 consecutive columns j of one output channel k per tile
    float* out = a.slab + (size_t)split * a.K * a.RSC;
    const int kl = lane & 15, jq = (lane >> 4) * 4;
#pragma unroll
    for (int kt = 0; kt < KH; ++kt) {
        const int k = k0 + grp * 16 * KH + kt * 16 + kl;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            const int j = j0 + wc * 64 + jt * 16 + jq;
            if (k < a.K && j < a.RSC) *reinterpret_cast<f32x4_t*>(out + (size_t)k * a.RSC + j) = acc[jt][kt];
        }
    }
}

template <int KH, int DM>
__global__ __launch_bounds__(512) void conv_wgrad_pp(WppArgs a) {
    wpp_run<KH, DM>(a, (int)xcd_remap(blockIdx.x, gridDim.x));
}

// ---- grouped launch: ONE grid over the weight gradients of up to WPP_MAXG layers (round 4).
// A 17x17 layer's dW is 192 x 1344: six tiles.  Alone it needs 42 pixel splits to fill the chip -- 252 blocks of 27 K-steps that
// each end in a 196 KB fp32 slab store (50 MB per launch written, then read back by wgrad_reduce: 1.68x the algorithmic bytes),
// and the prologue + slab store of a block are a quarter of its life.  The six 7-tap layers of one Inception-C block finish their
// d(raw) tensors one after the other, but nothing consumes a weight gradient before the optimizer: run them as ONE grid, 36 tiles
// x 7 splits -- the same 252 blocks, each with six times the K-steps, and one sixth of the slab bytes and reduce work per layer.
// Members may differ in everything but the tile template (KH); block -> member by the prefix table, then the single-layer body.
constexpr int WPP_MAXG = 8;
struct WppGroup {
    WppArgs a[WPP_MAXG];
    int blk0[WPP_MAXG + 1];
    int n;
};
template <int KH>
__global__ __launch_bounds__(512) void conv_wgrad_ppg(WppGroup g) {
    const int bid = (int)xcd_remap(blockIdx.x, gridDim.x);
    int gi = 0;
#pragma unroll
    for (int q = 1; q < WPP_MAXG; ++q)
        if (q < g.n && bid >= g.blk0[q]) gi = q;
    const WppArgs a = g.a[gi];
    wpp_run<KH, 0>(a, bid - g.blk0[gi]);
}

// slabs of every member -> its dW, one launch (the per-element sums are those of wgrad_reduce4: same order, same bits)
struct WppReduceGroup {
    const float* slab[WPP_MAXG];
    float* dw[WPP_MAXG];
    int nsplit[WPP_MAXG], total4[WPP_MAXG], blk0[WPP_MAXG + 1];
    int n, accumulate;
};
__global__ __launch_bounds__(256) void wgrad_reduce4g(WppReduceGroup g) {
    __shared__ float4 part[4][64];
    int gi = 0;
#pragma unroll
    for (int q = 1; q < WPP_MAXG; ++q)
        if (q < g.n && (int)blockIdx.x >= g.blk0[q]) gi = q;
    const int nsplit = g.nsplit[gi], total4 = g.total4[gi];
    const int o = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int i = ((int)blockIdx.x - g.blk0[gi]) * 64 + o;
    float4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < total4) {
        const float4* src = reinterpret_cast<const float4*>(g.slab[gi]) + i;
        const int64_t stride = total4;
        float4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
        int sp = sg;
        for (; sp + 4 < nsplit; sp += 8) {
            const float4 a = src[sp * stride], b = src[(sp + 4) * stride];
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
        }
        if (sp < nsplit) {
            const float4 a = src[sp * stride];
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        }
        s = float4{s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w};
    }
    part[sg][o] = s;
    __syncthreads();
    if (sg == 0 && i < total4) {
        const float4 p0 = part[0][o], p1 = part[1][o], p2 = part[2][o], p3 = part[3][o];
        float4 r = {(p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z),
                    (p0.w + p1.w) + (p2.w + p3.w)};
        float4* dst = reinterpret_cast<float4*>(g.dw[gi]) + i;
        if (g.accumulate) {
            const float4 d0 = *dst;
            r.x += d0.x; r.y += d0.y; r.z += d0.z; r.w += d0.w;
        }
        *dst = r;
    }
}

int wpp_mode() {
    const char* e = getenv("IFCBK_WGRAD_PP");
    return e ? atoi(e) : 1;
}

}  // namespace

// Plan: does the wide-tile kernel serve this weight gradient, with which channel tile (kh = 4, 5, 6: 128 / 160 / 192 output
// channels per block), how many pixel splits.  One block per CU: the splits fill the chip once.
bool ifcbk_wgrad_pp_plan(const ifcbk_conv_desc* d, int* kh_out, int* nsplit_out, int* split_len_out) {
    const int mode = wpp_mode();
    if (mode <= 0 || d->dtype != IFCBK_BF16 || d->Cw != d->C) return false;
    const int64_t M = (int64_t)d->N * d->P * d->Q;
    const int RSC = d->R * d->S * d->C;
    if (RSC % 4) return false;
    if (mode < 2 && (d->K < 128 || RSC < 192 || M < 8192)) return false;
    // channel tile: least padded work; ties -> the larger tile
    int best = 0;
    double bestc = 0;
    for (int kh = 4; kh <= 6; ++kh) {
        const double c = (double)cdiv(d->K, 32 * kh) * (32 * kh + 24);
        if (!best || c < bestc - 1e-9 || (c < bestc + 1e-9 && kh > best)) { best = kh; bestc = c; }
    }
    if (const char* e = getenv("IFCBK_WGRAD_PP_KH")) { const int f = atoi(e); if (f >= 4 && f <= 6) best = f; }
    const int tiles = cdiv(d->K, 32 * best) * cdiv(RSC, 256);
    const int cus = ifcbk_num_cus();
    const int64_t steps = (M + 63) / 64;
    int64_t ns = cus / tiles;
    const int64_t maxsplit = steps / 8 > 0 ? steps / 8 : 1;
    if (ns > maxsplit) ns = maxsplit;
    if (ns < 1) ns = 1;
    // measured at batch 256 against conv_wgrad_rows (scripts/wgrad_pp_check.py, interleaved rounds, op = kernel + reduce):
    // +4-13 % on the 17x17 layers with 192 output channels, +12 % Mixed_6a 3x3/s2, +31 % on the Mixed_7c sibling GEMM; SLOWER with
    // 128-channel tiles (0.81x), with fewer than ~20 K-steps per split (8x8 1x3 / 3x1, Mixed_7a 3x3/s2: 0.90-0.94x) and when the
    // grid leaves CUs idle.  Issuing the pieces from inside the MFMA cluster (DM = 1) was 5-10 % slower everywhere.
    if (mode < 2 && (best < 5 || (int64_t)tiles * ns < (3 * cus) / 4 || steps / ns < 20)) return false;
    const int64_t len = ((steps + ns - 1) / ns) * 64;
    ns = (M + len - 1) / len;
    *kh_out = best;
    *nsplit_out = (int)ns;
    *split_len_out = (int)len;
    return true;
}

static void wpp_fill(WppArgs& a, const ifcbk_conv_desc* d, const void* x, const void* dy, float* slab, int kh, int split_len) {
    const int es = 2;
    a.x = x; a.dy = dy; a.slab = slab;
    a.xbytes = (unsigned)((int64_t)d->N * d->H * d->W * d->ldx * es);
    a.dybytes = (unsigned)((int64_t)d->N * d->P * d->Q * d->ldy * es);
    a.H = d->H; a.W = d->W; a.C = d->C; a.ldx = d->ldx;
    a.K = d->K; a.S = d->S; a.ldy = d->ldy;
    a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w;
    a.M = d->N * d->P * d->Q; a.RSC = d->R * d->S * d->C;
    a.split_len = split_len;
    a.tilesJ = cdiv(a.RSC, 256);
    a.tiles = cdiv(d->K, 32 * kh) * a.tilesJ;
    a.fPQ = make_fastdiv(d->P * d->Q); a.fQ = make_fastdiv(d->Q);
}

int ifcbk_wgrad_pp_launch(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const void* x, const void* dy, float* slab, int kh, int nsplit,
                          int split_len, hipStream_t st) {
    WppArgs a;
    wpp_fill(a, d, x, dy, slab, kh, split_len);
    const dim3 grid((unsigned)(a.tiles * nsplit)), block(512);
    // (DM = 1 -- the pieces issued from inside the MFMA cluster -- was 5-10 % slower everywhere and is no longer instantiated)
    if (kh == 4) hipLaunchKernelGGL((conv_wgrad_pp<4, 0>), grid, block, 0, st, a);
    else if (kh == 5) hipLaunchKernelGGL((conv_wgrad_pp<5, 0>), grid, block, 0, st, a);
    else if (kh == 6) hipLaunchKernelGGL((conv_wgrad_pp<6, 0>), grid, block, 0, st, a);
    else IFCBK_FAIL(ctx, IFCBK_EINVAL, "wgrad_pp: kh=%d", kh);
    IFCBK_LAUNCH_CHECK(ctx, "conv_wgrad_pp");
    return 0;
}

// ---------------------------------------------------------------- grouped weight gradients (C-ABI: include/ifcbk.h)
namespace {
struct GroupPlan { int kh, nsplit[WPP_MAXG], split_len[WPP_MAXG], tiles[WPP_MAXG], blocks; size_t slab_off[WPP_MAXG], ws; };

// channel tile of a member (the single-layer rule of ifcbk_wgrad_pp_plan) or 0 when the wide-tile kernel cannot take the layer
constexpr int KH_FLAT = 16;      // pseudo channel tile: the member is a flat-slot layer (conv_wgrad_flat.hip)
int group_member_kh(const ifcbk_conv_desc* d) {
    if (d->dtype != IFCBK_BF16 || d->Cw != d->C) return 0;
    if (ifcbk_wgrad_flat_member(d)) return KH_FLAT;
    const int RSC = d->R * d->S * d->C;
    if (RSC % 4 || d->K % 8 || d->C % 8 || d->ldx % 8 || d->ldy % 8) return 0;
    if ((int64_t)d->N * d->P * d->Q * d->ldy * 2 >= (1ll << 31) || (int64_t)d->N * d->H * d->W * d->ldx * 2 >= (1ll << 31)) return 0;
    int best = 0;
    double bestc = 0;
    for (int kh = 4; kh <= 6; ++kh) {
        const double c = (double)cdiv(d->K, 32 * kh) * (32 * kh + 24);
        if (!best || c < bestc - 1e-9 || (c < bestc + 1e-9 && kh > best)) { best = kh; bestc = c; }
    }
    if (const char* e = getenv("IFCBK_WGRAD_PP_KH")) { const int f = atoi(e); if (f >= 4 && f <= 6) best = f; }
    if (wpp_mode() < 2) {
        // padded work of the (32*kh) x 256 tiling over the true K x RSC matrix: a 96 x 576 gradient (35x35 3x3 layers) would
        // multiply 1.8x its size; 128-channel tiles ran at 0.81x of conv_wgrad_rows as single launches (IFCBK_WGRAD_GROUP_MINKH)
        const double waste = (double)cdiv(d->K, 32 * best) * 32 * best * cdiv(RSC, 256) * 256 / ((double)d->K * RSC);
        int minkh = 5;
        if (const char* e = getenv("IFCBK_WGRAD_GROUP_MINKH")) minkh = atoi(e);
        if (waste > 1.3 || best < minkh) return 0;
    }
    return best;
}

// one grid for all members: the fewest K-steps per block L such that sum_i tiles_i * ceil(steps_i / L) fits the chip once
bool group_plan(int n, const ifcbk_conv_desc* ds, GroupPlan* gp) {
    if (n < 1 || n > WPP_MAXG) return false;
    const int mode = wpp_mode();
    if (mode <= 0) return false;
    int kh = 0;
    int64_t steps[WPP_MAXG];
    for (int i = 0; i < n; ++i) {
        const int k = group_member_kh(&ds[i]);
        if (!k || (kh && k != kh)) return false;
        kh = k;
        gp->tiles[i] = cdiv(ds[i].K, 32 * kh) * cdiv(ds[i].R * ds[i].S * ds[i].C, 256);
        steps[i] = ((int64_t)ds[i].N * ds[i].P * ds[i].Q + 63) / 64;      // (flat-slot members: a lower bound of their slot steps; only the acceptance test below reads it)
    }
    const int cus = ifcbk_num_cus();
    if (kh == KH_FLAT) {
        // flat-slot members: two 256-thread blocks per CU; one grid over all their filter rows
        if (!ifcbk_wgrad_flat_group_plan(n, ds, gp->nsplit, gp->split_len, gp->tiles, gp->slab_off, &gp->ws, &gp->blocks)) return false;
        gp->kh = KH_FLAT;
        if (mode < 2) {
            if (gp->blocks < (3 * 2 * cus) / 4) return false;
            for (int i = 0; i < n; ++i)
                if (steps[i] / gp->nsplit[i] < 20) return false;
        }
        return true;
    }
    int64_t lo = 1, hi = 1;
    for (int i = 0; i < n; ++i) hi = steps[i] > hi ? steps[i] : hi;
    auto blocks_at = [&](int64_t L) { int64_t b = 0; for (int i = 0; i < n; ++i) b += (int64_t)gp->tiles[i] * ((steps[i] + L - 1) / L); return b; };
    if (blocks_at(hi) > cus && mode < 2) return false;          // more tiles than CUs even unsplit: the single launches serve it
    while (lo < hi) {
        const int64_t mid = (lo + hi) / 2;
        if (blocks_at(mid) <= cus) hi = mid; else lo = mid + 1;
    }
    int64_t L = lo < 8 ? 8 : lo;                                 // (a split shorter than 8 K-steps is all prologue)
    size_t off = 0;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        int64_t ns = (steps[i] + L - 1) / L;
        const int64_t len = ((steps[i] + ns - 1) / ns) * 64;
        const int64_t M = (int64_t)ds[i].N * ds[i].P * ds[i].Q;
        ns = (M + len - 1) / len;
        gp->nsplit[i] = (int)ns;
        gp->split_len[i] = (int)len;
        gp->slab_off[i] = off;
        off += (size_t)ns * ds[i].K * ds[i].R * ds[i].S * ds[i].C * sizeof(float);
        off = (off + 255) & ~(size_t)255;
        blocks += gp->tiles[i] * (int)ns;
    }
    gp->kh = kh; gp->ws = off; gp->blocks = blocks;
    if (mode < 2) {
        // worth one grid only when it fills most of the chip with splits long enough to amortise a block's prologue + slab store
        if (blocks < (3 * cus) / 4) return false;
        for (int i = 0; i < n; ++i)
            if (steps[i] / gp->nsplit[i] < 20) return false;
    }
    return true;
}
}  // namespace

extern "C" int ifcbk_conv2d_wgrad_group_member_kh(const ifcbk_conv_desc* d) { return d ? group_member_kh(d) : 0; }

extern "C" size_t ifcbk_conv2d_wgrad_group_workspace(int n, const ifcbk_conv_desc* descs) {
    GroupPlan gp;
    if (!descs || !group_plan(n, descs, &gp)) return 0;
    return gp.ws;
}

extern "C" int ifcbk_conv2d_wgrad_group_info(int n, const ifcbk_conv_desc* descs, int* kh, int* blocks, int* nsplit) {
    GroupPlan gp;
    if (!descs || !group_plan(n, descs, &gp)) return IFCBK_EUNSUPPORTED;
    if (kh) *kh = gp.kh;
    if (blocks) *blocks = gp.blocks;
    if (nsplit) for (int i = 0; i < n; ++i) nsplit[i] = gp.nsplit[i];
    return IFCBK_OK;
}

extern "C" int ifcbk_conv2d_wgrad_group(ifcbk_ctx* ctx, int n, const ifcbk_conv_desc* descs, const void* const* xs, const void* const* dys,
                                        float* const* dws, int accumulate, void* stream) {
    if (!ctx || !descs || !xs || !dys || !dws) return IFCBK_EINVAL;
    GroupPlan gp;
    if (!group_plan(n, descs, &gp))
        IFCBK_FAIL(ctx, IFCBK_EUNSUPPORTED, "wgrad_group: %d members are not one wide-tile group (bf16, C == Cw, one channel tile)", n);
    if (gp.ws > ctx->ws_bytes) IFCBK_FAIL(ctx, IFCBK_ENOMEM, "wgrad_group: workspace %zu > reserved %zu (call ifcbk_ctx_reserve)", gp.ws, ctx->ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    WppGroup g;
    WppReduceGroup r;
    memset(&g, 0, sizeof(g));
    memset(&r, 0, sizeof(r));
    int blk = 0, rblk = 0;
    const bool flat = gp.kh == KH_FLAT;
    for (int i = 0; i < n; ++i) {
        if ((uintptr_t)dws[i] % 16) IFCBK_FAIL(ctx, IFCBK_EINVAL, "wgrad_group: dW of member %d is not 16-byte aligned", i);
        float* slab = (float*)((char*)ctx->ws + gp.slab_off[i]);
        if (!flat) wpp_fill(g.a[i], &descs[i], xs[i], dys[i], slab, gp.kh, gp.split_len[i]);
        g.blk0[i] = blk;
        blk += gp.tiles[i] * gp.nsplit[i];
        r.slab[i] = slab; r.dw[i] = dws[i]; r.nsplit[i] = gp.nsplit[i];
        r.total4[i] = (int)((int64_t)descs[i].K * descs[i].R * descs[i].S * descs[i].C / 4);
        r.blk0[i] = rblk;
        rblk += cdiv(r.total4[i], 64);
    }
    for (int i = n; i <= WPP_MAXG; ++i) { g.blk0[i] = blk; r.blk0[i] = rblk; }
    g.n = n; r.n = n; r.accumulate = accumulate;
    const dim3 grid((unsigned)blk), block(512);
    if (flat) { if (int e = ifcbk_wgrad_flat_group_launch(ctx, n, descs, xs, dys, gp.nsplit, gp.split_len, gp.slab_off, st)) return e; }
    else if (gp.kh == 4) hipLaunchKernelGGL((conv_wgrad_ppg<4>), grid, block, 0, st, g);
    else if (gp.kh == 5) hipLaunchKernelGGL((conv_wgrad_ppg<5>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((conv_wgrad_ppg<6>), grid, block, 0, st, g);
    IFCBK_LAUNCH_CHECK(ctx, "conv_wgrad_ppg");
    hipLaunchKernelGGL(wgrad_reduce4g, dim3((unsigned)rblk), dim3(256), 0, st, r);
    IFCBK_LAUNCH_CHECK(ctx, "wgrad_reduce4g");
    return 0;
}
