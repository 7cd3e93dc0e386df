// Convolution weight-gradient on MFMA (gfx950): dW[k][r,s,c] = sum_pix dy[pix][k] * x[gather(pix,r,s)][c].
//
// GEMM view: M' = output channels k, N' = (r,s,c) columns, reduction over pixels (up to 5.7 M).  Both
// operands are stored pixel-major (NHWC), i.e. the reduction index is the SLOW axis of both, which a
// k-contiguous MFMA fragment cannot read directly.  The tiles are staged untransposed (coalesced 16-byte
// chunks) and the fragments are read with ds_read_b64_tr_b16 -- gfx950's transposing LDS read -- so no
// transposed copy of any activation is ever written to HBM.
//
// Split-K over pixel ranges with fp32 partial slabs in the ctx workspace, summed in a fixed order by
// wgrad_reduce (bitwise reproducible; no float atomics).
#include "common.h"
#include <stdlib.h>

namespace {

struct WgradArgs {
    const void* x;
    const void* dy;
    float* slab;      // [nsplit][K][RSC]
    unsigned xbytes, dybytes;
    int H, W, C, ldx;
    int K, R, S;
    int P, Q, ldy;
    int sh, sw, ph, pw;
    int M;            // N*P*Q pixels
    int RSC;
    int split_len;    // pixels per split (multiple of BKP)
    int tilesN;       // column tiles
    int tiles;        // tilesM * tilesN
    fastdiv_t fPQ, fQ;
};

constexpr int NTHREADS = 256;
constexpr int BKP = 64;     // pixels per step
constexpr int BNW = 128;    // (r,s,c) columns per block
constexpr int TW = 128;     // LDS image width (elements) of BOTH tiles: [BKP rows][16 chunks of 16 B]
constexpr int TILE = BKP * TW;

typedef __attribute__((address_space(3))) void* lptr_t;

// ---------------------------------------------------------------- row-major tiles (K >= 96)
// Both tiles are pixel-major (the reduction index is the ROW), 256-byte rows, filled by LDS-DMA: one wave-instruction
// = 4 pixels x 256 contiguous bytes = whole cache lines.  16-byte chunk c of row r lives at physical chunk
// c ^ ((r & 7) << 1): the 8 rows one half-wave touches in a ds_read_b64_tr_b16 then fall into 8 different 32-byte
// bank slots (conflict-free transposing reads).
//
// What the first generation of this kernel taught (rocprofv3 SQ counters + s_memtime stamps on Mixed_6e 7x1, kept in
// DESIGN.md): it issued 11.5 VALU instructions per MFMA (per-lane pixel decode for every DMA row group, swizzled
// fragment addresses recomputed per read) -- the SIMDs were ~90 % busy ISSUING -- and the compiler put
// s_waitcnt vmcnt(0) in front of the first ds_read_b64_tr_b16 of every step (it orders every LDS access it can see
// behind ALL pending LDS-DMA), so the prefetch of tile k+1 never overlapped the math on tile k.  Hence:
//   * a wave's 16 pixels of a step are decoded ONCE (lane = pixel) into a 16-entry LDS table {h0, w0, x offset,
//     dy offset}; the four DMA row groups read their entry back with one broadcast ds_read_b128 each,
//   * the swizzled fragment addresses are per-lane constants (one VGPR per 16-column tile) and every
//     ds_read_b64_tr_b16 uses an immediate offset for (stage, tile, k half) -- the step loop is unrolled by two so
//     the stage is a compile-time constant,
//   * the fragment reads are inline asm (invisible to the LDS-DMA alias rule); the block waits for its own reads
//     (lgkmcnt(0)) and the stage being read was completed before the last barrier.
struct __attribute__((aligned(16))) PixEntry { int h0, w0, xoff, dyoff; };

template <int MT>
__global__ __launch_bounds__(NTHREADS, 2) void conv_wgrad_rows(WgradArgs a) {
    constexpr int BMW = 32 * MT;
    constexpr int CA = BMW / 8;                      // valid dy chunks per pixel row
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * 2 * TILE];     // [stage][A | B]
    __shared__ PixEntry ptab[4][16];                                         // [wave][row group j*4 + lrow4]

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // 1-D grid, split-major: all tiles of one pixel range sit on one XCD (they re-read the same pixels)
    const int lin = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int split = lin / a.tiles;
    const int tile = lin - split * a.tiles;
    const int mtile = tile / a.tilesN, ntile = tile - mtile * a.tilesN;
    const int k0 = mtile * BMW, n0 = ntile * BNW;
    const int pix_begin = split * a.split_len;
    const int pix_end = min(pix_begin + a.split_len, a.M);

    // LDS-DMA roles: wave-instruction j of wave w fills LDS rows (w*4+j)*4 .. +3; lane -> (row l>>4, phys chunk l&15)
    const int lrow4 = lane >> 4, phys = lane & 15;
    constexpr unsigned OOB = 0x80000000u;            // buffer range check: reads zeros
    constexpr int FAR = 1 << 24;                     // a row offset no valid input coordinate survives
    int acol[2], bcol_r[2], bcol_s[2], btap[2];
    bool avalid[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {              // logical chunk per instruction parity (row & 7 = (j&1)*4 + lrow4)
        const int c16 = phys ^ (((par * 4 + lrow4) & 7) << 1);
        avalid[par] = (c16 < CA) && (k0 + c16 * 8 < a.K);
        acol[par] = (k0 + c16 * 8) * 2;
        const int jcol = n0 + c16 * 8;
        const bool bv = jcol < a.RSC;
        const int jj = bv ? jcol : 0;
        const int rs = jj / a.C;
        const int c = jj - rs * a.C;
        const int r = rs / a.S;
        const int sx = rs - r * a.S;
        bcol_r[par] = bv ? r : FAR;
        bcol_s[par] = sx;
        btap[par] = ((r * a.W + sx) * a.ldx + c) * 2;
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    // the pixel this lane decodes for its wave: table slot (lane & 15) = j*4 + lrow4 -> tile row (wave*4 + j)*4 + lrow4
    const int slot = lane & 15;
    const int myrow = (wave * 4 + (slot >> 2)) * 4 + (slot & 3);
    PixEntry* const mytab = &ptab[wave][slot];
    const PixEntry* const rdtab = &ptab[wave][lrow4];      // + j*4

#define ISSUE_ROWS(pix0, stage)                                                                                 \
    {                                                                                                           \
        {                                                                                                       \
            const int pix = (pix0) + myrow;                                                                     \
            const bool pv = pix < pix_end;                                                                      \
            const uint32_t pp = pv ? (uint32_t)pix : 0u;                                                        \
            const uint32_t n = fdiv(pp, a.fPQ);                                                                 \
            const uint32_t rem = pp - n * a.fPQ.d;                                                              \
            const uint32_t p = fdiv(rem, a.fQ);                                                                 \
            const uint32_t q = rem - p * a.fQ.d;                                                                \
            PixEntry e;                                                                                         \
            e.h0 = pv ? (int)p * a.sh - a.ph : -FAR;                                                            \
            e.w0 = (int)q * a.sw - a.pw;                                                                        \
            e.xoff = (((int)n * a.H + ((int)p * a.sh - a.ph)) * a.W + e.w0) * a.ldx * 2;                        \
            e.dyoff = pv ? (int)(pp * (uint32_t)a.ldy * 2u) : (int)OOB;                                         \
            *mytab = e;       /* same-wave LDS traffic is ordered: no barrier between this store and the reads below */ \
        }                                                                                                       \
        bf16_t* dstA = smem + (stage) * 2 * TILE;                                                               \
        bf16_t* dstB = dstA + TILE;                                                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
            const int par = j & 1;                                                                              \
            const PixEntry e = rdtab[j * 4];                                                                    \
            const unsigned voA = avalid[par] ? (unsigned)e.dyoff + (unsigned)acol[par] : OOB;                   \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(dstA + (wave * 4 + j) * 4 * TW), 16, voA, 0, 0, 0); \
            const bool v = (unsigned)(e.h0 + bcol_r[par]) < (unsigned)a.H && (unsigned)(e.w0 + bcol_s[par]) < (unsigned)a.W; \
            const unsigned voB = v ? (unsigned)(e.xoff + btap[par]) : OOB;                                      \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)(dstB + (wave * 4 + j) * 4 * TW), 16, voB, 0, 0, 0); \
        }                                                                                                       \
    }

    f32x4_t acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nsteps = (pix_end - pix_begin + BKP - 1) / BKP;
    if (nsteps > 0) ISSUE_ROWS(pix_begin, 0)
    __syncthreads();

    // transposing fragment reads: lane (g, lq, lp) addresses LDS row 4g+lq (+32 per k half, +16 for the upper
    // registers), columns col0+4lp..+3; (row & 7) -- the swizzle key -- is the same for all of them
    const int g = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;
    const int trow = 4 * g + lq;
    const int swz = (trow & 7) << 1;
    unsigned fa0[MT], fb0[4];                        // LDS byte addresses
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int col = wm * (MT * 16) + mt * 16 + 4 * lp;
        fa0[mt] = (unsigned)(size_t)(lptr_t)(smem + trow * TW + (((col >> 3) ^ swz) << 3) + (col & 7));
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int col = wn * 64 + nt * 16 + 4 * lp;
        fb0[nt] = (unsigned)(size_t)(lptr_t)(smem + TILE + trow * TW + (((col >> 3) ^ swz) << 3) + (col & 7));
    }

#define TR_PAIR(lo, hi, addr, OFF)                                                                              \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                   \
                 : "=&v"(lo), "=&v"(hi)                                                                         \
                 : "v"(addr), "n"(OFF), "n"((OFF) + 16 * TW * 2));
#define MATH_ROWS(stage)                                                                                        \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                          \
        s16x4_t alo[MT], ahi[MT], blo[4], bhi[4];                                                               \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                     \
            if (kk == 0) { TR_PAIR(alo[mt], ahi[mt], fa0[mt], (stage) * 2 * TILE * 2) }                         \
            else { TR_PAIR(alo[mt], ahi[mt], fa0[mt], (stage) * 2 * TILE * 2 + 32 * TW * 2) }                   \
        }                                                                                                       \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                                      \
            if (kk == 0) { TR_PAIR(blo[nt], bhi[nt], fb0[nt], (stage) * 2 * TILE * 2) }                         \
            else { TR_PAIR(blo[nt], bhi[nt], fb0[nt], (stage) * 2 * TILE * 2 + 32 * TW * 2) }                   \
        }                                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
        /* empty volatile asms keep their order after the wait and make every fragment (hence every MFMA) depend on it */ \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(alo[mt]), "+v"(ahi[mt]));      \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) asm volatile("" : "+v"(blo[nt]), "+v"(bhi[nt]));       \
        bf16x8_t fa[MT], fb[4];                                                                                 \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                       \
            fa[mt] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(alo[mt], ahi[mt], 0, 1, 2, 3, 4, 5, 6, 7)); \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                        \
            fb[nt] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(blo[nt], bhi[nt], 0, 1, 2, 3, 4, 5, 6, 7)); \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                       \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                    \
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mt], fb[nt], acc[mt][nt], 0, 0, 0);    \
    }

    // __syncthreads() = s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier: the tile issued at the top of a step has landed
    // (in every wave) before the next step reads it, and nobody still reads the stage the next issue overwrites
    for (int st = 0; st < nsteps; st += 2) {
        if (st + 1 < nsteps) ISSUE_ROWS(pix_begin + (st + 1) * BKP, 1)
        MATH_ROWS(0)
        __syncthreads();
        if (st + 1 >= nsteps) break;
        if (st + 2 < nsteps) ISSUE_ROWS(pix_begin + (st + 2) * BKP, 0)
        MATH_ROWS(1)
        __syncthreads();
    }
#undef ISSUE_ROWS
#undef MATH_ROWS
#undef TR_PAIR

    // slab store: lane holds rows k = 4g+j, column l&15
    float* out = a.slab + (size_t)split * a.K * a.RSC;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            int col = n0 + wn * 64 + nt * 16 + (lane & 15);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int k = k0 + wm * (MT * 16) + mt * 16 + 4 * g + j;
                if (k < a.K && col < a.RSC) out[(size_t)k * a.RSC + col] = acc[mt][nt][j];
            }
        }
}

// ---------------------------------------------------------------- chunk-column tiles (K <= 64)
// Same GEMM and the same transposing fragment reads, but the LDS image of a step is stored CHUNK-COLUMN major:
// [column of 8 channels][64 pixels][16 B].  One LDS-DMA wave-instruction then fills one column -- lane = pixel -- so
//   * a lane decodes ONE pixel (n,p,q) per step instead of four,
//   * the (r,s,c) tap of a column is wave-uniform (SGPRs, set up once per block),
//   * dy columns need no per-lane math at all (per-pixel offset + scalar column offset).
// The leanest instruction stream, but a DMA instruction now touches 64 cache lines (16 B of each) instead of 8: it
// wins where the tile has few MFMAs per step to hide instructions behind (K <= 64: the 149^2 / 147^2 stem layers, 1.1-1.5x)
// and loses 15-30 % on the wider tiles (measured per layer, scripts/conv_layers.py).
// Columns are 1152 B apart (1024 + 128 skew): the 32 lanes of a ds_read_b64_tr_b16 group then touch 32 distinct
// 8-byte bank pairs (offsets (lp>>1)*128 + g*64 + lq*16 + (lp&1)*8 mod 256).
constexpr int CSE = 576;    // column stride in bf16 elements (1152 B)

// (a __device__ helper: with a run-time scalar offset the builtin is rejected -- silently, the kernel's host stub is
// simply not emitted -- when it appears directly in a __global__ template that the host pass instantiates)
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, lptr_t dst, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, voff, soff, 0, 0);
}

// NTW: 16-column tiles per wave along (r,s,c): block tile = (32*MT) output channels x (32*NTW) columns
template <int MT, int NTW>
__global__ __launch_bounds__(NTHREADS) void conv_wgrad_cols(WgradArgs a) {
    constexpr int BMW = 32 * MT;
    constexpr int CA = 4 * MT;                       // dy chunk columns of the block tile
    constexpr int CB = 4 * NTW;                      // x chunk columns of the block tile
    constexpr int BNC = 32 * NTW;
    constexpr int STAGE = (CA + CB) * CSE;           // elements per pipeline stage
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * STAGE];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lin = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int split = lin / a.tiles;
    const int tile = lin - split * a.tiles;
    const int mtile = tile / a.tilesN, ntile = tile - mtile * a.tilesN;
    const int k0 = mtile * BMW, n0 = ntile * BNC;
    const int pix_begin = split * a.split_len;
    const int pix_end = min(pix_begin + a.split_len, a.M);

    // columns owned by this wave: dy columns ca = wave + 4i (i < MT), x columns cb = wave + 4i (i < NTW)
    int soffA[MT];
    bool colA[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int kc = k0 + (wave + 4 * i) * 8;
        colA[i] = kc < a.K;
        soffA[i] = kc * 2;
    }
    int rB[NTW], sB[NTW], tapB[NTW];
    bool colB[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int jcol = n0 + (wave + 4 * i) * 8;
        colB[i] = jcol < a.RSC;
        const int jj = colB[i] ? jcol : 0;
        const int rs = jj / a.C;
        const int c = jj - rs * a.C;
        const int r = rs / a.S;
        const int sx = rs - r * a.S;
        rB[i] = __builtin_amdgcn_readfirstlane(r);
        sB[i] = __builtin_amdgcn_readfirstlane(sx);
        tapB[i] = __builtin_amdgcn_readfirstlane(((r * a.W + sx) * a.ldx + c) * 2);
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const bool nopad = (a.ph | a.pw) == 0;           // every tap of every output pixel is inside the input

#define ISSUE_COLS(pix0, stage)                                                                                 \
    {                                                                                                           \
        bf16_t* dst = smem + (stage) * STAGE;                                                                   \
        const int pix = (pix0) + lane;                                                                          \
        const bool pv = pix < pix_end;                                                                          \
        const unsigned voA = pv ? (unsigned)(pix * a.ldy) * 2u : OOB;                                           \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                          \
            if (colA[i])                                                                                        \
                lds_dma16(rsA, (lptr_t)(dst + (wave + 4 * i) * CSE), voA, soffA[i]); \
        const uint32_t pp = pv ? (uint32_t)pix : 0u;                                                            \
        const uint32_t n = fdiv(pp, a.fPQ);                                                                     \
        const uint32_t rem = pp - n * a.fPQ.d;                                                                  \
        const uint32_t p = fdiv(rem, a.fQ);                                                                     \
        const uint32_t q = rem - p * a.fQ.d;                                                                    \
        const int h0 = (int)p * a.sh - a.ph, w0 = (int)q * a.sw - a.pw;                                         \
        const int pixbase = (((int)n * a.H + h0) * a.W + w0) * a.ldx * 2;                                       \
        _Pragma("unroll") for (int i = 0; i < NTW; ++i)                                                         \
            if (colB[i]) {                                                                                      \
                bool v = pv;                                                                                    \
                if (!nopad) v = v && (unsigned)(h0 + rB[i]) < (unsigned)a.H && (unsigned)(w0 + sB[i]) < (unsigned)a.W; \
                const unsigned vo = v ? (unsigned)(pixbase + tapB[i]) : OOB;                                    \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)(dst + (CA + wave + 4 * i) * CSE), 16, vo, 0, 0, 0); \
            }                                                                                                   \
    }

    f32x4_t acc[MT][NTW];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nsteps = (pix_end - pix_begin + BKP - 1) / BKP;
    if (nsteps > 0) ISSUE_COLS(pix_begin, 0)
    __syncthreads();

    // transposing fragment reads: lane (g, lq, lp) addresses pixel 4g+lq (then +16), channels col0+4lp..+3
    const int g = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;
    const int lane_off = (lp >> 1) * CSE + (4 * g + lq) * 8 + (lp & 1) * 4;
    const bf16_t* pA = smem + wm * (MT * 2) * CSE + lane_off;             // + mt*2*CSE + kk*256 (+128)
    const bf16_t* pB = smem + (CA + wn * (NTW * 2)) * CSE + lane_off;             // + nt*2*CSE + kk*256 (+128)
    for (int st = 0; st < nsteps; ++st) {
        const int stage = st & 1;
        if (st + 1 < nsteps) ISSUE_COLS(pix_begin + (st + 1) * BKP, stage ^ 1)
        const bf16_t* tA = pA + stage * STAGE;
        const bf16_t* tB = pB + stage * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            // inline-asm fragment reads: see conv_wgrad_rows (no compiler-inserted vmcnt(0) behind the LDS-DMA prefetch)
            s16x4_t alo[MT], ahi[MT], blo[NTW], bhi[NTW];
            const unsigned ua = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)(tA + kk * 256);
            const unsigned ub = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)(tB + kk * 256);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                             : "=&v"(alo[mt]), "=&v"(ahi[mt]) : "v"(ua), "n"(mt * 2 * CSE * 2), "n"(mt * 2 * CSE * 2 + 256));
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
                asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                             : "=&v"(blo[nt]), "=&v"(bhi[nt]) : "v"(ub), "n"(nt * 2 * CSE * 2), "n"(nt * 2 * CSE * 2 + 256));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(alo[mt]), "+v"(ahi[mt]));
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) asm volatile("" : "+v"(blo[nt]), "+v"(bhi[nt]));
            bf16x8_t fa[MT], fb[NTW];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                fa[mt] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(alo[mt], ahi[mt], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
                fb[nt] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(blo[nt], bhi[nt], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mt], fb[nt], acc[mt][nt], 0, 0, 0);
        }
        __syncthreads();
    }
#undef ISSUE_COLS

    // slab store: lane holds rows k = 4g+j, column l&15
    float* out = a.slab + (size_t)split * a.K * a.RSC;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            int col = n0 + wn * (NTW * 16) + nt * 16 + (lane & 15);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int k = k0 + wm * (MT * 16) + mt * 16 + 4 * g + j;
                if (k < a.K && col < a.RSC) out[(size_t)k * a.RSC + col] = acc[mt][nt][j];
            }
        }
}

// ---------------------------------------------------------------- row-streaming weight gradient of the 3x3 stem layers
// Conv2d_2a / 2b (32 input channels, 149^2 / 147^2 maps): as a GEMM over (r,s,c) columns the kernels above re-read every
// input pixel nine times (once per tap) and every dy pixel three times (once per 128-column tile) from L2 -- 3.9 GB of
// L2->LDS traffic for 1.06 GB of tensors, which is what bounds them.  Here a block walks down output rows
// like conv_rows3x3: every x row goes to LDS ONCE (4-slot ring, chunk-column layout, one zero pixel left of the image) and
// serves the three output rows and nine taps that use it, every dy row once (2 slots).  The taps are the same transposing
// fragment reads at shifted pixel addresses (a pixel shift is a uniform 16-byte address shift in the chunk-column layout, so
// the bank pattern of conv_wgrad_cols carries over).  A block owns 32 output channels (blockIdx selects the K half) and all
// 288 columns: 18 (tap, 16-channel) units x 2 k-tiles of accumulators spread over the four waves; it loops over
// (image, 16-row strip) units and writes ONE fp32 slab at the end (512 blocks -> the usual fixed-order wgrad_reduce).
constexpr int ST_RSEG = 16;            // output rows per work unit
// template <CT, NG>: CT = input channels / 16; NG = 64-pixel DMA groups per row image: 3 -> rows of up to 160 output pixels
// (168-pixel x image, 5 k-steps of 32 pixels: Conv2d_2a / 2b), 2 -> up to 96 output pixels (104-pixel x image, 3 k-steps:
// Conv2d_4a with CT = 5, the 56x56 resnet layers with CT = 4).  Chunk-column stride = image pixels x 16 B, = 128 mod 256
// (conflict-free transposing reads).
struct StemArgs {
    const void* x;
    const void* dy;
    float* slab;
    unsigned xbytes, dybytes;
    int N, H, W, ldx, P, Q, ldy, K, pad, nstrip, units, nb, kh;
};

template <int CT, int NG>
__global__ __launch_bounds__(256, 2) void conv_wgrad_stem(StemArgs a) {
    constexpr int ST_CS = NG == 3 ? 2688 : 1664;         // 168 / 104 pixels x 16 B
    constexpr int KS = NG == 3 ? 5 : 3;                  // k-steps of 32 pixels per output row
    constexpr int XG1 = NG == 3 ? 104 : 40;              // first pixel of the LAST DMA group of an x row image (it overlaps the
    constexpr int DG1 = NG == 3 ? 96 : 32;               // previous group instead of spilling into the next column); dy likewise
    constexpr int XC = 2 * CT;                           // chunk columns of an x row
    constexpr int ST_XROW = XC * ST_CS;
    constexpr int ST_DROW = 4 * ST_CS;                   // one dy row of the block's 32 output channels
    constexpr int NXI = (XC * NG + 3) / 4;               // x-row DMA instructions per wave
    constexpr int NDI = (4 * NG + 3) / 4;                // dy-row DMA instructions per wave
    constexpr int NU = 9 * CT;                           // (tap, 16-channel tile) units
    constexpr int UPW = (NU + 3) / 4;                    // units per wave
    __shared__ __attribute__((aligned(16))) unsigned char sX[4 * ST_XROW];
    __shared__ __attribute__((aligned(16))) unsigned char sD[2 * ST_DROW];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lin = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int kh = lin % a.kh, b = lin / a.kh;           // K half (32 output channels) and slab split of this block

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dybytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    // LDS-DMA roles: 12 wave-instructions fill a row image (4 columns x 3 pixel groups; lane = pixel); wave w issues
    // instructions w, w+4, w+8 of the x row and of the dy row.  Pixel groups start at 0, 64 and 104 (x: 168 pixels) or 96
    // (dy: 160 pixels): the last group overlaps the second instead of spilling into the next column.
    int xoff[NXI], doff[NDI];      // byte offset inside an image row (x) / output row (dy), or -1
    unsigned xdst[NXI], ddst[NDI]; // LDS byte offset inside the row image
    bool xlive[NXI], dlive[NDI];   // this wave issues instruction j (wave-uniform)
#pragma unroll
    for (int j = 0; j < NXI; ++j) {
        const int inst = wave + 4 * j;
        xlive[j] = inst < XC * NG;
        const int col = inst / NG, grp = inst - col * NG;
        const int xi = (grp == NG - 1 ? XG1 : grp * 64) + lane;             // LDS pixel of the x image: image column xi - 1
        const int wcol = xi - 1;
        xoff[j] = (wcol >= 0 && wcol < a.W) ? (wcol * a.ldx + col * 8) * 2 : -1;
        xdst[j] = (unsigned)(col * ST_CS + (xi - lane) * 16);
    }
#pragma unroll
    for (int j = 0; j < NDI; ++j) {
        const int inst = wave + 4 * j;
        dlive[j] = inst < 4 * NG;
        const int col = inst / NG, grp = inst - col * NG;
        const int qi = (grp == NG - 1 ? DG1 : grp * 64) + lane;
        doff[j] = qi < a.Q ? (qi * a.ldy + kh * 32 + col * 8) * 2 : -1;
        ddst[j] = (unsigned)(col * ST_CS + (qi - lane) * 16);
    }
#define ST_ISSUE_X(n_, h_)                                                                                     \
    {                                                                                                           \
        const int hh_ = (h_);                                                                                   \
        const bool ok_ = hh_ >= 0 && hh_ < a.H;                                                                 \
        const unsigned rb_ = (unsigned)(((n_) * a.H + hh_) * a.W * a.ldx) * 2u;                                 \
        unsigned char* dst_ = sX + ((hh_ + 8) & 3) * ST_XROW;                                                   \
        _Pragma("unroll") for (int j = 0; j < NXI; ++j) if (xlive[j]) {                                         \
            const unsigned vo = (ok_ && xoff[j] >= 0) ? rb_ + (unsigned)xoff[j] : OOB;                          \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lptr_t)(dst_ + xdst[j]), 16, vo, 0, 0, 0);           \
        }                                                                                                       \
    }
#define ST_ISSUE_D(n_, p_, pend_)                                                                              \
    {                                                                                                           \
        const int pp_ = (p_);                                                                                   \
        const bool ok_ = pp_ < (pend_);                                                                         \
        const unsigned rb_ = (unsigned)(((n_) * a.P + pp_) * a.Q * a.ldy) * 2u;                                 \
        unsigned char* dst_ = sD + (pp_ & 1) * ST_DROW;                                                         \
        _Pragma("unroll") for (int j = 0; j < NDI; ++j) if (dlive[j]) {                                         \
            const unsigned vo = (ok_ && doff[j] >= 0) ? rb_ + (unsigned)doff[j] : OOB;                          \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (lptr_t)(dst_ + ddst[j]), 16, vo, 0, 0, 0);           \
        }                                                                                                       \
    }

    // transposing fragment reads (as conv_wgrad_cols): lane (g, lq, lp) addresses pixel 4g+lq (then +16), channels 4lp..+3 of
    // a 16-channel tile = chunk column lp>>1, 8-byte half lp&1
    const int g = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;
    const unsigned lane_off = (unsigned)((lp >> 1) * ST_CS + (4 * g + lq) * 16 + (lp & 1) * 8);
    const unsigned dbase = (unsigned)(size_t)(lptr_t)sD + lane_off;
    const unsigned xbase = (unsigned)(size_t)(lptr_t)sX + lane_off;
    // this wave's units: u = wave + 4i (i < UPW), u < NU: tap = u / CT (r = tap / 3, s = tap % 3), channel tile = u % CT
    int ur[UPW], uoff[UPW];
    bool ulive[UPW];
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
        const int u = wave + 4 * i;
        ulive[i] = u < NU;
        const int tap = ulive[i] ? u / CT : 0;
        const int ct = ulive[i] ? u - tap * CT : 0;
        ur[i] = tap / 3;
        uoff[i] = ct * 2 * ST_CS + (tap - ur[i] * 3 + 1 - a.pad) * 16;            // channel tile + column shift s + 1 - pad
    }
    f32x4_t acc[UPW][2];
#pragma unroll
    for (int i = 0; i < UPW; ++i) acc[i][0] = acc[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#define ST_TR(lo, hi, addr, OFF)                                                                                \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                   \
                 : "=&v"(lo), "=&v"(hi) : "v"(addr), "n"(OFF), "n"((OFF) + 256));

    for (int unit = b; unit < a.units; unit += a.nb) {
        const int n = unit / a.nstrip;
        const int p0 = (unit - n * a.nstrip) * ST_RSEG;
        const int p1 = min(p0 + ST_RSEG, a.P);
        __syncthreads();                                 // the previous unit's last row is consumed
        ST_ISSUE_X(n, p0 - a.pad)
        ST_ISSUE_X(n, p0 - a.pad + 1)
        ST_ISSUE_X(n, p0 - a.pad + 2)
        ST_ISSUE_D(n, p0, p1)
        for (int p = p0; p < p1; ++p) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                             // x rows p-pad .. p-pad+2 and dy row p are in LDS; row p-1 is consumed
            if (p + 1 < p1) {
                ST_ISSUE_X(n, p - a.pad + 3)             // its slot held row p-pad-1
                ST_ISSUE_D(n, p + 1, p1)                 // its slot held dy row p-1
            }
            const unsigned da = dbase + (unsigned)((p & 1) * ST_DROW);
            unsigned xa[UPW];
#pragma unroll
            for (int i = 0; i < UPW; ++i) xa[i] = xbase + (unsigned)(((p - a.pad + ur[i] + 8) & 3) * ST_XROW + uoff[i]);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s16x4_t alo[2], ahi[2], blo[UPW], bhi[UPW];
                ST_TR(alo[0], ahi[0], da, ks * 512)
                ST_TR(alo[1], ahi[1], da, ks * 512 + 2 * ST_CS)
#pragma unroll
                for (int i = 0; i < UPW; ++i) ST_TR(blo[i], bhi[i], xa[i], ks * 512)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(alo[0]), "+v"(ahi[0]), "+v"(alo[1]), "+v"(ahi[1]));
#pragma unroll
                for (int i = 0; i < UPW; ++i) asm volatile("" : "+v"(blo[i]), "+v"(bhi[i]));
                const bf16x8_t fa0 = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(alo[0], ahi[0], 0, 1, 2, 3, 4, 5, 6, 7));
                const bf16x8_t fa1 = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(alo[1], ahi[1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int i = 0; i < UPW; ++i) {
                    if (!ulive[i]) continue;             // wave-uniform
                    const bf16x8_t fb = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(blo[i], bhi[i], 0, 1, 2, 3, 4, 5, 6, 7));
                    acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa0, fb, acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa1, fb, acc[i][1], 0, 0, 0);
                }
            }
        }
    }
#undef ST_TR
#undef ST_ISSUE_X
#undef ST_ISSUE_D
    // slab [nb][K][9 * C]: lane holds rows k = 4g+j of a 16 x 16 tile, column lane & 15
    constexpr int RSCW = 9 * 16 * CT;
    float* out = a.slab + (size_t)b * a.K * RSCW;
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
        if (!ulive[i]) continue;
        const int u = wave + 4 * i;
        const int col = u * 16 + (lane & 15);                  // (tap * CT + ct) * 16: the (r, s, c) column
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j) out[(size_t)(kh * 32 + kt * 16 + 4 * g + j) * RSCW + col] = acc[i][kt][j];
    }
}

// ---------------------------------------------------------------- fp32 parity-mode wgrad
// Same decomposition on v_mfma_f32_16x16x4_f32.  Tiles are [32 pixels][128 fp32 columns] (512-byte rows, one wave
// LDS-DMA instruction fills two rows); a lane's MFMA operand is ONE float -- A[k = lane>>4][m = lane&15] -- so the
// pixel-major image is read directly with ds_read_b32, no transpose needed.
constexpr int F_BKP = 32;
constexpr int F_TILE = F_BKP * TW;      // floats

template <int MT>
__global__ __launch_bounds__(NTHREADS) void conv_wgrad_f32(WgradArgs a) {
    constexpr int BMW = 32 * MT;
    __shared__ __attribute__((aligned(16))) float smem[2 * 2 * F_TILE];     // [stage][A | B]
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lin = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int split = lin / a.tiles;
    const int tile = lin - split * a.tiles;
    const int mtile = tile / a.tilesN, ntile = tile - mtile * a.tilesN;
    const int k0 = mtile * BMW, n0 = ntile * BNW;
    const int pix_begin = split * a.split_len;
    const int pix_end = min(pix_begin + a.split_len, a.M);
    const int lrow2 = lane >> 5, c32 = lane & 31;
    const bool avalid = (c32 * 4 < BMW) && (k0 + c32 * 4 < a.K);
    int br, bs, bc;
    bool bvalid;
    {
        int jcol = n0 + c32 * 4;
        bvalid = jcol < a.RSC;
        int jj = bvalid ? jcol : 0;
        int rs = jj / a.C;
        bc = jj - rs * a.C;
        br = rs / a.S;
        bs = rs - br * a.S;
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int HW = a.H * a.W;

#define ISSUE_F32(pix0, stage)                                                                                  \
    {                                                                                                           \
        float* dstA = smem + (stage) * 2 * F_TILE;                                                              \
        float* dstB = dstA + F_TILE;                                                                            \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
            const int row = (wave * 4 + j) * 2 + lrow2;                                                         \
            const int pix = (pix0) + row;                                                                       \
            const bool pv = pix < pix_end;                                                                      \
            unsigned voA = (pv && avalid) ? (unsigned)(pix * a.ldy + k0 + c32 * 4) * 4u : OOB;                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(dstA + (wave * 4 + j) * 2 * TW), 16, voA, 0, 0, 0); \
            uint32_t pp = pv ? (uint32_t)pix : 0u;                                                              \
            uint32_t n = fdiv(pp, a.fPQ);                                                                       \
            uint32_t rem = pp - n * a.fPQ.d;                                                                    \
            uint32_t p = fdiv(rem, a.fQ);                                                                       \
            uint32_t q = rem - p * a.fQ.d;                                                                      \
            int hi = (int)p * a.sh - a.ph + br, wi = (int)q * a.sw - a.pw + bs;                                 \
            bool v = pv && bvalid && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;              \
            unsigned voB = v ? (unsigned)(((int)n * HW + hi * a.W + wi) * a.ldx + bc) * 4u : OOB;               \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)(dstB + (wave * 4 + j) * 2 * TW), 16, voB, 0, 0, 0); \
        }                                                                                                       \
    }

    f32x4_t acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int nsteps = (pix_end - pix_begin + F_BKP - 1) / F_BKP;
    if (nsteps > 0) ISSUE_F32(pix_begin, 0)
    __syncthreads();
    const int g = lane >> 4, li = lane & 15;
    for (int st = 0; st < nsteps; ++st) {
        const int stage = st & 1;
        if (st + 1 < nsteps) ISSUE_F32(pix_begin + (st + 1) * F_BKP, stage ^ 1)
        const float* tA = smem + stage * 2 * F_TILE;
        const float* tB = tA + F_TILE;
#pragma unroll
        for (int k4 = 0; k4 < F_BKP / 4; ++k4) {
            const int krow = k4 * 4 + g;
            float fa[MT], fb[4];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) fa[mt] = tA[krow * TW + wm * (MT * 16) + mt * 16 + li];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) fb[nt] = tB[krow * TW + wn * 64 + nt * 16 + li];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mt], fb[nt], acc[mt][nt], 0, 0, 0);
        }
        __syncthreads();
    }
#undef ISSUE_F32
    float* out = a.slab + (size_t)split * a.K * a.RSC;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            int col = n0 + wn * 64 + nt * 16 + li;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int k = k0 + wm * (MT * 16) + mt * 16 + 4 * g + j;
                if (k < a.K && col < a.RSC) out[(size_t)k * a.RSC + col] = acc[mt][nt][j];
            }
        }
}

// dw[k][rs][cw] (+)= sum_split slab[split][k][rs*C + cw]; 64 outputs x 4 split lanes per block, the 4 lane sums are
// combined in a fixed order (bitwise reproducible)
// (k_off, Kseg): the rows [k_off, k_off+Kseg) of a horizontally fused conv go to their own destination tensor
// up to 8 destination tensors (the members of a sibling GEMM own consecutive ranges of output channels): ONE launch reduces
// the slabs into all of them -- the blocks of segment s start at blk0[s]
struct ReduceSegs {
    float* dw[8];
    int k_off[8], kseg[8], blk0[9];
    int nseg;
};
__global__ __launch_bounds__(256) void wgrad_reduce(const float* slab, ReduceSegs sg_, int nsplit, int K, int RS, int C, int Cw,
                                                    int accumulate) {
    __shared__ float part[4][64];
    int si = 0;
#pragma unroll
    for (int q = 1; q < 8; ++q)
        if (q < sg_.nseg && (int)blockIdx.x >= sg_.blk0[q]) si = q;
    float* dw = sg_.dw[si];
    const int k_off = sg_.k_off[si], Kseg = sg_.kseg[si];
    const int64_t total = (int64_t)Kseg * RS * Cw;
    const int o = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int64_t i = (int64_t)((int)blockIdx.x - sg_.blk0[si]) * 64 + o;
    float s = 0.f;
    if (i < total) {
        int cw = (int)(i % Cw);
        int64_t krs = i / Cw;
        const float* src = slab + ((int64_t)k_off * RS + krs) * C + cw;
        const int64_t stride = (int64_t)K * RS * C;
        float s0 = 0.f, s1 = 0.f;
        int sp = sg;
        for (; sp + 4 < nsplit; sp += 8) {
            s0 += src[sp * stride];
            s1 += src[(sp + 4) * stride];
        }
        if (sp < nsplit) s0 += src[sp * stride];
        s = s0 + s1;
    }
    part[sg][o] = s;
    __syncthreads();
    if (sg == 0 && i < total) {
        float r = (part[0][o] + part[1][o]) + (part[2][o] + part[3][o]);
        dw[i] = accumulate ? dw[i] + r : r;
    }
}

// the same sums (identical order per output: bitwise the same result) four outputs per thread: 16-byte loads, 256 outputs per
// block.  For tensors whose filter rows are whole float4s and unpadded (C == Cw): every layer but the 3-channel stem conv.
// The scalar kernel moved its 34 MB per launch at 2.1 TB/s and ran 68 times per step.
__global__ __launch_bounds__(256) void wgrad_reduce4(const float* slab, ReduceSegs sg_, int nsplit, int K, int RSC, int accumulate) {
    __shared__ float4 part[4][64];
    int si = 0;
#pragma unroll
    for (int q = 1; q < 8; ++q)
        if (q < sg_.nseg && (int)blockIdx.x >= sg_.blk0[q]) si = q;
    float* dw = sg_.dw[si];
    const int k_off = sg_.k_off[si], Kseg = sg_.kseg[si];
    const int64_t total4 = (int64_t)Kseg * RSC / 4;
    const int o = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int64_t i = (int64_t)((int)blockIdx.x - sg_.blk0[si]) * 64 + o;
    float4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < total4) {
        const float4* src = reinterpret_cast<const float4*>(slab + (int64_t)k_off * RSC) + i;
        const int64_t stride = (int64_t)K * RSC / 4;
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
        float4* dst = reinterpret_cast<float4*>(dw) + i;
        if (accumulate) {
            const float4 d0 = *dst;
            r.x += d0.x; r.y += d0.y; r.z += d0.z; r.w += d0.w;
        }
        *dst = r;
    }
}

int pick_mt(int K, int maxmt) {
    int best = 1;
    long bestc = -1;
    for (int mt = 1; mt <= maxmt; ++mt) {
        int bm = 32 * mt;
        long c = (long)cdiv(K, bm) * (bm + 48);
        if (bestc < 0 || c < bestc || (c == bestc && mt > best)) { bestc = c; best = mt; }
    }
    return best;
}

struct Plan { int mt, cols, stem, tilesM, tilesN, nsplit, split_len; size_t ws; int pp, flat; };

Plan make_plan(const ifcbk_conv_desc* d) {
    Plan p;
    p.pp = 0; p.flat = 0;
    const int bkp = d->dtype == IFCBK_F32 ? F_BKP : BKP;
    int64_t M = (int64_t)d->N * d->P * d->Q;
    int RSC = d->R * d->S * d->C;
    p.mt = pick_mt(d->K, 4);
    // bf16: chunk-column tiles for the narrow (K <= 64) layers, row-major tiles otherwise; IFCBK_WGRAD_COLS=0/1 forces one
    static int force = -2;
    if (force == -2) { const char* e = getenv("IFCBK_WGRAD_COLS"); force = e ? atoi(e) : -1; }
    p.cols = d->dtype == IFCBK_F32 ? 0 : (force >= 0 ? force : (p.mt <= 2));
    // row-streaming kernel for the 3x3 / stride-1 stem layers (IFCBK_WGRAD_STEM: 0 off, 1 the 32-channel layers only, 2 also
    // Conv2d_4a's 80 channels: 0.687 -> 0.616 ms, less than the 32-channel layers gain because 71-pixel rows fill only 74 % of
    // their three 32-pixel k-steps and the layer is MFMA- rather than load-bound)
    static int stem = -1;
    if (stem < 0) { const char* e = getenv("IFCBK_WGRAD_STEM"); stem = e ? atoi(e) : 2; }
    // stem: 1 = <2,3> (32 channels, rows up to 160 pixels), 2 = <5,2> (80 channels, rows up to 96 pixels: Conv2d_4a)
    p.stem = 0;
    if (stem && d->dtype == IFCBK_BF16 && d->R == 3 && d->S == 3 && d->stride_h == 1 && d->stride_w == 1 && d->Cw == d->C &&
        d->pad_h == d->pad_w && d->pad_h <= 1 && d->K % 32 == 0 && d->K <= 512) {
        if (d->C == 32 && d->Q <= 160 && d->W <= 166 && d->K <= 64) p.stem = 1;
        else if (d->C == 80 && d->Q <= 96 && d->W <= 102 && stem >= 2) p.stem = 2;
    }
    if (p.stem != 1) {
        // flat-slot kernel (conv_wgrad_flat.hip): the stride-1 multi-tap layers over 48..96 input channels (35x35 stage, Conv2d_4a)
        int ns = 0, len = 0;
        if (ifcbk_wgrad_flat_plan(d, &ns, &len)) {
            p.flat = 1; p.stem = 0; p.cols = 2;
            p.mt = (cdiv(cdiv(d->K, cdiv(d->K, 96)), 16) * 16 + 31) / 32;
            p.tilesM = cdiv(d->K, 96); p.tilesN = d->R;
            p.nsplit = ns; p.split_len = len;
            p.ws = (size_t)ns * d->K * RSC * sizeof(float);
            return p;
        }
    }
    if (p.stem) {
        const int kh = d->K / 32;
        p.mt = 0;
        p.nsplit = 512 / kh;
        const int units = d->N * cdiv(d->P, ST_RSEG);
        if (p.nsplit > units) p.nsplit = units;
        p.tilesM = p.tilesN = 1;
        p.split_len = 0;
        p.ws = (size_t)p.nsplit * d->K * RSC * sizeof(float);
        return p;
    }
    {
        // wide-tile ping-pong kernel (conv_wgrad_pp.hip): 128..192 output channels x 256 columns per block, one block per CU
        int kh = 0, ns = 0, len = 0;
        if (ifcbk_wgrad_pp_plan(d, &kh, &ns, &len)) {
            p.pp = kh; p.mt = -kh; p.cols = 0;
            p.tilesM = cdiv(d->K, 32 * kh); p.tilesN = cdiv(RSC, 256);
            p.nsplit = ns; p.split_len = len;
            p.ws = (size_t)ns * d->K * RSC * sizeof(float);
            return p;
        }
    }
    p.tilesM = cdiv(d->K, 32 * p.mt);
    p.tilesN = cdiv(RSC, BNW);
    int tiles = p.tilesM * p.tilesN;
    int64_t steps = (M + bkp - 1) / bkp;
    // split count: fill the resident block slots (2 blocks per CU; 3 for the 46 KB chunk-column MT=1 kernel) for a whole
    // number of rounds -- a few blocks over (cdiv) would cost a nearly empty extra round
    static int rounds = -1;
    if (rounds < 0) { const char* e = getenv("IFCBK_WGRAD_ROUNDS"); rounds = e ? atoi(e) : 1; }
    const int slots = 256 * ((p.cols && p.mt == 1) ? 3 : 2);
    int64_t ns = (int64_t)rounds * slots / tiles;
    int64_t maxsplit = steps / 8 > 0 ? steps / 8 : 1;
    if (ns > maxsplit) ns = maxsplit;
    if (ns < 1) ns = 1;
    int64_t len = ((steps + ns - 1) / ns) * bkp;
    ns = (M + len - 1) / len;
    p.nsplit = (int)ns;
    p.split_len = (int)len;
    p.ws = (size_t)ns * d->K * RSC * sizeof(float);
    return p;
}

template <int MT>
void launch(const WgradArgs& a, const Plan& p, hipStream_t st) {
    const dim3 grid(p.tilesM * p.tilesN * p.nsplit);
    if (p.cols) hipLaunchKernelGGL((conv_wgrad_cols<MT, 4>), grid, dim3(NTHREADS), 0, st, a);
    else hipLaunchKernelGGL(conv_wgrad_rows<MT>, grid, dim3(NTHREADS), 0, st, a);
}

}  // namespace

void ifcbk_conv_wgrad_shape(const ifcbk_conv_desc* d, int* mt, int* cols) {
    Plan p = make_plan(d);
    *mt = p.mt;
    *cols = p.cols;
}

extern "C" size_t ifcbk_conv2d_wgrad_workspace(const ifcbk_conv_desc* d) { return make_plan(d).ws; }

static int wgrad_impl(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const void* x, const void* dy, int nseg, float* const* dws,
                      const int* kseg, int accumulate, void* stream);

extern "C" int ifcbk_conv2d_wgrad(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const void* x, const void* dy, float* dw,
                                  int accumulate, void* stream) {
    if (!d) IFCBK_FAIL(ctx, IFCBK_EINVAL, "wgrad: null desc");
    int k = d->K;
    return wgrad_impl(ctx, d, x, dy, 1, &dw, &k, accumulate, stream);
}

extern "C" int ifcbk_conv2d_wgrad_segments(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const void* x, const void* dy, int nseg,
                                           float* const* dws, const int32_t* kseg, int accumulate, void* stream) {
    if (!d || nseg < 1 || nseg > 8 || !dws || !kseg) IFCBK_FAIL(ctx, IFCBK_EINVAL, "wgrad_segments: bad args");
    int sum = 0;
    for (int i = 0; i < nseg; ++i) sum += kseg[i];
    if (sum != d->K) IFCBK_FAIL(ctx, IFCBK_EINVAL, "wgrad_segments: segment sizes sum to %d, K=%d", sum, d->K);
    return wgrad_impl(ctx, d, x, dy, nseg, dws, kseg, accumulate, stream);
}

static int wgrad_impl(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const void* x, const void* dy, int nseg, float* const* dws,
                      const int* kseg, int accumulate, void* stream) {
    if (!d || (d->dtype != IFCBK_BF16 && d->dtype != IFCBK_F32)) IFCBK_FAIL(ctx, IFCBK_EUNSUPPORTED, "wgrad: dtype must be bf16 or f32");
    const int ce = dtype_chunk(d->dtype), es = dtype_esize(d->dtype);
    if (d->C % ce || d->K % ce || d->ldx % ce || d->ldy % ce) IFCBK_FAIL(ctx, IFCBK_EINVAL, "wgrad: channels must be multiples of %d", ce);
    if (d->Cw > d->C || d->Cw <= 0) IFCBK_FAIL(ctx, IFCBK_EINVAL, "wgrad: bad Cw");
    if ((int64_t)d->N * d->P * d->Q * d->ldy * es >= (1ll << 31) || (int64_t)d->N * d->H * d->W * d->ldx * es >= (1ll << 31))
        IFCBK_FAIL(ctx, IFCBK_EINVAL, "wgrad: a tensor exceeds the 2 GiB buffer-descriptor window");
    Plan p = make_plan(d);
    if (p.ws > ctx->ws_bytes)
        IFCBK_FAIL(ctx, IFCBK_ENOMEM, "wgrad: workspace %zu > reserved %zu (call ifcbk_ctx_reserve)", p.ws, ctx->ws_bytes);
    WgradArgs a;
    a.x = x; a.dy = dy; a.slab = (float*)ctx->ws;
    a.xbytes = (unsigned)((int64_t)d->N * d->H * d->W * d->ldx * es); a.dybytes = (unsigned)((int64_t)d->N * d->P * d->Q * d->ldy * es);
    a.H = d->H; a.W = d->W; a.C = d->C; a.ldx = d->ldx;
    a.K = d->K; a.R = d->R; a.S = d->S; a.P = d->P; a.Q = d->Q; a.ldy = d->ldy;
    a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w;
    a.M = d->N * d->P * d->Q; a.RSC = d->R * d->S * d->C;
    a.split_len = p.split_len; a.tilesN = p.tilesN; a.tiles = p.tilesM * p.tilesN;
    a.fPQ = make_fastdiv(d->P * d->Q); a.fQ = make_fastdiv(d->Q);
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == IFCBK_F32) {
        dim3 grid(p.tilesM * p.tilesN * p.nsplit), block(NTHREADS);
        switch (p.mt) {
            case 1: hipLaunchKernelGGL(conv_wgrad_f32<1>, grid, block, 0, st, a); break;
            case 2: hipLaunchKernelGGL(conv_wgrad_f32<2>, grid, block, 0, st, a); break;
            case 3: hipLaunchKernelGGL(conv_wgrad_f32<3>, grid, block, 0, st, a); break;
            default: hipLaunchKernelGGL(conv_wgrad_f32<4>, grid, block, 0, st, a); break;
        }
    } else if (p.flat) {
        if (int e = ifcbk_wgrad_flat_launch(ctx, d, x, dy, (float*)ctx->ws, p.nsplit, p.split_len, st)) return e;
    } else if (p.pp) {
        if (int e = ifcbk_wgrad_pp_launch(ctx, d, x, dy, (float*)ctx->ws, p.pp, p.nsplit, p.split_len, st)) return e;
    } else if (p.stem) {
        StemArgs sa;
        sa.x = x; sa.dy = dy; sa.slab = (float*)ctx->ws; sa.xbytes = a.xbytes; sa.dybytes = a.dybytes;
        sa.N = d->N; sa.H = d->H; sa.W = d->W; sa.ldx = d->ldx; sa.P = d->P; sa.Q = d->Q; sa.ldy = d->ldy; sa.K = d->K;
        sa.pad = d->pad_h; sa.nstrip = cdiv(d->P, ST_RSEG); sa.units = d->N * sa.nstrip; sa.kh = d->K / 32; sa.nb = p.nsplit;
        if (p.stem == 2) hipLaunchKernelGGL((conv_wgrad_stem<5, 2>), dim3(sa.nb * sa.kh), dim3(256), 0, st, sa);
        else hipLaunchKernelGGL((conv_wgrad_stem<2, 3>), dim3(sa.nb * sa.kh), dim3(256), 0, st, sa);
    } else {
        switch (p.mt) {
            case 1: launch<1>(a, p, st); break;
            case 2: launch<2>(a, p, st); break;
            case 3: launch<3>(a, p, st); break;
            default: launch<4>(a, p, st); break;
        }
    }
    IFCBK_LAUNCH_CHECK(ctx, "conv_wgrad_bf16");
    ReduceSegs rs;
    rs.nseg = nseg;
    const int RSCw = d->R * d->S * d->Cw;
    bool vec4 = d->C == d->Cw && RSCw % 4 == 0;
    for (int sgi = 0; sgi < nseg; ++sgi) vec4 = vec4 && ((uintptr_t)dws[sgi] % 16 == 0);
    const int per_blk = vec4 ? 256 : 64;
    int k_off = 0, blk = 0;
    for (int sgi = 0; sgi < 8; ++sgi) {
        const bool live = sgi < nseg;
        rs.dw[sgi] = live ? dws[sgi] : nullptr;
        rs.k_off[sgi] = k_off;
        rs.kseg[sgi] = live ? kseg[sgi] : 0;
        rs.blk0[sgi] = blk;
        if (live) {
            blk += cdiv((int64_t)kseg[sgi] * RSCw, per_blk);
            k_off += kseg[sgi];
        }
    }
    rs.blk0[8] = blk;
    if (vec4)
        hipLaunchKernelGGL(wgrad_reduce4, dim3(blk), dim3(256), 0, st, (const float*)ctx->ws, rs, p.nsplit, d->K, RSCw, accumulate);
    else
        hipLaunchKernelGGL(wgrad_reduce, dim3(blk), dim3(256), 0, st, (const float*)ctx->ws, rs, p.nsplit, d->K, d->R * d->S, d->C,
                           d->Cw, accumulate);
    IFCBK_LAUNCH_CHECK(ctx, "wgrad_reduce");
    return 0;
}

// ---------------------------------------------------------------- weight pack
namespace {
// w[k][rs][c] (storage type T, c<C zero padded) and wT[c][RS-1-rs][k]
template <class T>
__global__ void weight_pack_kernel(const float* wm, T* w, T* wT, int K, int RS, int C, int Cw) {
    int64_t total = (int64_t)K * RS * C;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int c = (int)(i % C);
    int64_t krs = i / C;
    int rs = (int)(krs % RS);
    int k = (int)(krs / RS);
    float v = c < Cw ? wm[krs * Cw + c] : 0.f;
    T b = from_f32<T>(v);
    w[i] = b;
    if (wT) wT[((int64_t)c * RS + (RS - 1 - rs)) * K + k] = b;
}
// One launch for every conv of the model.  A block owns a 32 (k) x 32 (c) tile of one filter tap and goes through LDS,
// so that the master read and the forward-filter write run along c and the transposed dgrad-filter write runs along k
// (a thread-per-element version wrote wT with a 2-byte store every R*S*K elements: 166 us for 24.6 M weights).
// first_block of an item counts these tiles: cdiv(K,32) * RS * cdiv(C,32).
template <class T>
__global__ __launch_bounds__(256) void weight_pack_multi_kernel(const ifcbk_pack_item* items, int n_items) {
    __shared__ T tile[32][33];
    int lo = 0, hi = n_items - 1;
    const int64_t b = blockIdx.x;
    while (lo < hi) {                                // binary search: last item whose first_block <= blockIdx.x
        int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_block <= b) lo = mid; else hi = mid - 1;
    }
    const ifcbk_pack_item it = items[lo];
    const int tilesC = (it.C + 31) >> 5;
    int lb = (int)(b - it.first_block);
    const int ct = lb % tilesC;
    lb /= tilesC;
    const int rs = lb % it.RS;
    const int kt = lb / it.RS;
    if (kt * 32 >= it.K) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = kt * 32 + ty + 8 * j, c = ct * 32 + tx;
        T q = from_f32<T>(0.f);
        if (k < it.K && c < it.C) {
            const int64_t krs = (int64_t)k * it.RS + rs;
            q = from_f32<T>(c < it.Cw ? it.w_master[krs * it.Cw + c] : 0.f);
            ((T*)it.w)[krs * it.C + c] = q;
        }
        tile[ty + 8 * j][tx] = q;
    }
    __syncthreads();
    if (!it.wT) return;
    const int ldT = it.wT_ld > 0 ? it.wT_ld : it.K;      // horizontally fused convs share one [C][RS][Ktot] dgrad filter
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = ct * 32 + ty + 8 * j, k = kt * 32 + tx;
        if (c < it.C && k < it.K) ((T*)it.wT)[((int64_t)c * it.RS + (it.RS - 1 - rs)) * ldT + k] = tile[tx][ty + 8 * j];
    }
}
}  // namespace

extern "C" int ifcbk_weight_pack_multi(ifcbk_ctx* ctx, const ifcbk_pack_item* items_dev, int n_items, int64_t total_blocks,
                                       int dtype, void* stream) {
    if (!items_dev || n_items <= 0 || total_blocks <= 0 || total_blocks >= (1ll << 31)) IFCBK_FAIL(ctx, IFCBK_EINVAL, "weight_pack_multi: bad args");
    if (dtype == IFCBK_F32)
        hipLaunchKernelGGL(weight_pack_multi_kernel<float>, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, items_dev, n_items);
    else if (dtype == IFCBK_BF16)
        hipLaunchKernelGGL(weight_pack_multi_kernel<bf16_t>, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, items_dev, n_items);
    else IFCBK_FAIL(ctx, IFCBK_EINVAL, "weight_pack_multi: bad dtype");
    IFCBK_LAUNCH_CHECK(ctx, "weight_pack_multi");
    return 0;
}

extern "C" int ifcbk_weight_pack(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const float* w_master, void* w, void* wT,
                                 void* stream) {
    if (!d || (d->dtype != IFCBK_BF16 && d->dtype != IFCBK_F32)) IFCBK_FAIL(ctx, IFCBK_EUNSUPPORTED, "weight_pack: bad dtype");
    int64_t total = (int64_t)d->K * d->R * d->S * d->C;
    if (d->dtype == IFCBK_F32)
        hipLaunchKernelGGL(weight_pack_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w_master,
                           (float*)w, (float*)wT, d->K, d->R * d->S, d->C, d->Cw);
    else
        hipLaunchKernelGGL(weight_pack_kernel<bf16_t>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w_master,
                           (bf16_t*)w, (bf16_t*)wT, d->K, d->R * d->S, d->C, d->Cw);
    IFCBK_LAUNCH_CHECK(ctx, "weight_pack");
    return 0;
}
