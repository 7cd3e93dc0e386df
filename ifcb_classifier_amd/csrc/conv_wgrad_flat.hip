// Convolution weight gradient over FLAT SLOTS (bf16, gfx950), round 5: the stride-1 multi-tap layers with at most 96 output
// channels per tile -- the 35x35 stage's 3x3 / 5x5 layers and Conv2d_4a -- whose weight gradients ran at 16-20 % MFMA-busy.
//
//   dW[k][(r,s,c)] = sum_pix dy[pix][k] * x[pix + (r,s)][c]
//
// conv_wgrad_rows / _cols (conv_wgrad.hip) gather the x operand once per TAP: a 64-pixel step brings 64 dy rows and, for a
// 128-column tile, 64 gathered x rows per tap segment -- 32 LDS-DMA instructions for 24 MFMAs per wave (96 x 128 tile).  The global->LDS
// path takes ~50 cycles per instruction whatever it carries, so those kernels are bound by it at a quarter of the
// matrix pipe.  Here the reduction runs over the FLAT SLOTS of conv_flat.hip / conv_slab.hip (image n, row h, column w at slot
// (n*Hp + h + ph)*Wp + w + pw: one shared band of invalid slots between rows and images serves as the padding of both
// neighbours; an output pixel (p, q) is slot (n*Hp + p)*Wp + q and tap (r, s) of ANY output slot reads slot + r*Wp + s), and a
// block owns ONE FILTER ROW r: all S taps x C channels of it (192 / 288 / 240 columns) x up to 96 output channels.  A step of 64
// slots then needs the 64 dy rows and ONE x slab of 64 + S - 1 slots -- the S taps are the same slab read at row offsets 0..S-1
// (transposing fragment reads at per-lane constant addresses: the slab of every step lands at the same LDS rows) -- 36
// instructions for 2 x 27 MFMAs per wave (96 x 288): 2.25 x the arithmetic per instruction.  dy at a band slot is requested
// through an out-of-range offset and reads zero, so band slots add nothing (5.6 % of the slots of a 35x35 map).
//
// Structure: conv_wgrad_rows' (256 threads, two blocks per CU that cover each other's loads, two stages, one barrier per step,
// lane = pixel decode once per step into a per-wave LDS table, inline-asm ds_read_b64_tr_b16 at immediate offsets), fp32 split-K
// slabs [split][K][R*S*C] in the ctx workspace summed in a fixed order by the existing reduce kernels.  The accumulator tile is
// 2.25 x that of conv_wgrad_rows<3>, and slab bytes = blocks x accumulator tile: a single launch writes 56 MB of
// slabs per 96 -> 96 layer.  The members of one Inception block therefore run as ONE grid (ifcbk_conv2d_wgrad_group: 11 filter
// rows x 46 splits instead of 3 launches of 170 / 170 / 102 splits: a third of the slab bytes, one reduce launch).
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace {

struct WfArgs {
    const void* x;
    const void* dy;
    float* slab;          // [nsplit][K][RSC]
    unsigned xbytes, dybytes;
    int H, W, C, ldx;
    int K, R, S;
    int P, Q, ldy;
    int ph, pw;
    int N, Hp, Wp, HpWp, G;       // flat geometry: G = N * Hp * Wp slots
    int RSC;
    int split_len;        // slots per split (multiple of 64)
    int tilesK, tiles;    // K tiles of <= 96 channels; tiles = R * tilesK
    int kt;               // output channels per K tile (multiple of 16, <= 96)
    int ncolt, c16;       // 16-column tiles of a filter row (S*C/16), of one tap (C/16)
    fastdiv_t fHW, fW;
};

constexpr int WF_THREADS = 256;
constexpr int WF_TW = 128;                 // LDS row: 128 elements = 256 B (both operands)
constexpr int WF_AROWS = 64, WF_BROWS = 80;
constexpr int WF_STAGE = (WF_AROWS + WF_BROWS) * WF_TW;      // elements per stage
constexpr int WF_MT = 3, WF_NTW = 9;       // per wave: up to 3 x 9 accumulator tiles (wave = K half x column half)

typedef __attribute__((address_space(3))) void* lptr_t;

#define WF_TR(lo, hi, addr, OFF)                                                                                    \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                       \
                 : "=&v"(lo), "=&v"(hi)                                                                             \
                 : "v"(addr), "n"(OFF), "n"((OFF) + 16 * WF_TW * 2));

__device__ __forceinline__ void wf_run(const WfArgs& a, const int lin, bf16_t* smem, int (*ptab)[36]) {
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int split = lin / a.tiles;
    const int tile = lin - split * a.tiles;
    const int fr = tile / a.tilesK, ktile = tile - fr * a.tilesK;        // filter row, K tile
    const int k0 = ktile * a.kt;
    const int kmt = (a.kt + 31) / 32;                     // accumulator tiles of 16 channels per wave row (K half)
    const int ntw = (a.ncolt + 1) / 2;                    // ... of 16 columns per wave column (column half)
    const int g_begin = split * a.split_len;
    const int g_end = min(g_begin + a.split_len, a.G);
    const int nsteps = (g_end - g_begin + 63) / 64;

    // LDS-DMA roles (conv_wgrad_rows): one wave-instruction = 4 rows x 256 B; lane -> (row l>>4 of the group, phys chunk l&15);
    // 16-byte chunk c of row r lives at physical chunk c ^ ((r & 7) << 1).  dy: rows (wave*4 + j)*4 + lrow4, j = 0..3; slab: rows
    // (wave*5 + j)*4 + lrow4, j = 0..4.
    const int lrow4 = lane >> 4, phys = lane & 15;
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    unsigned acol[2], bcol[2];                            // byte offset of this lane's chunk inside a dy / x pixel row, per row parity
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int c16 = phys ^ (((par * 4 + lrow4) & 7) << 1);
        acol[par] = (c16 * 8 < a.kt && k0 + c16 * 8 < a.K) ? (unsigned)(k0 + c16 * 8) * 2u : OOB;
        bcol[par] = (c16 * 8 < a.C) ? (unsigned)(c16 * 8) * 2u : OOB;
    }
    // decode of this wave's rows of a step, one row per lane: lanes 0-15 its 16 dy rows, lanes 16-35 its 20 slab rows
    int* const mytab = &ptab[wave][lane < 36 ? lane : 0];
    const int* const rdA = &ptab[wave][lrow4];            // + j*4
    const int* const rdB = &ptab[wave][16 + lrow4];       // + j*4

#define WF_ISSUE(g0, stage)                                                                                         \
    {                                                                                                               \
        if (lane < 36) {                                                                                            \
            const bool isA = lane < 16;                                                                             \
            const int row = isA ? wave * 16 + lane : wave * 20 + (lane - 16);                                       \
            const unsigned F = (unsigned)((g0) + row + (isA ? 0 : fr * a.Wp));                                      \
            const unsigned n = fdiv(F, a.fHW);                                                                      \
            const unsigned rem = F - n * (unsigned)a.HpWp;                                                          \
            const unsigned line = fdiv(rem, a.fW);                                                                  \
            const unsigned col = rem - line * (unsigned)a.Wp;                                                       \
            int off;                                                                                                \
            if (isA) {                                                                                              \
                const bool v = (int)F < g_end && (int)n < a.N && (int)line < a.P && (int)col < a.Q;                 \
                off = v ? (int)(((n * (unsigned)a.P + line) * (unsigned)a.Q + col) * (unsigned)a.ldy * 2u) : (int)OOB; \
            } else {                                                                                                \
                const int h = (int)line - a.ph, w = (int)col - a.pw;                                                \
                const bool v = (int)n < a.N && h >= 0 && w >= 0;                                                    \
                off = v ? (int)(((n * (unsigned)a.H + (unsigned)h) * (unsigned)a.W + (unsigned)w) * (unsigned)a.ldx * 2u) : (int)OOB; \
            }                                                                                                       \
            *mytab = off;       /* same-wave LDS traffic is ordered: no barrier between this store and the reads below */ \
        }                                                                                                           \
        bf16_t* dstA = smem + (stage) * WF_STAGE;                                                                   \
        bf16_t* dstB = dstA + WF_AROWS * WF_TW;                                                                     \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                             \
            const unsigned e = (unsigned)rdA[j * 4];                                                                \
            const unsigned ac = acol[j & 1];                                                                        \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(dstA + (wave * 4 + j) * 4 * WF_TW), 16,          \
                                                     (e != OOB && ac != OOB) ? e + ac : OOB, 0, 0, 0);              \
        }                                                                                                           \
        _Pragma("unroll") for (int j = 0; j < 5; ++j) {                                                             \
            const unsigned e = (unsigned)rdB[j * 4];                                                                \
            const unsigned bc = ((wave + j) & 1) ? bcol[1] : bcol[0];                                               \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)(dstB + (wave * 5 + j) * 4 * WF_TW), 16,          \
                                                     (e != OOB && bc != OOB) ? e + bc : OOB, 0, 0, 0);              \
        }                                                                                                           \
    }

    f32x4_t acc[WF_MT][WF_NTW];
#pragma unroll
    for (int i = 0; i < WF_MT; ++i)
#pragma unroll
        for (int j = 0; j < WF_NTW; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (nsteps > 0) WF_ISSUE(g_begin, 0)
    __syncthreads();

    // transposing fragment reads (conv_wgrad_rows): lane (g, lq, lp) addresses LDS row 4g+lq (+32 per k half, +16 for the upper
    // registers), columns col0 + 4lp .. +3.  dy: row = reduction index; x: row = reduction index + tap s of the column tile --
    // a per-lane CONSTANT, because every step's slab starts at LDS row 0 of its stage.
    const int g = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;
    const int trow = 4 * g + lq;
    unsigned fa0[WF_MT], fb0[WF_NTW];
#pragma unroll
    for (int mt = 0; mt < WF_MT; ++mt) {
        const int col = (wm * kmt + (mt < kmt ? mt : 0)) * 16 + 4 * lp;
        fa0[mt] = (unsigned)(size_t)(lptr_t)(smem + trow * WF_TW + (((col >> 3) ^ ((trow & 7) << 1)) << 3) + (col & 7));
    }
#pragma unroll
    for (int nt = 0; nt < WF_NTW; ++nt) {
        int ct = wn * ntw + nt;
        if (nt >= ntw || ct >= a.ncolt) ct = 0;           // (a tile that does not exist: any valid address, never multiplied)
        const int s = ct / a.c16;
        const int col = (ct - s * a.c16) * 16 + 4 * lp;
        const int row = trow + s;
        fb0[nt] = (unsigned)(size_t)(lptr_t)(smem + WF_AROWS * WF_TW + row * WF_TW + (((col >> 3) ^ ((row & 7) << 1)) << 3) + (col & 7));
    }
    int ntv = a.ncolt - wn * ntw;                         // column tiles this wave really has
    ntv = ntv < 0 ? 0 : (ntv > ntw ? ntw : ntv);

#define WF_MATH(stage)                                                                                              \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                              \
        s16x4_t alo[WF_MT], ahi[WF_MT], blo[WF_NTW], bhi[WF_NTW];                                                   \
        _Pragma("unroll") for (int mt = 0; mt < WF_MT; ++mt) {                                                      \
            if (kk == 0) { WF_TR(alo[mt], ahi[mt], fa0[mt], (stage) * WF_STAGE * 2) }                               \
            else { WF_TR(alo[mt], ahi[mt], fa0[mt], (stage) * WF_STAGE * 2 + 32 * WF_TW * 2) }                      \
        }                                                                                                           \
        _Pragma("unroll") for (int nt = 0; nt < WF_NTW; ++nt) {                                                     \
            if (kk == 0) { WF_TR(blo[nt], bhi[nt], fb0[nt], (stage) * WF_STAGE * 2) }                               \
            else { WF_TR(blo[nt], bhi[nt], fb0[nt], (stage) * WF_STAGE * 2 + 32 * WF_TW * 2) }                      \
        }                                                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                          \
        _Pragma("unroll") for (int mt = 0; mt < WF_MT; ++mt) asm volatile("" : "+v"(alo[mt]), "+v"(ahi[mt]));       \
        _Pragma("unroll") for (int nt = 0; nt < WF_NTW; ++nt) asm volatile("" : "+v"(blo[nt]), "+v"(bhi[nt]));      \
        bf16x8_t fa[WF_MT], fb[WF_NTW];                                                                             \
        _Pragma("unroll") for (int mt = 0; mt < WF_MT; ++mt)                                                        \
            fa[mt] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(alo[mt], ahi[mt], 0, 1, 2, 3, 4, 5, 6, 7)); \
        _Pragma("unroll") for (int nt = 0; nt < WF_NTW; ++nt)                                                       \
            fb[nt] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(blo[nt], bhi[nt], 0, 1, 2, 3, 4, 5, 6, 7)); \
        _Pragma("unroll") for (int nt = 0; nt < WF_NTW; ++nt)                                                       \
            if (nt < ntv) {                                                                                         \
                _Pragma("unroll") for (int mt = 0; mt < WF_MT; ++mt)                                                \
                    if (mt < kmt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mt], fb[nt], acc[mt][nt], 0, 0, 0); \
            }                                                                                                       \
    }

    // __syncthreads() = s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier: the stage issued at the top of a step has landed (in every wave)
    // before the next step reads it, and nobody still reads the stage the next issue overwrites
    for (int st = 0; st < nsteps; st += 2) {
        if (st + 1 < nsteps) WF_ISSUE(g_begin + (st + 1) * 64, 1)
        WF_MATH(0)
        __syncthreads();
        if (st + 1 >= nsteps) break;
        if (st + 2 < nsteps) WF_ISSUE(g_begin + (st + 2) * 64, 0)
        WF_MATH(1)
        __syncthreads();
    }
#undef WF_ISSUE
#undef WF_MATH

    // slab store: lane holds rows k = 4g+j of its tile, column l&15
    float* out = a.slab + (size_t)split * a.K * a.RSC;
    const int j0 = fr * a.S * a.C;
#pragma unroll
    for (int mt = 0; mt < WF_MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < WF_NTW; ++nt) {
            if (mt >= kmt || nt >= ntv) continue;
            const int col = j0 + (wn * ntw + nt) * 16 + (lane & 15);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kl = (wm * kmt + mt) * 16 + 4 * g + j;
                const int k = k0 + kl;
                if (kl < a.kt && k < a.K) out[(size_t)k * a.RSC + col] = acc[mt][nt][j];
            }
        }
}

__global__ __launch_bounds__(WF_THREADS, 2) void conv_wgrad_flat(WfArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * WF_STAGE];
    __shared__ int ptab[4][36];
    wf_run(a, (int)xcd_remap(blockIdx.x, gridDim.x), smem, ptab);
}

constexpr int WF_MAXG = 8;
struct WfGroup {
    WfArgs a[WF_MAXG];
    int blk0[WF_MAXG + 1];
    int n;
};
__global__ __launch_bounds__(WF_THREADS, 2) void conv_wgrad_flatg(WfGroup g) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * WF_STAGE];
    __shared__ int ptab[4][36];
    const int bid = (int)xcd_remap(blockIdx.x, gridDim.x);
    int gi = 0;
#pragma unroll
    for (int q = 1; q < WF_MAXG; ++q)
        if (q < g.n && bid >= g.blk0[q]) gi = q;
    const WfArgs a = g.a[gi];
    wf_run(a, bid - g.blk0[gi], smem, ptab);
}

// IFCBK_WGRAD_FLAT: 0 = never, 1 = the layers it was measured on (default), 2 = wherever the kernel applies (tests)
int wf_mode() {
    const char* e = getenv("IFCBK_WGRAD_FLAT");
    return e ? atoi(e) : 1;
}

// geometry checks shared by the single and the grouped launch; fills the shape part of WfArgs
bool wf_shape(const ifcbk_conv_desc* d, WfArgs* a) {
    if (d->dtype != IFCBK_BF16 || d->Cw != d->C) return false;
    if (d->stride_h != 1 || d->stride_w != 1) return false;
    if (d->pad_h < 0 || d->pad_w < 0 || d->pad_h > d->R - 1 || d->pad_w > d->S - 1) return false;
    if (d->P != d->H + 2 * d->pad_h - d->R + 1 || d->Q != d->W + 2 * d->pad_w - d->S + 1 || d->P < 1 || d->Q < 1) return false;
    if (d->C % 16 || d->C > 128 || d->K % 8 || d->ldx % 8 || d->ldy % 8) return false;
    if (d->S > 16 || (d->R * d->S * d->C) % 4) return false;
    const int ncolt = d->S * d->C / 16;
    if (ncolt > 2 * WF_NTW) return false;
    const int tilesK = cdiv(d->K, 96);
    int kt = cdiv(cdiv(d->K, tilesK), 16) * 16;           // channels per K tile: even split, whole 16-channel tiles
    if (kt > 96) return false;
    const int Hp = d->H + d->pad_h, Wp = d->W + d->pad_w;
    const int64_t G = (int64_t)d->N * Hp * Wp;
    if (G + (int64_t)(d->R + 2) * Wp + 4096 >= (1ll << 31)) return false;
    if ((int64_t)d->N * d->P * d->Q * d->ldy * 2 >= (1ll << 31) || (int64_t)d->N * d->H * d->W * d->ldx * 2 >= (1ll << 31)) return false;
    if (a) {
        a->H = d->H; a->W = d->W; a->C = d->C; a->ldx = d->ldx;
        a->K = d->K; a->R = d->R; a->S = d->S; a->P = d->P; a->Q = d->Q; a->ldy = d->ldy;
        a->ph = d->pad_h; a->pw = d->pad_w;
        a->N = d->N; a->Hp = Hp; a->Wp = Wp; a->HpWp = Hp * Wp; a->G = (int)G;
        a->RSC = d->R * d->S * d->C;
        a->tilesK = tilesK; a->tiles = d->R * tilesK; a->kt = kt;
        a->ncolt = ncolt; a->c16 = d->C / 16;
        a->fHW = make_fastdiv((uint32_t)(Hp * Wp)); a->fW = make_fastdiv((uint32_t)Wp);
        a->xbytes = (unsigned)((int64_t)d->N * d->H * d->W * d->ldx * 2);
        a->dybytes = (unsigned)((int64_t)d->N * d->P * d->Q * d->ldy * 2);
    }
    return true;
}

// the measured niche (mode 1): filters of at least 3 taps per row over 48..96 input channels -- the 35x35 stage's 3x3 / 5x5 layers
// and Conv2d_4a; 1x1 layers (S = 1: nothing to reuse) and the 32-channel stem layers (their row kernel is leaner) stay where they are
bool wf_wanted(const ifcbk_conv_desc* d) {
    const int mode = wf_mode();
    if (mode <= 0 || !wf_shape(d, nullptr)) return false;
    if (mode >= 2) return true;
    return d->S >= 3 && d->R >= 3 && d->C >= 48 && d->C <= 96 && (int64_t)d->N * d->P * d->Q >= 100000;
}

}  // namespace

// single launch: pixel splits that fill the chip's 2 x 256 block slots once
bool ifcbk_wgrad_flat_plan(const ifcbk_conv_desc* d, int* nsplit_out, int* split_len_out) {
    WfArgs a;
    if (!wf_wanted(d) || !wf_shape(d, &a)) return false;
    const int64_t steps = ((int64_t)a.G + 63) / 64;
    int64_t ns = (2 * ifcbk_num_cus()) / a.tiles;
    const int64_t maxsplit = steps / 8 > 0 ? steps / 8 : 1;
    if (ns > maxsplit) ns = maxsplit;
    if (ns < 1) ns = 1;
    const int64_t len = ((steps + ns - 1) / ns) * 64;
    ns = ((int64_t)a.G + len - 1) / len;
    *nsplit_out = (int)ns;
    *split_len_out = (int)len;
    return true;
}

int ifcbk_wgrad_flat_tiles(const ifcbk_conv_desc* d) {
    WfArgs a;
    return wf_shape(d, &a) ? a.tiles : 0;
}

int ifcbk_wgrad_flat_launch(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const void* x, const void* dy, float* slab, int nsplit, int split_len,
                            hipStream_t st) {
    WfArgs a;
    if (!wf_shape(d, &a)) IFCBK_FAIL(ctx, IFCBK_EINVAL, "wgrad_flat: shape not served");
    a.x = x; a.dy = dy; a.slab = slab; a.split_len = split_len;
    hipLaunchKernelGGL(conv_wgrad_flat, dim3((unsigned)(a.tiles * nsplit)), dim3(WF_THREADS), 0, st, a);
    IFCBK_LAUNCH_CHECK(ctx, "conv_wgrad_flat");
    return 0;
}

// ---- grouped launch (called by ifcbk_conv2d_wgrad_group when every member is a flat-slot layer)
bool ifcbk_wgrad_flat_member(const ifcbk_conv_desc* d) { return wf_wanted(d); }

// one grid for all members: the fewest steps per block L such that sum_i tiles_i * ceil(steps_i / L) fits the 2 x CUs block slots
bool ifcbk_wgrad_flat_group_plan(int n, const ifcbk_conv_desc* ds, int* nsplit, int* split_len, int* tiles, size_t* slab_off, size_t* ws,
                                 int* blocks_out) {
    if (n < 1 || n > WF_MAXG) return false;
    int64_t steps[WF_MAXG];
    WfArgs a;
    for (int i = 0; i < n; ++i) {
        if (!wf_wanted(&ds[i]) || !wf_shape(&ds[i], &a)) return false;
        tiles[i] = a.tiles;
        steps[i] = ((int64_t)a.G + 63) / 64;
    }
    const int slots = 2 * ifcbk_num_cus();
    int64_t lo = 1, hi = 1;
    for (int i = 0; i < n; ++i) hi = steps[i] > hi ? steps[i] : hi;
    auto blocks_at = [&](int64_t L) { int64_t b = 0; for (int i = 0; i < n; ++i) b += (int64_t)tiles[i] * ((steps[i] + L - 1) / L); return b; };
    if (blocks_at(hi) > slots && wf_mode() < 2) return false;
    while (lo < hi) {
        const int64_t mid = (lo + hi) / 2;
        if (blocks_at(mid) <= slots) hi = mid; else lo = mid + 1;
    }
    const int64_t L = lo < 8 ? 8 : lo;
    size_t off = 0;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        wf_shape(&ds[i], &a);
        int64_t ns = (steps[i] + L - 1) / L;
        const int64_t len = ((steps[i] + ns - 1) / ns) * 64;
        ns = ((int64_t)a.G + len - 1) / len;
        nsplit[i] = (int)ns;
        split_len[i] = (int)len;
        slab_off[i] = off;
        off += (size_t)ns * ds[i].K * a.RSC * sizeof(float);
        off = (off + 255) & ~(size_t)255;
        blocks += tiles[i] * (int)ns;
    }
    *ws = off;
    *blocks_out = blocks;
    return true;
}

int ifcbk_wgrad_flat_group_launch(ifcbk_ctx* ctx, int n, const ifcbk_conv_desc* ds, const void* const* xs, const void* const* dys,
                                  const int* nsplit, const int* split_len, const size_t* slab_off, hipStream_t st) {
    WfGroup g;
    memset(&g, 0, sizeof(g));
    int blk = 0;
    for (int i = 0; i < n; ++i) {
        if (!wf_shape(&ds[i], &g.a[i])) IFCBK_FAIL(ctx, IFCBK_EINVAL, "wgrad_flat group: member %d not served", i);
        g.a[i].x = xs[i]; g.a[i].dy = dys[i]; g.a[i].slab = (float*)((char*)ctx->ws + slab_off[i]); g.a[i].split_len = split_len[i];
        g.blk0[i] = blk;
        blk += g.a[i].tiles * nsplit[i];
    }
    for (int i = n; i <= WF_MAXG; ++i) g.blk0[i] = blk;
    g.n = n;
    hipLaunchKernelGGL(conv_wgrad_flatg, dim3((unsigned)blk), dim3(WF_THREADS), 0, st, g);
    IFCBK_LAUNCH_CHECK(ctx, "conv_wgrad_flatg");
    return 0;
}
