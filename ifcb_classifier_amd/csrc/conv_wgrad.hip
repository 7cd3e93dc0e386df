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

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ s16x4_t tr_read(const bf16_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
}
// Both tiles are pixel-major (the reduction index is the ROW), 256-byte rows, filled by LDS-DMA.  16-byte
// chunk c of row r lives at physical chunk c ^ ((r & 7) << 1): the 8 rows one half-wave touches in a
// ds_read_b64_tr_b16 then fall into 8 different 32-byte bank slots (conflict-free transposing reads).
__device__ __forceinline__ const bf16_t* tr_addr(const bf16_t* tile, int row, int col) {
    int c16 = col >> 3;
    return tile + row * TW + ((c16 ^ ((row & 7) << 1)) << 3) + (col & 7);
}

template <int MT>
__global__ __launch_bounds__(NTHREADS) void conv_wgrad_bf16(WgradArgs a) {
    constexpr int BMW = 32 * MT;
    constexpr int CA = BMW / 8;                      // valid dy chunks per pixel row
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * 2 * TILE];     // [stage][A | B]

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
    // logical chunk per instruction parity (row & 7 = (j&1)*4 + lrow4)
    int ac16[2], bcol_r[2], bcol_s[2], bcol_c[2];
    bool avalid[2], bvalid[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        int c16 = phys ^ (((par * 4 + lrow4) & 7) << 1);
        ac16[par] = c16;
        avalid[par] = (c16 < CA) && (k0 + c16 * 8 < a.K);
        int jcol = n0 + c16 * 8;
        bvalid[par] = jcol < a.RSC;
        int jj = bvalid[par] ? jcol : 0;
        int rs = jj / a.C;
        bcol_c[par] = jj - rs * a.C;
        bcol_r[par] = rs / a.S;
        bcol_s[par] = rs - bcol_r[par] * a.S;
    }

    // buffer_load ... lds through SRDs: an out-of-range offset (padding, tail pixels, tail channels) reads zeros
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int HW = a.H * a.W;

#define ISSUE_TILE(pix0, stage)                                                                                 \
    {                                                                                                           \
        bf16_t* dstA = smem + (stage) * 2 * TILE;                                                               \
        bf16_t* dstB = dstA + TILE;                                                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
            const int par = j & 1;                                                                              \
            const int row = (wave * 4 + j) * 4 + lrow4;                                                         \
            const int pix = (pix0) + row;                                                                       \
            const bool pv = pix < pix_end;                                                                      \
            unsigned voA = (pv && avalid[par]) ? (unsigned)(pix * a.ldy + k0 + ac16[par] * 8) * 2u : OOB;       \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(dstA + (wave * 4 + j) * 4 * TW), 16, voA, 0, 0, 0); \
            uint32_t pp = pv ? (uint32_t)pix : 0u;                                                              \
            uint32_t n = fdiv(pp, a.fPQ);                                                                       \
            uint32_t rem = pp - n * a.fPQ.d;                                                                    \
            uint32_t p = fdiv(rem, a.fQ);                                                                       \
            uint32_t q = rem - p * a.fQ.d;                                                                      \
            int hi = (int)p * a.sh - a.ph + bcol_r[par], wi = (int)q * a.sw - a.pw + bcol_s[par];               \
            bool v = pv && bvalid[par] && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;         \
            unsigned voB = v ? (unsigned)(((int)n * HW + hi * a.W + wi) * a.ldx + bcol_c[par]) * 2u : OOB;      \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)(dstB + (wave * 4 + j) * 4 * TW), 16, voB, 0, 0, 0); \
        }                                                                                                       \
    }

    f32x4_t acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nsteps = (pix_end - pix_begin + BKP - 1) / BKP;
    if (nsteps > 0) ISSUE_TILE(pix_begin, 0)
    __syncthreads();

    // transposing fragment reads: lane (g, q, p) addresses LDS row 4g+q (then +16), columns col0+4p..+3
    const int g = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;
    const int trow = 4 * g + lq;
    for (int st = 0; st < nsteps; ++st) {
        const int stage = st & 1;
        if (st + 1 < nsteps) ISSUE_TILE(pix_begin + (st + 1) * BKP, stage ^ 1)
        const bf16_t* tA = smem + stage * 2 * TILE;
        const bf16_t* tB = tA + TILE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t fa[MT], fb[4];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                int col = wm * (MT * 16) + mt * 16 + 4 * lp;
                s16x4_t lo = tr_read(tr_addr(tA, kk * 32 + trow, col)), hi = tr_read(tr_addr(tA, kk * 32 + 16 + trow, col));
                fa[mt] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                int col = wn * 64 + nt * 16 + 4 * lp;
                s16x4_t lo = tr_read(tr_addr(tB, kk * 32 + trow, col)), hi = tr_read(tr_addr(tB, kk * 32 + 16 + trow, col));
                fb[nt] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mt], fb[nt], acc[mt][nt], 0, 0, 0);
        }
        __syncthreads();
    }
#undef ISSUE_TILE

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
__global__ __launch_bounds__(256) void wgrad_reduce(const float* slab, float* dw, int nsplit, int K, int RS, int C, int Cw,
                                                    int accumulate, int k_off, int Kseg) {
    __shared__ float part[4][64];
    const int64_t total = (int64_t)Kseg * RS * Cw;
    const int o = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + o;
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

int pick_mt(int K) {
    int best = 1;
    long bestc = -1;
    for (int mt = 1; mt <= 4; ++mt) {
        int bm = 32 * mt;
        long c = (long)cdiv(K, bm) * (bm + 48);
        if (bestc < 0 || c < bestc || (c == bestc && mt > best)) { bestc = c; best = mt; }
    }
    return best;
}

struct Plan { int mt, tilesM, tilesN, nsplit, split_len; size_t ws; };

Plan make_plan(const ifcbk_conv_desc* d) {
    Plan p;
    const int bkp = d->dtype == IFCBK_F32 ? F_BKP : BKP;
    int64_t M = (int64_t)d->N * d->P * d->Q;
    int RSC = d->R * d->S * d->C;
    p.mt = pick_mt(d->K);
    p.tilesM = cdiv(d->K, 32 * p.mt);
    p.tilesN = cdiv(RSC, BNW);
    int tiles = p.tilesM * p.tilesN;
    int64_t steps = (M + bkp - 1) / bkp;
    int64_t ns = cdiv(1024, tiles);
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
    hipLaunchKernelGGL(conv_wgrad_bf16<MT>, dim3(p.tilesM * p.tilesN * p.nsplit), dim3(NTHREADS), 0, st, a);
}

}  // namespace

int ifcbk_conv_wgrad_mt(int K) { return pick_mt(K); }

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
    } else {
        switch (p.mt) {
            case 1: launch<1>(a, p, st); break;
            case 2: launch<2>(a, p, st); break;
            case 3: launch<3>(a, p, st); break;
            default: launch<4>(a, p, st); break;
        }
    }
    IFCBK_LAUNCH_CHECK(ctx, "conv_wgrad_bf16");
    int k_off = 0;
    for (int sgi = 0; sgi < nseg; ++sgi) {
        int64_t total = (int64_t)kseg[sgi] * d->R * d->S * d->Cw;
        hipLaunchKernelGGL(wgrad_reduce, dim3(cdiv(total, 64)), dim3(256), 0, st, (const float*)ctx->ws, dws[sgi], p.nsplit,
                           d->K, d->R * d->S, d->C, d->Cw, accumulate, k_off, kseg[sgi]);
        IFCBK_LAUNCH_CHECK(ctx, "wgrad_reduce");
        k_off += kseg[sgi];
    }
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
template <class T>
__global__ __launch_bounds__(256) void weight_pack_multi_kernel(const ifcbk_pack_item* items, int n_items) {
    // binary search: last item whose first_block <= blockIdx.x
    int lo = 0, hi = n_items - 1;
    const int64_t b = blockIdx.x;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_block <= b) lo = mid; else hi = mid - 1;
    }
    const ifcbk_pack_item it = items[lo];
    const int64_t total = (int64_t)it.K * it.RS * it.C;
    const int64_t i = (b - it.first_block) * 256 + threadIdx.x;
    if (i >= total) return;
    int c = (int)(i % it.C);
    int64_t krs = i / it.C;
    int rs = (int)(krs % it.RS);
    int k = (int)(krs / it.RS);
    float v = c < it.Cw ? it.w_master[krs * it.Cw + c] : 0.f;
    T q = from_f32<T>(v);
    ((T*)it.w)[i] = q;
    const int ldT = it.wT_ld > 0 ? it.wT_ld : it.K;      // horizontally fused convs share one [C][RS][Ktot] dgrad filter
    if (it.wT) ((T*)it.wT)[((int64_t)c * it.RS + (it.RS - 1 - rs)) * ldT + k] = q;
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
