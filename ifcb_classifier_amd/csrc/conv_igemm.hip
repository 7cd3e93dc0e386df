// Implicit-GEMM convolution forward / input-gradient on MFMA (gfx950), bf16 storage, fp32 accumulate.
//
//   C[m][n] = sum_k A[m][k] * B[n][k]      m = output pixel (n,p,q), n = output channel,
//                                          k = (r,s,c) with c fastest  (NHWC activations, KRSC filters)
// A is gathered on the fly (never materialised): each 16-byte chunk = 8 consecutive channels of one
// input pixel.  One kernel serves forward and dgrad: dgrad is the same gather over dy with the flipped,
// transposed filter and "input dilation" (positions not divisible by the stride contribute zero).
//
// Tile: 128 pixels x (32*NT) channels x 64 k per step, 4 waves (2x2), v_mfma_f32_16x16x32_bf16; both operand
// tiles are staged by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip) into a 2-stage ring: the DMA of
// tile k+1 is in flight while tile k is multiplied.
// The filter fragment is the MFMA A operand and the pixel fragment the B operand, so each lane ends with
// 4 consecutive CHANNELS of one pixel (one 8-byte LDS store per tile); the block tile is then written out
// through LDS as whole 16-byte chunks (coalesced rows) and the BatchNorm batch statistics (sum, sum of
// squares of the ROUNDED outputs) are reduced in the same pass -- no extra read of the conv output.
#include "common.h"

namespace {

struct ConvArgs {
    const bf16_t* x;
    const bf16_t* w;
    bf16_t* y;
    float* part;       // [mblocks][2][K] or null
    const bf16_t* zero;   // >= 16 bytes of zeros (source of padded / out-of-range chunks)
    int H, W, C, ldx;
    int K, R, S;
    int P, Q, ldy;
    int ostr_h, ostr_w, base_h, base_w, ish, isw;
    int M, Kg;
    int accumulate;
    int PQ;
    int tilesN;
};

constexpr int BM = 128;
constexpr int BK = 64;
constexpr int NTHREADS = 256;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS image of a tile: [rows][8 chunks of 16 B]; chunk c of row r lives at physical chunk c ^ (r & 7), which
// makes every ds_read_b128 fragment read conflict-free.  The image is filled by LDS-DMA
// (global_load_lds_dwordx4: destination = wave-uniform base + lane*16), so the swizzle is applied on the
// SOURCE side: the lane that lands on (row, phys) fetches logical chunk phys ^ (row & 7).
__device__ __forceinline__ const bf16x8_t* frag_ptr(const bf16_t* tile, int row, int chunk) {
    return reinterpret_cast<const bf16x8_t*>(tile + row * BK + ((chunk ^ (row & 7)) << 3));
}

template <int NT>
__global__ __launch_bounds__(NTHREADS) void conv_igemm_bf16(ConvArgs a) {
    constexpr int BN = 32 * NT;
    constexpr int CPR = BN / 8;                        // 16-byte chunks per output row
    constexpr int CPRP = CPR <= 4 ? 4 : CPR <= 8 ? 8 : CPR <= 16 ? 16 : 32;
    constexpr int LDC = BN + 8;                        // C-tile row stride (elements)
    constexpr int STAGE = (BM + BN) * BK;              // elements per pipeline stage
    constexpr int STAGE_BYTES = 2 * STAGE * 2;
    constexpr int CT_BYTES = BM * LDC * 2;
    constexpr int MAIN_BYTES = STAGE_BYTES > CT_BYTES ? STAGE_BYTES : CT_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[MAIN_BYTES + 4 * BN * 2 * 4];
    bf16_t* sStage = reinterpret_cast<bf16_t*>(smem);             // [2][A: BM*BK | B: BN*BK]
    bf16_t* sC = reinterpret_cast<bf16_t*>(smem);                 // [BM][LDC] (epilogue)
    float* sRed = reinterpret_cast<float*>(smem + MAIN_BYTES);    // [4][2][BN]

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int bid = blockIdx.x;
    const int mtile = bid / a.tilesN, ntile = bid - mtile * a.tilesN;
    const int m0 = mtile * BM, n0 = ntile * BN;

    // ---- LDS-DMA roles: one wave-instruction fills 8 tile rows (1 KiB); lane -> (row l>>3, phys chunk l&7)
    const int lrow8 = lane >> 3;
    const int csrc = (lane & 7) ^ lrow8;               // logical 16-byte chunk this lane fetches, every row group
    const bf16_t* xrow[4];
    int bh[4], bw[4];
    bool rvalid[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int m = m0 + (wave * 4 + j) * 8 + lrow8;
        rvalid[j] = m < a.M;
        int mm = rvalid[j] ? m : 0;
        int n = mm / a.PQ;
        int rem = mm - n * a.PQ;
        int p = rem / a.Q;
        int q = rem - p * a.Q;
        bh[j] = p * a.ostr_h + a.base_h;
        bw[j] = q * a.ostr_w + a.base_w;
        xrow[j] = a.x + (size_t)n * a.H * a.W * a.ldx;
    }
    int kc, kr, ks;
    {
        int k = csrc * 8;
        int rs = k / a.C;
        kc = k - rs * a.C;
        kr = rs / a.S;
        ks = rs - kr * a.S;
    }
    const bf16_t* wrow[NT];
    bool nvalid[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        int n = n0 + (j * 4 + wave) * 8 + lrow8;
        nvalid[j] = n < a.K;
        wrow[j] = a.w + (size_t)(nvalid[j] ? n : 0) * a.Kg + csrc * 8;
    }
    const int nk = (a.Kg + BK - 1) / BK;
    const int hmask = (1 << a.ish) - 1, wmask = (1 << a.isw) - 1;

#define ISSUE_TILE(kt, stage)                                                                              \
    {                                                                                                      \
        bf16_t* dstA = sStage + (stage) * STAGE;                                                           \
        bf16_t* dstB = dstA + BM * BK;                                                                     \
        const bool kvalid = kr < a.R;                                                                      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                    \
            int hr = bh[j] + kr, wr = bw[j] + ks;                                                          \
            bool v = rvalid[j] && kvalid && hr >= 0 && wr >= 0 && ((hr & hmask) == 0) && ((wr & wmask) == 0); \
            int hi = hr >> a.ish, wi = wr >> a.isw;                                                        \
            v = v && hi < a.H && wi < a.W;                                                                 \
            const bf16_t* src = v ? xrow[j] + ((size_t)hi * a.W + wi) * a.ldx + kc : a.zero;              \
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dstA + (wave * 4 + j) * 8 * BK), 16, 0, 0); \
        }                                                                                                  \
        const bool kv2 = ((kt) * BK + csrc * 8) < a.Kg;                                                    \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                   \
            const bf16_t* src = (nvalid[j] && kv2) ? wrow[j] + (size_t)(kt) * BK : a.zero;                 \
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dstB + (j * 4 + wave) * 8 * BK), 16, 0, 0); \
        }                                                                                                  \
        kc += BK;                                                                                          \
        while (kc >= a.C) {                                                                                \
            kc -= a.C;                                                                                     \
            if (++ks == a.S) { ks = 0; ++kr; }                                                             \
        }                                                                                                  \
    }

    f32x4_t acc[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    ISSUE_TILE(0, 0)
    __syncthreads();            // hipcc drains vmcnt(0) before the barrier: tile 0 has landed for every wave

    const int frow = lane & 15, fchunk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & 1;
        if (kt + 1 < nk) ISSUE_TILE(kt + 1, stage ^ 1)
        const bf16_t* tA = sStage + stage * STAGE;
        const bf16_t* tB = tA + BM * BK;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t fa[4], fb[NT];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) fa[mt] = *frag_ptr(tA, wm * 64 + mt * 16 + frow, kk * 4 + fchunk);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) fb[nt] = *frag_ptr(tB, wn * (NT * 16) + nt * 16 + frow, kk * 4 + fchunk);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[nt][mt], 0, 0, 0);
        }
        __syncthreads();        // next tile landed (vmcnt(0) drained) and this stage is free to refill
    }
#undef ISSUE_TILE

    // ---- epilogue: acc -> bf16 C tile in LDS (lane: 4 consecutive channels of one pixel)
    {
        const int g = lane >> 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                int m = wm * 64 + mt * 16 + frow;
                int n = wn * (NT * 16) + nt * 16 + 4 * g;
                uint2 v;
                v.x = pack2bf(acc[nt][mt][0], acc[nt][mt][1]);
                v.y = pack2bf(acc[nt][mt][2], acc[nt][mt][3]);
                *reinterpret_cast<uint2*>(sC + m * LDC + n) = v;
            }
    }
    __syncthreads();
    {
        constexpr int RPP = NTHREADS / CPRP;
        const int cc = t & (CPRP - 1);
        const int r0 = t / CPRP;
        const bool cvalid = (cc < CPR) && (n0 + cc * 8 < a.K);
        float s1[8], s2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
        if (cvalid) {
            for (int r = r0; r < BM; r += RPP) {
                int m = m0 + r;
                if (m >= a.M) break;
                uint4 v = *reinterpret_cast<const uint4*>(sC + r * LDC + cc * 8);
                bf16_t* dst = a.y + (size_t)m * a.ldy + n0 + cc * 8;
                if (a.accumulate) {
                    uint4 o = *reinterpret_cast<const uint4*>(dst);
                    float fo[8], fv[8];
                    unpack8(o, fo);
                    unpack8(v, fv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) fv[j] += fo[j];
                    v = pack8(fv);
                }
                *reinterpret_cast<uint4*>(dst) = v;
                if (a.part) {
                    float fv[8];
                    unpack8(v, fv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        s1[j] += fv[j];
                        s2[j] += fv[j] * fv[j];
                    }
                }
            }
        }
        if (a.part) {
#pragma unroll
            for (int off = CPRP; off < 64; off <<= 1)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    s1[j] += __shfl_xor(s1[j], off);
                    s2[j] += __shfl_xor(s2[j], off);
                }
            if (lane < CPRP && cc < CPR) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sRed[(wave * 2 + 0) * BN + cc * 8 + j] = s1[j];
                    sRed[(wave * 2 + 1) * BN + cc * 8 + j] = s2[j];
                }
            }
            __syncthreads();
            for (int i = t; i < 2 * BN; i += NTHREADS) {
                int which = i / BN, n = i - which * BN;
                if (n0 + n < a.K) {
                    float s = sRed[(0 * 2 + which) * BN + n] + sRed[(1 * 2 + which) * BN + n] +
                              sRed[(2 * 2 + which) * BN + n] + sRed[(3 * 2 + which) * BN + n];
                    a.part[((size_t)mtile * 2 + which) * a.K + n0 + n] = s;
                }
            }
        }
    }
}

int pick_nt(int K) {
    int best = 1;
    long bestc = -1;
    for (int nt = 1; nt <= 5; ++nt) {
        int bn = 32 * nt;
        long c = (long)cdiv(K, bn) * (bn + 48);
        if (bestc < 0 || c < bestc || (c == bestc && nt > best)) { bestc = c; best = nt; }
    }
    return best;
}

int check_desc(ifcbk_ctx* ctx, const ifcbk_conv_desc* d) {
    if (!d) IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv: null desc");
    if (d->dtype != IFCBK_BF16) IFCBK_FAIL(ctx, IFCBK_EUNSUPPORTED, "conv: only bf16 storage is implemented");
    if (d->C % 8 || d->K % 8 || d->ldx % 8 || d->ldy % 8 || d->C <= 0 || d->K <= 0)
        IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv: C=%d K=%d ldx=%d ldy=%d must be positive multiples of 8", d->C, d->K, d->ldx, d->ldy);
    if (d->stride_h < 1 || d->stride_h > 2 || d->stride_w < 1 || d->stride_w > 2)
        IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv: stride must be 1 or 2");
    int P = (d->H + 2 * d->pad_h - d->R) / d->stride_h + 1, Q = (d->W + 2 * d->pad_w - d->S) / d->stride_w + 1;
    if (P != d->P || Q != d->Q) IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv: P,Q=%d,%d inconsistent (expect %d,%d)", d->P, d->Q, P, Q);
    if ((int64_t)d->N * d->P * d->Q >= (1ll << 31) || (int64_t)d->N * d->H * d->W >= (1ll << 31))
        IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv: pixel count exceeds 2^31");
    return 0;
}

template <int NT>
void launch(const ConvArgs& a, int tilesM, hipStream_t st) {
    hipLaunchKernelGGL(conv_igemm_bf16<NT>, dim3((unsigned)(tilesM * a.tilesN)), dim3(NTHREADS), 0, st, a);
}

int run(ifcbk_ctx* ctx, ConvArgs& a, hipStream_t st) {
    int nt = pick_nt(a.K);
    a.tilesN = cdiv(a.K, 32 * nt);
    int tilesM = cdiv(a.M, BM);
    if ((int64_t)tilesM * a.tilesN >= (1ll << 31)) IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv: grid too large");
    switch (nt) {
        case 1: launch<1>(a, tilesM, st); break;
        case 2: launch<2>(a, tilesM, st); break;
        case 3: launch<3>(a, tilesM, st); break;
        case 4: launch<4>(a, tilesM, st); break;
        default: launch<5>(a, tilesM, st); break;
    }
    IFCBK_LAUNCH_CHECK(ctx, "conv_igemm_bf16");
    return 0;
}

}  // namespace

int ifcbk_conv_fwd_nt(int K) { return pick_nt(K); }

extern "C" int ifcbk_conv2d_fwd_mblocks(const ifcbk_conv_desc* d) { return cdiv((int64_t)d->N * d->P * d->Q, BM); }

extern "C" int ifcbk_conv2d_fwd(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const void* x, const void* w, void* y,
                                float* bn_part, void* stream) {
    if (int e = check_desc(ctx, d)) return e;
    ConvArgs a;
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.y = (bf16_t*)y; a.part = bn_part; a.zero = (const bf16_t*)ctx->zeros;
    a.H = d->H; a.W = d->W; a.C = d->C; a.ldx = d->ldx;
    a.K = d->K; a.R = d->R; a.S = d->S;
    a.P = d->P; a.Q = d->Q; a.ldy = d->ldy;
    a.ostr_h = d->stride_h; a.ostr_w = d->stride_w; a.base_h = -d->pad_h; a.base_w = -d->pad_w;
    a.ish = 0; a.isw = 0;
    a.M = d->N * d->P * d->Q; a.Kg = d->R * d->S * d->C; a.accumulate = 0; a.PQ = d->P * d->Q;
    return run(ctx, a, (hipStream_t)stream);
}

extern "C" int ifcbk_conv2d_dgrad(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const void* dy, const void* wT, void* dx,
                                  int accumulate, void* stream) {
    if (int e = check_desc(ctx, d)) return e;
    // gather over dy [N,P,Q,K] producing dx [N,H,W,C]: roles of (H,W,C) and (P,Q,K) swap
    ConvArgs a;
    a.x = (const bf16_t*)dy; a.w = (const bf16_t*)wT; a.y = (bf16_t*)dx; a.part = nullptr; a.zero = (const bf16_t*)ctx->zeros;
    a.H = d->P; a.W = d->Q; a.C = d->K; a.ldx = d->ldy;
    a.K = d->C; a.R = d->R; a.S = d->S;
    a.P = d->H; a.Q = d->W; a.ldy = d->ldx;
    a.ostr_h = 1; a.ostr_w = 1;
    a.base_h = -(d->R - 1 - d->pad_h); a.base_w = -(d->S - 1 - d->pad_w);
    a.ish = d->stride_h == 2 ? 1 : 0; a.isw = d->stride_w == 2 ? 1 : 0;
    a.M = d->N * d->H * d->W; a.Kg = d->R * d->S * d->K; a.accumulate = accumulate; a.PQ = d->H * d->W;
    return run(ctx, a, (hipStream_t)stream);
}
