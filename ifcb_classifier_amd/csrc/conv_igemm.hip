// Implicit-GEMM convolution forward / input-gradient on MFMA (gfx950), bf16 storage, fp32 accumulate.
//
//   C[m][n] = sum_k A[m][k] * B[n][k]      m = output pixel (n,p,q), n = output channel,
//                                          k = (r,s,c) with c fastest  (NHWC activations, KRSC filters)
// A is gathered on the fly (never materialised): each 16-byte chunk = 8 consecutive channels of one
// input pixel.  One kernel serves forward and dgrad: dgrad is the same gather over dy with the flipped,
// transposed filter and "input dilation" (positions not divisible by the stride contribute zero).
//
// Tile: 128 pixels x (32*NT) channels x 32 k per step, 4 waves (2x2), v_mfma_f32_16x16x32_bf16.
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
constexpr int BK = 32;
constexpr int NTHREADS = 256;

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((-(row >> 2)) & 3); }

template <int NT>
__global__ __launch_bounds__(NTHREADS) void conv_igemm_bf16(ConvArgs a) {
    constexpr int BN = 32 * NT;
    constexpr int PB = (BN + 63) / 64;                 // B-tile load passes
    constexpr int CPR = BN / 8;                        // 16-byte chunks per output row
    constexpr int CPRP = CPR <= 4 ? 4 : CPR <= 8 ? 8 : CPR <= 16 ? 16 : 32;
    constexpr int LDC = BN + 8;                        // C-tile row stride (elements)
    constexpr int STAGE_BYTES = 2 * (BM + BN) * BK * 2;
    constexpr int CT_BYTES = BM * LDC * 2;
    constexpr int MAIN_BYTES = STAGE_BYTES > CT_BYTES ? STAGE_BYTES : CT_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[MAIN_BYTES + 4 * BN * 2 * 4];
    bf16_t* sA = reinterpret_cast<bf16_t*>(smem);                 // [2][BM*BK]
    bf16_t* sB = sA + 2 * BM * BK;                                // [2][BN*BK]
    bf16_t* sC = reinterpret_cast<bf16_t*>(smem);                 // [BM][LDC] (epilogue)
    float* sRed = reinterpret_cast<float*>(smem + MAIN_BYTES);    // [4][2][BN]

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bid = blockIdx.x;
    const int mtile = bid / a.tilesN, ntile = bid - mtile * a.tilesN;
    const int m0 = mtile * BM, n0 = ntile * BN;

    // ---- per-thread gather state: chunk column (t&3), rows (t>>2) and (t>>2)+64
    const int lchunk = t & 3;
    const int lrow = t >> 2;
    const bf16_t* xrow[2];
    int bh[2], bw[2];
    bool rvalid[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int m = m0 + lrow + 64 * i;
        rvalid[i] = m < a.M;
        int mm = rvalid[i] ? m : 0;
        int n = mm / a.PQ;
        int rem = mm - n * a.PQ;
        int p = rem / a.Q;
        int q = rem - p * a.Q;
        bh[i] = p * a.ostr_h + a.base_h;
        bw[i] = q * a.ostr_w + a.base_w;
        xrow[i] = a.x + (size_t)n * a.H * a.W * a.ldx;
    }
    // k decode for this thread's chunk
    int kc, kr, ks;
    {
        int k = lchunk * 8;
        int rs = k / a.C;
        kc = k - rs * a.C;
        kr = rs / a.S;
        ks = rs - kr * a.S;
    }
    const bf16_t* wrow[PB];
    bool nvalid[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        int n = lrow + 64 * i;
        nvalid[i] = (n < BN) && (n0 + n < a.K);
        wrow[i] = a.w + (size_t)(nvalid[i] ? n0 + n : 0) * a.Kg + lchunk * 8;
    }

    const int nk = (a.Kg + BK - 1) / BK;
    uint4 ra[2], rb[PB];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

    auto load_tiles = [&](int kt) {
        const bool kvalid = kr < a.R;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int hr = bh[i] + kr, wr = bw[i] + ks;
            bool v = rvalid[i] && kvalid && hr >= 0 && wr >= 0 && ((hr & ((1 << a.ish) - 1)) == 0) &&
                     ((wr & ((1 << a.isw) - 1)) == 0);
            int hi = hr >> a.ish, wi = wr >> a.isw;
            v = v && hi < a.H && wi < a.W;
            ra[i] = v ? *reinterpret_cast<const uint4*>(xrow[i] + ((size_t)hi * a.W + wi) * a.ldx + kc) : zero4;
        }
        const bool kv2 = (kt * BK + lchunk * 8) < a.Kg;
#pragma unroll
        for (int i = 0; i < PB; ++i)
            rb[i] = (nvalid[i] && kv2) ? *reinterpret_cast<const uint4*>(wrow[i] + (size_t)kt * BK) : zero4;
        // advance (r,s,c) by BK
        kc += BK;
        while (kc >= a.C) {
            kc -= a.C;
            if (++ks == a.S) { ks = 0; ++kr; }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int row = lrow + 64 * i;
            *reinterpret_cast<uint4*>(sA + buf * BM * BK + row * BK + swz(row, lchunk) * 8) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            int row = lrow + 64 * i;
            if (row < BN) *reinterpret_cast<uint4*>(sB + buf * BN * BK + row * BK + swz(row, lchunk) * 8) = rb[i];
        }
    };

    f32x4_t acc[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    const int frow = lane & 15, fchunk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles(kt + 1);
        bf16x8_t fa[4], fb[NT];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            int row = wm * 64 + mt * 16 + frow;
            fa[mt] = *reinterpret_cast<const bf16x8_t*>(sA + buf * BM * BK + row * BK + swz(row, fchunk) * 8);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            int row = wn * (NT * 16) + nt * 16 + frow;
            fb[nt] = *reinterpret_cast<const bf16x8_t*>(sB + buf * BN * BK + row * BK + swz(row, fchunk) * 8);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[nt][mt], 0, 0, 0);
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

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

extern "C" int ifcbk_conv2d_fwd_mblocks(const ifcbk_conv_desc* d) { return cdiv((int64_t)d->N * d->P * d->Q, BM); }

extern "C" int ifcbk_conv2d_fwd(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const void* x, const void* w, void* y,
                                float* bn_part, void* stream) {
    if (int e = check_desc(ctx, d)) return e;
    ConvArgs a;
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.y = (bf16_t*)y; a.part = bn_part;
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
    a.x = (const bf16_t*)dy; a.w = (const bf16_t*)wT; a.y = (bf16_t*)dx; a.part = nullptr;
    a.H = d->P; a.W = d->Q; a.C = d->K; a.ldx = d->ldy;
    a.K = d->C; a.R = d->R; a.S = d->S;
    a.P = d->H; a.Q = d->W; a.ldy = d->ldx;
    a.ostr_h = 1; a.ostr_w = 1;
    a.base_h = -(d->R - 1 - d->pad_h); a.base_w = -(d->S - 1 - d->pad_w);
    a.ish = d->stride_h == 2 ? 1 : 0; a.isw = d->stride_w == 2 ? 1 : 0;
    a.M = d->N * d->H * d->W; a.Kg = d->R * d->S * d->K; a.accumulate = accumulate; a.PQ = d->H * d->W;
    return run(ctx, a, (hipStream_t)stream);
}
