// Conv2d_1a_3x3 (3x3 / stride 2 / no padding, 32 output channels) straight from the RESIZED u8 ROI plane.
//
// The reference feeds the stem with  PIL 'L' -> convert('RGB') -> Resize -> ToTensor -> Normalize  (neuston_data.py:342-371,
// 456-464): three copies of one grey plane g, each under its own affine x_c = a_c * g + b_c (ToTensor's /255, Normalize,
// [TV] transform_input folded into a_c, b_c).  The conv of those three planes is a ONE-plane conv plus a constant:
//     y[k] = sum_{r,s} g[r,s] * (sum_c a_c w[k,r,s,c])  +  sum_{r,s,c} b_c w[k,r,s,c]            (no padding: the constant is exact)
// and its weight gradient needs only  A[k,r,s] = sum dy[k] g[r,s]  and  B[k] = sum dy[k]:
//     dw[k,r,s,c] = a_c A[k,r,s] + b_c B[k].
// So the [B,S,S,8] input tensor (366 MB per batch of 256 at 299 px, written by the resize and read back 2.25 times by the GEMM
// kernels) never exists: the resize writes the u8 plane (23 MB), these kernels read it through L1/L2.  Both are bound by the one
// tensor they must move (the raw output / its gradient, 364 MB): HBM-bound, 9 taps x 32 channels of fp32 FMA per pixel is noise.
// Arithmetic: fp32 on the fp32 MASTER weights and exact u8 pixels (the GEMM path rounds x_c and w to bf16 first), outputs rounded
// to the storage type; BatchNorm partial sums over the rounded outputs like every other conv epilogue.
// Replaces aten::conv2d fwd / weight-grad of [TV] Inception3.Conv2d_1a_3x3 (reference call site neuston_models.py:66-68, 81-86).
#include "common.h"

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
};

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
        // (a dependent chain index -> bytes -> FMAs -> store per trip left the kernel latency bound at three waves per SIMD: 145 us for
        // 364 MB; out-of-range trips load pixel M-1 and skip the store)
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
    return a;
}

}  // namespace

// rows of the BatchNorm partial sums ifcbk_stem_u8_fwd writes (= its grid); 0: this descriptor is not served
extern "C" int ifcbk_stem_u8_rows(const ifcbk_conv_desc* d) { return stem_ok(d) ? cdiv((int64_t)d->N * d->P * d->Q, PIXB) : 0; }

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
        else hipLaunchKernelGGL((stem_u8_fwd_kernel<bf16_t, true>), grid, blk, 0, st, a);
    } else {
        if (f32) hipLaunchKernelGGL((stem_u8_fwd_kernel<float, false>), grid, blk, 0, st, a);
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
    else hipLaunchKernelGGL(stem_u8_wgrad_kernel<bf16_t>, dim3(nblk), dim3(256), 0, st, a);
    IFCBK_LAUNCH_CHECK(ctx, "stem_u8_wgrad");
    hipLaunchKernelGGL(stem_u8_wgrad_reduce_kernel, dim3(K1), dim3(640), 0, st, (const float*)ctx->ws, nblk, ab, dw, accumulate);
    IFCBK_LAUNCH_CHECK(ctx, "stem_u8_wgrad_reduce");
    return IFCBK_OK;
}
