// BatchNorm (train/eval) fused with ReLU and the resnet residual add; HBM-bound elementwise/reduction kernels.
// Every access is a 16-byte chunk (8 bf16 or 4 fp32 channels), consecutive lanes on consecutive chunks; all
// kernels are templated on the storage type T (bf16 performance mode / fp32 parity mode).
#include "common.h"
#include <stdlib.h>

namespace {

// timing experiment (never set in the product; scripts/README.md): IFCBK_EXPERIMENT_NOFINALIZE=<n> skips every finalize launch
// after the first n -- the statistics of the first steps stay in place, so the data keep their scale --, with
// IFCBK_EXPERIMENT_EMPTYFINALIZE=1 an empty one-wave kernel takes its place.  What the 192 finalize launches of an inception_v3
// step cost: DESIGN 5.10
__global__ void experiment_empty_kernel(float* p) { if (p && threadIdx.x == 1000) p[0] = 0.f; }
bool experiment_skip_finalize(hipStream_t st) {
    static const char* e = getenv("IFCBK_EXPERIMENT_NOFINALIZE");
    static const char* k = getenv("IFCBK_EXPERIMENT_EMPTYFINALIZE");
    static long seen = 0;
    if (!e) return false;
    const bool skip = ++seen > atol(e);
    if (skip && k) hipLaunchKernelGGL(experiment_empty_kernel, dim3(1), dim3(64), 0, st, (float*)nullptr);
    return skip;
}

// ---------------------------------------------------------------- finalize
// one block per 16 channels; 64 row groups stride over the per-M-block partials written by the conv epilogue
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* part, int mblocks, int ldp, int C, double invM,
                                                            double unbias, const float* gamma, const float* beta,
                                                            float* rmean, float* rvar, float* mean_o, float* invstd_o,
                                                            float* scale, float* shift, float eps, float momentum) {
    __shared__ double s1[64][16], s2[64][16];
    const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int ch = blockIdx.x * 16 + c;
    double a = 0.0, b = 0.0;
    if (ch < C) {
        // four row blocks per trip: the loads of a trip are independent (one memory round trip), the adds keep a fixed order
        int mb = rg;
        for (; mb + 192 < mblocks; mb += 256) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[2 * u] = part[((size_t)(mb + 64 * u) * 2 + 0) * ldp + ch];
                v[2 * u + 1] = part[((size_t)(mb + 64 * u) * 2 + 1) * ldp + ch];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a += (double)v[2 * u];
                b += (double)v[2 * u + 1];
            }
        }
        for (; mb < mblocks; mb += 64) {
            a += (double)part[((size_t)mb * 2 + 0) * ldp + ch];
            b += (double)part[((size_t)mb * 2 + 1) * ldp + ch];
        }
    }
    s1[rg][c] = a;
    s2[rg][c] = b;
    __syncthreads();
    if (threadIdx.x < 16 && ch < C) {
        double sa = 0.0, sb = 0.0;
        for (int i = 0; i < 64; ++i) {
            sa += s1[i][c];
            sb += s2[i][c];
        }
        double mean = sa * invM;
        double var = sb * invM - mean * mean;
        if (var < 0.0) var = 0.0;
        float invstd = (float)(1.0 / sqrt(var + (double)eps));
        float g = gamma[ch], sc = g * invstd;
        mean_o[ch] = (float)mean;
        invstd_o[ch] = invstd;
        scale[ch] = sc;
        shift[ch] = beta[ch] - (float)mean * sc;
        if (rmean) {
            rmean[ch] = (1.f - momentum) * rmean[ch] + momentum * (float)mean;
            rvar[ch] = (1.f - momentum) * rvar[ch] + momentum * (float)(var * unbias);
        }
    }
}

// stage 0 for layers with many M-blocks: chunk `blockIdx.y` of the partial rows -> one fp32 row (double accumulate)
__global__ __launch_bounds__(256) void bn_prereduce_kernel(const float* part, int mblocks, int ldp, int C, int rows_per_chunk,
                                                           float* out) {
    __shared__ double s[2][4][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int ch = blockIdx.x * 64 + c;
    const int r0 = blockIdx.y * rows_per_chunk;
    const int r1 = min(r0 + rows_per_chunk, mblocks);
    double a = 0.0, b = 0.0;
    if (ch < C)
        for (int mb = r0 + rg; mb < r1; mb += 4) {
            a += (double)part[((size_t)mb * 2 + 0) * ldp + ch];
            b += (double)part[((size_t)mb * 2 + 1) * ldp + ch];
        }
    s[0][rg][c] = a;
    s[1][rg][c] = b;
    __syncthreads();
    if (rg == 0 && ch < C) {
        out[((size_t)blockIdx.y * 2 + 0) * C + ch] = (float)((s[0][0][c] + s[0][1][c]) + (s[0][2][c] + s[0][3][c]));
        out[((size_t)blockIdx.y * 2 + 1) * C + ch] = (float)((s[1][0][c] + s[1][1][c]) + (s[1][2][c] + s[1][3][c]));
    }
}

__global__ void bn_eval_scale_kernel(int C, const float* gamma, const float* beta, const float* rmean,
                                     const float* rvar, float* scale, float* shift, float eps) {
    int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= C) return;
    float sc = gamma[ch] / sqrtf(rvar[ch] + eps);
    scale[ch] = sc;
    shift[ch] = beta[ch] - rmean[ch] * sc;
}

// ---------------------------------------------------------------- statistics of a stored tensor
// partial (sum x, sum x^2) per 1024-row tile; block: 8 chunks x 32 rows in flight, fixed reduction order
constexpr int STAT_ROWS = 1024;
template <class T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* x, int ldx, float* part, int64_t M, int C) {
    constexpr int E = Chunk<T>::N;
    constexpr int CG = 8 * E;
    __shared__ float red[4][2][CG];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int cc = t & 7, r0 = t >> 3;
    const int c = blockIdx.y * CG + cc * E;
    const int64_t mbeg = (int64_t)blockIdx.x * STAT_ROWS;
    const int64_t mend = mbeg + STAT_ROWS < M ? mbeg + STAT_ROWS : M;
    float s1[E], s2[E];
#pragma unroll
    for (int j = 0; j < E; ++j) s1[j] = s2[j] = 0.f;
    if (c < C) {
        int64_t m = mbeg + r0;
        for (; m + 96 < mend; m += 128) {          // four rows per trip: independent loads, fixed add order
            float f[4][E];
#pragma unroll
            for (int u = 0; u < 4; ++u) Chunk<T>::load(x + (m + 32 * u) * ldx + c, f[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < E; ++j) { s1[j] += f[u][j]; s2[j] += f[u][j] * f[u][j]; }
        }
        for (; m < mend; m += 32) {
            float f[E];
            Chunk<T>::load(x + m * ldx + c, f);
#pragma unroll
            for (int j = 0; j < E; ++j) { s1[j] += f[j]; s2[j] += f[j] * f[j]; }
        }
    }
#pragma unroll
    for (int off = 8; off < 64; off <<= 1)
#pragma unroll
        for (int j = 0; j < E; ++j) {
            s1[j] += __shfl_xor(s1[j], off);
            s2[j] += __shfl_xor(s2[j], off);
        }
    if (lane < 8) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            red[wave][0][cc * E + j] = s1[j];
            red[wave][1][cc * E + j] = s2[j];
        }
    }
    __syncthreads();
    if (t < 2 * CG) {
        int which = t / CG, n = t - which * CG;
        int ch = blockIdx.y * CG + n;
        if (ch < C) part[((size_t)blockIdx.x * 2 + which) * C + ch] = (red[0][which][n] + red[1][which][n]) + (red[2][which][n] + red[3][which][n]);
    }
}

// ---------------------------------------------------------------- apply
// per-channel coefficients are staged once per block in LDS (8 chunks per thread amortise it): the payload loads are
// the only global traffic in the loop
constexpr int EW_ITER = 8;
template <class T, bool RELU, bool RES>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* x, int ldx, const float* scale, const float* shift,
                                                        const T* res, int ldr, T* y, int ldy, int64_t M, int cpr,
                                                        fastdiv_t fcpr) {
    constexpr int E = Chunk<T>::N;
    extern __shared__ __attribute__((aligned(16))) float coef[];
    const int C = cpr * E;
    for (int c = threadIdx.x; c < C; c += 256) {
        coef[c] = scale[c];
        coef[C + c] = shift[c];
    }
    __syncthreads();
    const int64_t total = M * cpr;
    const int64_t i0 = (int64_t)blockIdx.x * (256 * EW_ITER) + threadIdx.x;
    // All EW_ITER loads of a thread are issued before its first store: memory operations retire in order, so a load issued behind a
    // store cannot be waited for without waiting for that store's round trip too (load -> store -> load -> store was four serial round
    // trips per thread; trips past the end re-read the last chunk and store nothing)
    typename Chunk<T>::raw_t rx[EW_ITER], rq[EW_ITER];
    uint32_t mm[EW_ITER];
    int cc[EW_ITER];
#pragma unroll
    for (int it = 0; it < EW_ITER; ++it) {
        const int64_t i = i0 + it * 256 < total ? i0 + it * 256 : total - 1;
        mm[it] = fdiv((uint32_t)i, fcpr);
        cc[it] = ((int)i - (int)mm[it] * cpr) * E;
        rx[it] = Chunk<T>::load_raw(x + (int64_t)mm[it] * ldx + cc[it]);
        if (RES) rq[it] = Chunk<T>::load_raw(res + (int64_t)mm[it] * ldr + cc[it]);
    }
#pragma unroll
    for (int it = 0; it < EW_ITER; ++it) {
        if (i0 + it * 256 >= total) break;
        const int c = cc[it];
        float f[E], r[E];
        Chunk<T>::widen(rx[it], f);
        if (RES) Chunk<T>::widen(rq[it], r);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            float o = f[j] * coef[c + j] + coef[C + c + j];
            if (RES) o += r[j];
            if (RELU) o = fmaxf(o, 0.f);
            f[j] = o;
        }
        Chunk<T>::store(y + (int64_t)mm[it] * ldy + c, f);
    }
}

// ---------------------------------------------------------------- upstream gradient through a 3x3/stride-2 max pool
// bn_bwd_maxpool: the activation of this BN feeds ONLY a max pool (inception Conv2d_2b / Conv2d_4a, the resnet stem),
// so neither the activation nor its gradient is ever materialised: dy[n,h,w,c] is gathered from the pooled gradient
// and the pool's u8 arg-max (an input pixel lies in at most 2x2 windows) -- the arithmetic of maxpool3x3s2_bwd_kernel.
struct PoolGather {
    const uint8_t* arg;
    int H, W, P, Q, ph, pw, ldp, C;
    fastdiv_t fHW, fW;
};
template <class T>
__device__ __forceinline__ void pooled_dy(const PoolGather& g, const T* dp, uint32_t m, int c, float* fd) {
    constexpr int E = Chunk<T>::N;
    const uint32_t n = fdiv(m, g.fHW);
    const uint32_t rem = m - n * g.fHW.d;
    const int h = (int)fdiv(rem, g.fW);
    const int w = (int)rem - h * g.W;
    int plo = h + g.ph - 2; plo = plo <= 0 ? 0 : (plo + 1) >> 1;
    int phi = (h + g.ph) >> 1; if (phi >= g.P) phi = g.P - 1;
    int qlo = w + g.pw - 2; qlo = qlo <= 0 ? 0 : (qlo + 1) >> 1;
    int qhi = (w + g.pw) >> 1; if (qhi >= g.Q) qhi = g.Q - 1;
#pragma unroll
    for (int j = 0; j < E; ++j) fd[j] = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int p = plo + u, q = qlo + v;
            if (p <= phi && q <= qhi) {
                const int want = (h - (p * 2 - g.ph)) * 3 + (w - (q * 2 - g.pw));
                const int64_t opix = (int64_t)(n * g.P + p) * g.Q + q;
                int idx[E];
                float f[E];
                ArgPack<E>::load(g.arg + opix * g.C + c, idx);
                Chunk<T>::load(dp + opix * g.ldp + c, f);
#pragma unroll
                for (int j = 0; j < E; ++j)
                    if (idx[j] == want) fd[j] += f[j];
            }
        }
}

// Unpadded pools (inception): a thread takes a 2x2 block of input pixels.  The block meets only the windows
// (hb-1..hb, wb-1..wb): four (arg-max, gradient) loads and four arg-max unpacks serve nine (pixel, window) incidences
// instead of nine loads and unpacks -- the per-pixel gather above is VALU-bound (~160 instructions per 16-byte chunk).
// fd[k]: pixel (2hb + k/2, 2wb + k%2).
template <class T>
__device__ __forceinline__ void pooled_dy_2x2(const PoolGather& g, const T* dp, uint32_t n, int hb, int wb, int c,
                                              float (*fd)[Chunk<T>::N]) {
    constexpr int E = Chunk<T>::N;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < E; ++j) fd[k][j] = 0.f;
    // all eight loads are issued before the first is used: windows outside the pooled map read a clamped address and are
    // switched off by data (arg-max 255 matches no tap).  With the loads inside `if (inside) { load; use }` bodies every window
    // cost its own memory round trip and these kernels ran at 3.2-3.8 TB/s, latency-bound
    typename Chunk<T>::raw_t rf[4];
    int idx[4][E];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int p = hb - 1 + u, q = wb - 1 + v;
            const bool in = p >= 0 && p < g.P && q >= 0 && q < g.Q;
            const int pc = p < 0 ? 0 : (p >= g.P ? g.P - 1 : p), qc = q < 0 ? 0 : (q >= g.Q ? g.Q - 1 : q);
            const int64_t opix = (int64_t)(n * g.P + pc) * g.Q + qc;
            ArgPack<E>::load(g.arg + opix * g.C + c, idx[u * 2 + v]);
            rf[u * 2 + v] = Chunk<T>::load_raw(dp + opix * g.ldp + c);
            if (!in) {
#pragma unroll
                for (int j = 0; j < E; ++j) idx[u * 2 + v][j] = 255;
            }
        }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            float f[E];
            Chunk<T>::widen(rf[u * 2 + v], f);
            // window (u,v) holds pixel (a,b) of the block at tap (a + 2(1-u), b + 2(1-v)) when that is < 3
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int r = a + 2 * (1 - u), sx = b + 2 * (1 - v);
                    if (r > 2 || sx > 2) continue;
                    const int want = r * 3 + sx;
#pragma unroll
                    for (int j = 0; j < E; ++j)
                        if (idx[u * 2 + v][j] == want) fd[a * 2 + b][j] += f[j];
                }
        }
}

template <class T, int MASK>
__global__ __launch_bounds__(256) void bn_bwd_reduce_pool2x2_kernel(const T* x, int ldx, const T* dp, const float* mean,
                                                                     const float* invstd, const float* scale, const float* shift,
                                                                     float* part, int C, PoolGather pg, int HB, int WB,
                                                                     uint32_t nblk, fastdiv_t fHBWB, fastdiv_t fWB) {
    constexpr int E = Chunk<T>::N;
    constexpr int CG = 8 * E;
    constexpr int TILE = 256;                    // 2x2 blocks per partial row (1024 pixels)
    __shared__ float red[4][2][CG];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int cc = t & 7, r0 = t >> 3;
    const int c = blockIdx.y * CG + cc * E;
    const uint32_t bbeg = blockIdx.x * TILE;
    const uint32_t bend = bbeg + TILE < nblk ? bbeg + TILE : nblk;
    float sb[E], sg[E];
#pragma unroll
    for (int j = 0; j < E; ++j) sb[j] = sg[j] = 0.f;
    if (c < C) {
        float mu[E], is[E], sc[E], sh[E];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            mu[j] = mean[c + j];
            is[j] = invstd[c + j];
            sc[j] = MASK == 2 ? scale[c + j] : 0.f;
            sh[j] = MASK == 2 ? shift[c + j] : 0.f;
        }
        for (uint32_t b = bbeg + r0; b < bend; b += 32) {
            const uint32_t n = fdiv(b, fHBWB);
            const uint32_t rem = b - n * fHBWB.d;
            const int hb = (int)fdiv(rem, fWB);
            const int wb = (int)rem - hb * WB;
            typename Chunk<T>::raw_t rx[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {           // (a block on the odd last row / column re-reads its neighbour; masked below)
                const int h = min(2 * hb + (k >> 1), pg.H - 1), w = min(2 * wb + (k & 1), pg.W - 1);
                rx[k] = Chunk<T>::load_raw(x + ((int64_t)(n * pg.H + h) * pg.W + w) * ldx + c);
            }
            float fd[4][E];
            pooled_dy_2x2<T>(pg, dp, n, hb, wb, c, fd);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool in = 2 * hb + (k >> 1) < pg.H && 2 * wb + (k & 1) < pg.W;
                float fx[E];
                Chunk<T>::widen(rx[k], fx);
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    float dz = in ? fd[k][j] : 0.f;
                    if (MASK == 2) dz = (fx[j] * sc[j] + sh[j]) > 0.f ? dz : 0.f;
                    sb[j] += dz;
                    sg[j] += dz * ((fx[j] - mu[j]) * is[j]);
                }
            }
        }
    }
#pragma unroll
    for (int off = 8; off < 64; off <<= 1)
#pragma unroll
        for (int j = 0; j < E; ++j) {
            sb[j] += __shfl_xor(sb[j], off);
            sg[j] += __shfl_xor(sg[j], off);
        }
    if (lane < 8) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            red[wave][0][cc * E + j] = sb[j];
            red[wave][1][cc * E + j] = sg[j];
        }
    }
    __syncthreads();
    if (t < 2 * CG) {
        int which = t / CG, nn = t - which * CG;
        int ch = blockIdx.y * CG + nn;
        if (ch < C) {
            float s_ = red[0][which][nn] + red[1][which][nn] + red[2][which][nn] + red[3][which][nn];
            part[((size_t)blockIdx.x * 2 + which) * C + ch] = s_;
        }
    }
}

template <class T, int MASK>
__global__ __launch_bounds__(256) void bn_bwd_dx_pool2x2_kernel(const T* x, int ldx, const T* dp, const float* gamma,
                                                                 const float* mean, const float* invstd, const float* scale,
                                                                 const float* shift, const float* tmp, T* dx, int lddx, int C,
                                                                 float invM, PoolGather pg, int HB, int WB, uint32_t total,
                                                                 fastdiv_t fcpr, fastdiv_t fHBWB, fastdiv_t fWB) {
    constexpr int E = Chunk<T>::N;
    extern __shared__ __attribute__((aligned(16))) float coef[];
    for (int c = threadIdx.x; c < C; c += 256) {
        float is = invstd[c], a_ = gamma[c] * is, dg = tmp[C + c] * invM, db = tmp[c] * invM;
        coef[c] = a_;
        coef[C + c] = -a_ * is * dg;
        coef[2 * C + c] = a_ * (mean[c] * is * dg - db);
        if (MASK == 2) {
            coef[3 * C + c] = scale[c];
            coef[4 * C + c] = shift[c];
        }
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= total) return;
    const uint32_t b = fdiv(i, fcpr);
    const int c = (int)(i - b * fcpr.d) * E;
    const uint32_t n = fdiv(b, fHBWB);
    const uint32_t rem = b - n * fHBWB.d;
    const int hb = (int)fdiv(rem, fWB);
    const int wb = (int)rem - hb * WB;
    typename Chunk<T>::raw_t rx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int h = min(2 * hb + (k >> 1), pg.H - 1), w = min(2 * wb + (k & 1), pg.W - 1);
        rx[k] = Chunk<T>::load_raw(x + ((int64_t)(n * pg.H + h) * pg.W + w) * ldx + c);
    }
    float fd[4][E];
    pooled_dy_2x2<T>(pg, dp, n, hb, wb, c, fd);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int h = 2 * hb + (k >> 1), w = 2 * wb + (k & 1);
        if (h >= pg.H || w >= pg.W) continue;
        const int64_t pix = (int64_t)(n * pg.H + h) * pg.W + w;
        float fx[E], o[E];
        Chunk<T>::widen(rx[k], fx);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            float dz = fd[k][j];
            if (MASK == 2) dz = (fx[j] * coef[3 * C + c + j] + coef[4 * C + c + j]) > 0.f ? dz : 0.f;
            o[j] = coef[c + j] * dz + (coef[C + c + j] * fx[j] + coef[2 * C + c + j]);
        }
        Chunk<T>::store(dx + pix * lddx + c, o);
    }
}

// ---------------------------------------------------------------- backward
// pass 1: per-row-tile partial sums of dz and dz*xhat.  block: 8 chunks x 32 rows in flight
constexpr int BWD_ROWS = 1024;      // largest row tile (the engine reserves the workspace for 256-row tiles: bwd_rows() stays >= 256)
// Rows per tile of the reduction pass: a tile is one block, and with 1024-row tiles a 17x17 layer (73,984 rows x 192 channels)
// was 219 blocks of 256 threads on 256 CUs -- 4 waves per CU, two loads in flight per lane: 2.2 TB/s.  Smaller tiles where the
// tensor is small: ~3,000 blocks or more, never below 256 rows (workspace), never above 1024 (the finalize pass reads the rows)
static inline int bwd_rows(int64_t M, int cgroups) {
    static int force = -1;
    if (force < 0) { const char* e = getenv("IFCBK_BN_BWD_ROWS"); force = e ? atoi(e) : 0; }
    if (force >= 32) return force / 32 * 32;
    int64_t r = M * cgroups / 3000;
    r = (r + 31) / 32 * 32;
    return (int)(r < 256 ? 256 : r > BWD_ROWS ? BWD_ROWS : r);
}
// MASK: 0 = no ReLU, 1 = ReLU mask read from y (residual case), 2 = ReLU mask recomputed as x*scale+shift > 0
// (bit-identical to what bn_apply stored: same expression, the sign survives rounding) -- saves the y read
template <class T, int MASK, bool POOLED>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* x, int ldx, const T* y, int ldy, const T* dy, int lddy,
                                                             const float* mean, const float* invstd, const float* scale,
                                                             const float* shift, float* part, int64_t M, int C, PoolGather pg, int rows) {
    constexpr int E = Chunk<T>::N;
    constexpr int CG = 8 * E;                    // channels per block
    __shared__ float red[4][2][CG];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int cc = t & 7, r0 = t >> 3;
    const int c = blockIdx.y * CG + cc * E;
    const int64_t mbeg = (int64_t)blockIdx.x * rows;
    const int64_t mend = mbeg + rows < M ? mbeg + rows : M;
    float sb[E], sg[E];
#pragma unroll
    for (int j = 0; j < E; ++j) sb[j] = sg[j] = 0.f;
    if (c < C) {
        float mu[E], is[E], sc[E], sh[E];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            mu[j] = mean[c + j];
            is[j] = invstd[c + j];
            sc[j] = MASK == 2 ? scale[c + j] : 0.f;
            sh[j] = MASK == 2 ? shift[c + j] : 0.f;
        }
        for (int64_t m = mbeg + r0; m < mend; m += 32) {
            float fx[E], fy[E], fd[E];
            Chunk<T>::load(x + m * ldx + c, fx);
            if (POOLED) pooled_dy<T>(pg, dy, (uint32_t)m, c, fd);
            else Chunk<T>::load(dy + m * lddy + c, fd);
            if (MASK == 1) Chunk<T>::load(y + m * ldy + c, fy);
#pragma unroll
            for (int j = 0; j < E; ++j) {
                float dz = fd[j];
                if (MASK == 1) dz = fy[j] > 0.f ? dz : 0.f;
                if (MASK == 2) dz = (fx[j] * sc[j] + sh[j]) > 0.f ? dz : 0.f;
                sb[j] += dz;
                sg[j] += dz * ((fx[j] - mu[j]) * is[j]);
            }
        }
    }
#pragma unroll
    for (int off = 8; off < 64; off <<= 1)
#pragma unroll
        for (int j = 0; j < E; ++j) {
            sb[j] += __shfl_xor(sb[j], off);
            sg[j] += __shfl_xor(sg[j], off);
        }
    if (lane < 8) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            red[wave][0][cc * E + j] = sb[j];
            red[wave][1][cc * E + j] = sg[j];
        }
    }
    __syncthreads();
    if (t < 2 * CG) {
        int which = t / CG, n = t - which * CG;
        int ch = blockIdx.y * CG + n;
        if (ch < C) {
            float s = red[0][which][n] + red[1][which][n] + red[2][which][n] + red[3][which][n];
            part[((size_t)blockIdx.x * 2 + which) * C + ch] = s;
        }
    }
}

// pass 2: sum the row-tile partials (fixed order) -> dbeta, dgamma (+ temp copy used by pass 3)
// (1024 threads = 64 row groups x 16 channels: the 17x17 layers have 289 row tiles -- with 16 row groups a thread walked 18 of
// them in five dependent round trips and the kernel took 11.7 us, 96 times per step)
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* part, int ntiles, int C, float* dgamma,
                                                                float* dbeta, float* tmp, int accumulate, int ldp) {
    __shared__ double s[2][64][16];
    const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int ch = blockIdx.x * 16 + c;
    double a = 0.0, b = 0.0;
    if (ch < C) {
        int i = rg;
        for (; i + 192 < ntiles; i += 256) {         // four tiles per trip: independent loads, fixed-order adds
            float v[8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[2 * u] = part[((size_t)(i + 64 * u) * 2 + 0) * ldp + ch];
                v[2 * u + 1] = part[((size_t)(i + 64 * u) * 2 + 1) * ldp + ch];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a += (double)v[2 * u];
                b += (double)v[2 * u + 1];
            }
        }
        for (; i < ntiles; i += 64) {
            a += (double)part[((size_t)i * 2 + 0) * ldp + ch];
            b += (double)part[((size_t)i * 2 + 1) * ldp + ch];
        }
    }
    s[0][rg][c] = a;
    s[1][rg][c] = b;
    __syncthreads();
    if (threadIdx.x < 16 && ch < C) {
        double sa = 0.0, sg = 0.0;
        for (int i = 0; i < 64; ++i) {
            sa += s[0][i][c];
            sg += s[1][i][c];
        }
        tmp[ch] = (float)sa;        // dbeta
        tmp[C + ch] = (float)sg;    // dgamma
        dbeta[ch] = accumulate ? dbeta[ch] + (float)sa : (float)sa;
        dgamma[ch] = accumulate ? dgamma[ch] + (float)sg : (float)sg;
    }
}

// pass 3: dx (and the residual branch gradient).  dx = A_c*dz + B_c*x + K_c with
//   A = gamma*invstd,  B = -gamma*invstd^2*dgamma/M,  K = gamma*invstd*(mean*invstd*dgamma - dbeta)/M
// staged per block in LDS together with bn_apply's (scale, shift) for the recomputed ReLU mask.
template <class T, int MASK, bool POOLED>
__global__ __launch_bounds__(256) void bn_bwd_dx_kernel(const T* x, int ldx, const T* y, int ldy, const T* dy, int lddy,
                                                         const float* gamma, const float* mean, const float* invstd,
                                                         const float* scale, const float* shift, const float* tmp,
                                                         T* dx, int lddx, T* dres, int lddres, int dres_acc, int64_t M,
                                                         int C, float invM, fastdiv_t fcpr, PoolGather pg) {
    constexpr int E = Chunk<T>::N;
    extern __shared__ __attribute__((aligned(16))) float coef[];
    for (int c = threadIdx.x; c < C; c += 256) {
        float is = invstd[c], a_ = gamma[c] * is, dg = tmp[C + c] * invM, db = tmp[c] * invM;
        coef[c] = a_;
        coef[C + c] = -a_ * is * dg;
        coef[2 * C + c] = a_ * (mean[c] * is * dg - db);
        if (MASK == 2) {
            coef[3 * C + c] = scale[c];
            coef[4 * C + c] = shift[c];
        }
    }
    __syncthreads();
    const int cpr = C / E;
    const int64_t total = M * cpr;
    int64_t i = (int64_t)blockIdx.x * (256 * EW_ITER) + threadIdx.x;
#pragma unroll 2
    for (int it = 0; it < EW_ITER; ++it, i += 256) {
        if (i >= total) break;
        uint32_t m = fdiv((uint32_t)i, fcpr);
        int c = ((int)i - (int)m * cpr) * E;
        float fx[E], fy[E], fd[E], o[E];
        Chunk<T>::load(x + (int64_t)m * ldx + c, fx);
        if (POOLED) pooled_dy<T>(pg, dy, m, c, fd);
        else Chunk<T>::load(dy + (int64_t)m * lddy + c, fd);
        if (MASK == 1) Chunk<T>::load(y + (int64_t)m * ldy + c, fy);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            float dz = fd[j];
            if (MASK == 1) dz = fy[j] > 0.f ? dz : 0.f;
            if (MASK == 2) dz = (fx[j] * coef[3 * C + c + j] + coef[4 * C + c + j]) > 0.f ? dz : 0.f;
            fd[j] = dz;
            o[j] = coef[c + j] * dz + (coef[C + c + j] * fx[j] + coef[2 * C + c + j]);
        }
        if (dres_acc & 2) {                      // dx accumulates (densenet: a concatenation's gradient collects every later layer's share)
            float fo[E];
            Chunk<T>::load(dx + (int64_t)m * lddx + c, fo);
#pragma unroll
            for (int j = 0; j < E; ++j) o[j] += fo[j];
        }
        if (dres) {
            T* rp = dres + (int64_t)m * lddres + c;
            if (dres_acc & 1) {
                float fr[E];
                Chunk<T>::load(rp, fr);
#pragma unroll
                for (int j = 0; j < E; ++j) fd[j] += fr[j];
            }
            Chunk<T>::store(rp, fd);
        }
        Chunk<T>::store(dx + (int64_t)m * lddx + c, o);
    }
}

template <class T>
int apply_t(ifcbk_ctx* ctx, const ifcbk_bn_desc* d, const void* x, const float* scale, const float* shift,
            const void* residual, int ldr, void* y, hipStream_t st) {
    constexpr int E = Chunk<T>::N;
    if (d->C % E || d->ldx % E || d->ldy % E) IFCBK_FAIL(ctx, IFCBK_EINVAL, "bn_apply: channels must be multiples of %d", E);
    const int cpr = d->C / E;
    const int64_t total = (int64_t)d->M * cpr;
    if (total >= (1ll << 31)) IFCBK_FAIL(ctx, IFCBK_EINVAL, "bn_apply: tensor too large");
    dim3 grid(cdiv(total, 256 * EW_ITER)), block(256);
    const size_t shm = (size_t)2 * d->C * sizeof(float);
    const fastdiv_t fc = make_fastdiv(cpr);
    const T* xx = (const T*)x; const T* rr = (const T*)residual; T* yy = (T*)y;
    if (d->relu && rr) hipLaunchKernelGGL((bn_apply_kernel<T, true, true>), grid, block, shm, st, xx, d->ldx, scale, shift, rr, ldr, yy, d->ldy, (int64_t)d->M, cpr, fc);
    else if (d->relu) hipLaunchKernelGGL((bn_apply_kernel<T, true, false>), grid, block, shm, st, xx, d->ldx, scale, shift, rr, ldr, yy, d->ldy, (int64_t)d->M, cpr, fc);
    else if (rr) hipLaunchKernelGGL((bn_apply_kernel<T, false, true>), grid, block, shm, st, xx, d->ldx, scale, shift, rr, ldr, yy, d->ldy, (int64_t)d->M, cpr, fc);
    else hipLaunchKernelGGL((bn_apply_kernel<T, false, false>), grid, block, shm, st, xx, d->ldx, scale, shift, rr, ldr, yy, d->ldy, (int64_t)d->M, cpr, fc);
    IFCBK_LAUNCH_CHECK(ctx, "bn_apply");
    return 0;
}

template <class T>
int bwd_t(ifcbk_ctx* ctx, const ifcbk_bn_desc* d, const void* x, const void* y, const void* dy, int lddy,
          const float* gamma, const float* mean, const float* invstd, void* dx, int lddx, void* dres, int lddres,
          int dres_accumulate, float* dgamma, float* dbeta, int param_accumulate, const float* scale, const float* shift,
          hipStream_t st, const PoolGather* pool = nullptr, const float* part_in = nullptr, int ntiles_in = 0, int part_ld_in = 0) {
    constexpr int E = Chunk<T>::N;
    constexpr int CG = 8 * E;
    PoolGather pg = {};
    if (pool) pg = *pool;
    const int C = d->C;
    if (C % E || d->ldx % E || lddy % E || lddx % E) IFCBK_FAIL(ctx, IFCBK_EINVAL, "bn_bwd: channels must be multiples of %d", E);
    const int64_t M = d->M;
    const int rows = bwd_rows(M, cdiv(C, CG));
    int ntiles = cdiv(M, rows);
    size_t need = ((size_t)ntiles * 2 * C + 2 * C) * sizeof(float);
    if (need > ctx->ws_bytes) IFCBK_FAIL(ctx, IFCBK_ENOMEM, "bn_bwd: workspace %zu > reserved %zu", need, ctx->ws_bytes);
    float* part = (float*)ctx->ws;
    float* tmp = part + (size_t)ntiles * 2 * C;
    const T* xx = (const T*)x; const T* yy = (const T*)y; const T* dd = (const T*)dy;
    dim3 g1(ntiles, cdiv(C, CG));
    // ReLU mask: recomputed from x when the caller hands over bn_apply's scale/shift and there is no residual
    const int mask = !d->relu ? 0 : ((scale && shift && !dres) ? 2 : 1);
    if (mask == 1 && !yy) IFCBK_FAIL(ctx, IFCBK_EINVAL, "bn_bwd: y is required for the ReLU mask");
    if (pool && pg.ph == 0 && pg.pw == 0) {
        if (mask == 1) IFCBK_FAIL(ctx, IFCBK_EINVAL, "bn_bwd_maxpool: needs scale/shift (the activation is not stored)");
        const int HB = (pg.H + 1) / 2, WB = (pg.W + 1) / 2;
        const int64_t nblk = (int64_t)(M / ((int64_t)pg.H * pg.W)) * HB * WB;
        const int64_t tot = nblk * (C / E);
        if (tot >= (1ll << 31) - 256) IFCBK_FAIL(ctx, IFCBK_EINVAL, "bn_bwd_maxpool: tensor too large");
        const int nt2 = cdiv(nblk, 256);
        if (((size_t)nt2 * 2 * C + 2 * C) * sizeof(float) > ctx->ws_bytes) IFCBK_FAIL(ctx, IFCBK_ENOMEM, "bn_bwd_maxpool: workspace");
        float* tmp2 = part + (size_t)nt2 * 2 * C;
        const fastdiv_t fHBWB = make_fastdiv(HB * WB), fWB = make_fastdiv(WB), fc2 = make_fastdiv(C / E);
        dim3 g2(nt2, cdiv(C, CG));
        if (mask == 2) hipLaunchKernelGGL((bn_bwd_reduce_pool2x2_kernel<T, 2>), g2, dim3(256), 0, st, xx, d->ldx, dd, mean, invstd, scale, shift, part, C, pg, HB, WB, (uint32_t)nblk, fHBWB, fWB);
        else hipLaunchKernelGGL((bn_bwd_reduce_pool2x2_kernel<T, 0>), g2, dim3(256), 0, st, xx, d->ldx, dd, mean, invstd, scale, shift, part, C, pg, HB, WB, (uint32_t)nblk, fHBWB, fWB);
        IFCBK_LAUNCH_CHECK(ctx, "bn_bwd_reduce_pool2x2");
        if (!experiment_skip_finalize(st)) hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 16)), dim3(1024), 0, st, (const float*)part, nt2, C, dgamma, dbeta, tmp2, param_accumulate, C);
        IFCBK_LAUNCH_CHECK(ctx, "bn_bwd_finalize");
        const float invM2 = (float)(1.0 / (double)M);
        const size_t shm2 = (size_t)5 * C * sizeof(float);
        if (mask == 2) hipLaunchKernelGGL((bn_bwd_dx_pool2x2_kernel<T, 2>), dim3(cdiv(tot, 256)), dim3(256), shm2, st, xx, d->ldx, dd, gamma, mean, invstd, scale, shift, (const float*)tmp2, (T*)dx, lddx, C, invM2, pg, HB, WB, (uint32_t)tot, fc2, fHBWB, fWB);
        else hipLaunchKernelGGL((bn_bwd_dx_pool2x2_kernel<T, 0>), dim3(cdiv(tot, 256)), dim3(256), shm2, st, xx, d->ldx, dd, gamma, mean, invstd, scale, shift, (const float*)tmp2, (T*)dx, lddx, C, invM2, pg, HB, WB, (uint32_t)tot, fc2, fHBWB, fWB);
        IFCBK_LAUNCH_CHECK(ctx, "bn_bwd_dx_pool2x2");
        return 0;
    }
    if (part_in) {
        // the partial sums were produced by the epilogue of the consumer's input-gradient kernel (ifcbk_conv2d_dgrad_bnstat)
        if (mask != 2 || dres) IFCBK_FAIL(ctx, IFCBK_EINVAL, "bn_bwd_partials: ReLU layer without residual, scale/shift required");
        part = const_cast<float*>(part_in);
        ntiles = ntiles_in;
        tmp = (float*)ctx->ws;
        if ((size_t)2 * C * sizeof(float) > ctx->ws_bytes) IFCBK_FAIL(ctx, IFCBK_ENOMEM, "bn_bwd_partials: workspace");
    } else if (pool) {
        if (mask == 1) IFCBK_FAIL(ctx, IFCBK_EINVAL, "bn_bwd_maxpool: needs scale/shift (the activation is not stored)");
        if (mask == 2) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 2, true>), g1, dim3(256), 0, st, xx, d->ldx, yy, d->ldy, dd, lddy, mean, invstd, scale, shift, part, M, C, pg, rows);
        else hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 0, true>), g1, dim3(256), 0, st, xx, d->ldx, yy, d->ldy, dd, lddy, mean, invstd, scale, shift, part, M, C, pg, rows);
    } else if (mask == 2) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 2, false>), g1, dim3(256), 0, st, xx, d->ldx, yy, d->ldy, dd, lddy, mean, invstd, scale, shift, part, M, C, pg, rows);
    else if (mask == 1) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 1, false>), g1, dim3(256), 0, st, xx, d->ldx, yy, d->ldy, dd, lddy, mean, invstd, scale, shift, part, M, C, pg, rows);
    else hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 0, false>), g1, dim3(256), 0, st, xx, d->ldx, yy, d->ldy, dd, lddy, mean, invstd, scale, shift, part, M, C, pg, rows);
    IFCBK_LAUNCH_CHECK(ctx, "bn_bwd_reduce");
    if (!experiment_skip_finalize(st)) hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 16)), dim3(1024), 0, st, (const float*)part, ntiles, C, dgamma, dbeta, tmp, param_accumulate, (part_in && part_ld_in > 0) ? part_ld_in : C);
    IFCBK_LAUNCH_CHECK(ctx, "bn_bwd_finalize");
    const int64_t total = M * (C / E);
    if (total >= (1ll << 31)) IFCBK_FAIL(ctx, IFCBK_EINVAL, "bn_bwd: tensor too large");
    const float invM = (float)(1.0 / (double)M);
    dim3 g3(cdiv(total, 256 * EW_ITER));
    const size_t shm = (size_t)5 * C * sizeof(float);
    const fastdiv_t fc = make_fastdiv(C / E);
    if (pool) {
        if (mask == 2) hipLaunchKernelGGL((bn_bwd_dx_kernel<T, 2, true>), g3, dim3(256), shm, st, xx, d->ldx, yy, d->ldy, dd, lddy, gamma, mean, invstd, scale, shift, (const float*)tmp, (T*)dx, lddx, (T*)dres, lddres, dres_accumulate, M, C, invM, fc, pg);
        else hipLaunchKernelGGL((bn_bwd_dx_kernel<T, 0, true>), g3, dim3(256), shm, st, xx, d->ldx, yy, d->ldy, dd, lddy, gamma, mean, invstd, scale, shift, (const float*)tmp, (T*)dx, lddx, (T*)dres, lddres, dres_accumulate, M, C, invM, fc, pg);
    } else if (mask == 2) hipLaunchKernelGGL((bn_bwd_dx_kernel<T, 2, false>), g3, dim3(256), shm, st, xx, d->ldx, yy, d->ldy, dd, lddy, gamma, mean, invstd, scale, shift, (const float*)tmp, (T*)dx, lddx, (T*)dres, lddres, dres_accumulate, M, C, invM, fc, pg);
    else if (mask == 1) hipLaunchKernelGGL((bn_bwd_dx_kernel<T, 1, false>), g3, dim3(256), shm, st, xx, d->ldx, yy, d->ldy, dd, lddy, gamma, mean, invstd, scale, shift, (const float*)tmp, (T*)dx, lddx, (T*)dres, lddres, dres_accumulate, M, C, invM, fc, pg);
    else hipLaunchKernelGGL((bn_bwd_dx_kernel<T, 0, false>), g3, dim3(256), shm, st, xx, d->ldx, yy, d->ldy, dd, lddy, gamma, mean, invstd, scale, shift, (const float*)tmp, (T*)dx, lddx, (T*)dres, lddres, dres_accumulate, M, C, invM, fc, pg);
    IFCBK_LAUNCH_CHECK(ctx, "bn_bwd_dx");
    return 0;
}

}  // namespace

extern "C" int ifcbk_bn_finalize(ifcbk_ctx* ctx, const ifcbk_bn_desc* d, const float* part, int mblocks,
                                 const float* gamma, const float* beta, float* running_mean, float* running_var,
                                 float* mean, float* invstd, float* scale, float* shift, void* stream) {
    return ifcbk_bn_finalize_ld(ctx, d, part, mblocks, 0, gamma, beta, running_mean, running_var, mean, invstd, scale, shift, stream);
}

extern "C" int ifcbk_bn_finalize_ld(ifcbk_ctx* ctx, const ifcbk_bn_desc* d, const float* part, int mblocks, int part_ld,
                                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                                    float* mean, float* invstd, float* scale, float* shift, void* stream) {
    if (!d || d->C <= 0) IFCBK_FAIL(ctx, IFCBK_EINVAL, "bn_finalize: bad desc");
    hipStream_t st = (hipStream_t)stream;
    int ldp = part_ld > 0 ? part_ld : d->C;
    if (part) {
        double M = (double)d->M;
        double unbias = d->M > 1 ? M / (M - 1.0) : 1.0;
        if (mblocks > 1536) {
            // two-stage: 64-row chunks are pre-reduced by many blocks into the ctx workspace
            const int rpc = 64;
            int nchunk = cdiv(mblocks, rpc);
            size_t need = (size_t)nchunk * 2 * d->C * sizeof(float);
            if (need > ctx->ws_bytes) IFCBK_FAIL(ctx, IFCBK_ENOMEM, "bn_finalize: workspace %zu > reserved %zu", need, ctx->ws_bytes);
            hipLaunchKernelGGL(bn_prereduce_kernel, dim3(cdiv(d->C, 64), nchunk), dim3(256), 0, st, part, mblocks, ldp, d->C, rpc,
                               (float*)ctx->ws);
            IFCBK_LAUNCH_CHECK(ctx, "bn_prereduce");
            part = (const float*)ctx->ws;
            mblocks = nchunk;
            ldp = d->C;
        }
        if (!experiment_skip_finalize(st)) hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(d->C, 16)), dim3(1024), 0, st, part, mblocks, ldp, d->C, 1.0 / M,
                           unbias, gamma, beta, running_mean, running_var, mean, invstd, scale, shift, d->eps,
                           d->momentum);
    } else {
        hipLaunchKernelGGL(bn_eval_scale_kernel, dim3(cdiv(d->C, 256)), dim3(256), 0, st, d->C, gamma, beta,
                           (const float*)running_mean, (const float*)running_var, scale, shift, d->eps);
    }
    IFCBK_LAUNCH_CHECK(ctx, "bn_finalize");
    return 0;
}

extern "C" int ifcbk_bn_stats_rows(int64_t M) { return (int)((M + STAT_ROWS - 1) / STAT_ROWS); }

extern "C" int ifcbk_bn_stats(ifcbk_ctx* ctx, const ifcbk_bn_desc* d, const void* x, float* part, void* stream) {
    if (!d || !x || !part) IFCBK_FAIL(ctx, IFCBK_EINVAL, "bn_stats: null operand");
    if (d->dtype != IFCBK_BF16 && d->dtype != IFCBK_F32) IFCBK_FAIL(ctx, IFCBK_EUNSUPPORTED, "bn_stats: dtype must be bf16 or f32");
    const int e = dtype_chunk(d->dtype);
    if (d->C % e || d->ldx % e || d->M <= 0) IFCBK_FAIL(ctx, IFCBK_EINVAL, "bn_stats: C=%d ldx=%d must be multiples of %d, M > 0", d->C, d->ldx, e);
    const dim3 grid((unsigned)ifcbk_bn_stats_rows(d->M), (unsigned)cdiv(d->C, 8 * e));
    if (d->dtype == IFCBK_F32) hipLaunchKernelGGL(bn_stats_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, d->ldx, part, d->M, d->C);
    else hipLaunchKernelGGL(bn_stats_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, d->ldx, part, d->M, d->C);
    IFCBK_LAUNCH_CHECK(ctx, "bn_stats");
    return 0;
}

extern "C" int ifcbk_bn_apply(ifcbk_ctx* ctx, const ifcbk_bn_desc* d, const void* x, const float* scale,
                              const float* shift, const void* residual, int ldr, void* y, void* stream) {
    if (!d) IFCBK_FAIL(ctx, IFCBK_EINVAL, "bn_apply: null desc");
    if (d->dtype == IFCBK_F32) return apply_t<float>(ctx, d, x, scale, shift, residual, ldr, y, (hipStream_t)stream);
    if (d->dtype == IFCBK_BF16) return apply_t<bf16_t>(ctx, d, x, scale, shift, residual, ldr, y, (hipStream_t)stream);
    IFCBK_FAIL(ctx, IFCBK_EINVAL, "bn_apply: bad dtype");
}

extern "C" int ifcbk_bn_bwd(ifcbk_ctx* ctx, const ifcbk_bn_desc* d, const void* x, const void* y, const void* dy,
                            int lddy, const float* gamma, const float* mean, const float* invstd, void* dx, int lddx,
                            void* dres, int lddres, int dres_accumulate, float* dgamma, float* dbeta,
                            int param_accumulate, const float* scale, const float* shift, void* stream) {
    if (!d) IFCBK_FAIL(ctx, IFCBK_EINVAL, "bn_bwd: null desc");
    if (d->dtype == IFCBK_F32)
        return bwd_t<float>(ctx, d, x, y, dy, lddy, gamma, mean, invstd, dx, lddx, dres, lddres, dres_accumulate, dgamma,
                            dbeta, param_accumulate, scale, shift, (hipStream_t)stream);
    if (d->dtype == IFCBK_BF16)
        return bwd_t<bf16_t>(ctx, d, x, y, dy, lddy, gamma, mean, invstd, dx, lddx, dres, lddres, dres_accumulate, dgamma,
                             dbeta, param_accumulate, scale, shift, (hipStream_t)stream);
    IFCBK_FAIL(ctx, IFCBK_EINVAL, "bn_bwd: bad dtype");
}

// ---------------------------------------------------------------- BN apply (+ReLU) fused into a 3x3 max pool
// y = maxpool3x3(act(x*scale+shift)); the activation is rounded to the storage type before the comparison, so values
// and arg-max are those of bn_apply followed by maxpool_fwd -- without writing and re-reading the activation.
namespace {
struct ApplyPoolArgs {
    int H, W, P, Q, cpr, ldx, ldy, sh, sw, ph, pw, relu;
    uint32_t total;
    fastdiv_t f_cpr, f_q, f_p;
};
template <class T>
__global__ __launch_bounds__(256) void bn_apply_maxpool_kernel(const T* x, const float* scale, const float* shift, T* y,
                                                               uint8_t* arg, ApplyPoolArgs a) {
    constexpr int E = Chunk<T>::N;
    extern __shared__ __attribute__((aligned(16))) float coef[];
    const int C = a.cpr * E;
    for (int c = threadIdx.x; c < C; c += 256) {
        coef[c] = scale[c];
        coef[C + c] = shift[c];
    }
    __syncthreads();
    const uint32_t i = xcd_remap(blockIdx.x, gridDim.x) * 256u + threadIdx.x;
    if (i >= a.total) return;
    const uint32_t pix = fdiv(i, a.f_cpr);
    const int c = (int)(i - pix * a.cpr) * E;
    const uint32_t t2 = fdiv(pix, a.f_q);
    const int q = (int)(pix - t2 * a.Q);
    const uint32_t n = fdiv(t2, a.f_p);
    const int p = (int)(t2 - n * a.P);
    const int h0 = p * a.sh - a.ph, w0 = q * a.sw - a.pw;
    float sc[E], sf[E], best[E];
    int bi[E];
#pragma unroll
    for (int j = 0; j < E; ++j) { sc[j] = coef[c + j]; sf[j] = coef[C + c + j]; best[j] = -INFINITY; bi[j] = 0; }
    // the nine taps are requested before the first is used (clamped addresses; taps in the padding are switched off by data):
    // with `if (inside) { load; compare }` bodies every tap was its own memory round trip (3.5 TB/s)
    typename Chunk<T>::raw_t rv[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int h = min(max(h0 + r, 0), a.H - 1), w = min(max(w0 + s, 0), a.W - 1);
            rv[r * 3 + s] = Chunk<T>::load_raw(x + ((int64_t)(n * a.H + h) * a.W + w) * a.ldx + c);
        }
    bool first = true;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int h = h0 + r, w = w0 + s;
            const bool in = h >= 0 && h < a.H && w >= 0 && w < a.W;
            float f[E];
            Chunk<T>::widen(rv[r * 3 + s], f);
#pragma unroll
            for (int j = 0; j < E; ++j) {
                float v = f[j] * sc[j] + sf[j];
                if (a.relu) v = fmaxf(v, 0.f);
                v = Chunk<T>::round(v);
                if (in && (first || v > best[j] || v != v)) { best[j] = v; bi[j] = r * 3 + s; }
            }
            first = first && !in;
        }
    Chunk<T>::store(y + (int64_t)pix * a.ldy + c, best);
    if (arg) ArgPack<E>::store(arg + (int64_t)pix * C + c, bi);
}

int pooled_check(ifcbk_ctx* ctx, const ifcbk_pool_desc* d, const char* who) {
    if (!d || (d->dtype != IFCBK_BF16 && d->dtype != IFCBK_F32)) IFCBK_FAIL(ctx, IFCBK_EINVAL, "%s: bad desc", who);
    const int e = dtype_chunk(d->dtype);
    if (d->C % e || d->ldx % e || d->ldy % e) IFCBK_FAIL(ctx, IFCBK_EINVAL, "%s: channels must be multiples of %d", who, e);
    if (d->R != 3 || d->S != 3 || d->stride_h != 2 || d->stride_w != 2 || d->pad_h > 1 || d->pad_w > 1 || d->pad_h < 0 || d->pad_w < 0)
        IFCBK_FAIL(ctx, IFCBK_EUNSUPPORTED, "%s: only 3x3 / stride 2 / pad <= 1 max pools are fused", who);
    if ((int64_t)d->N * d->H * d->W * (d->C / e) >= (1ll << 31) - 256) IFCBK_FAIL(ctx, IFCBK_EINVAL, "%s: tensor too large", who);
    return 0;
}
}  // namespace

extern "C" int ifcbk_bn_apply_maxpool(ifcbk_ctx* ctx, const ifcbk_pool_desc* d, const void* x, const float* scale,
                                      const float* shift, int relu, void* y, uint8_t* argmax, void* stream) {
    if (int e = pooled_check(ctx, d, "bn_apply_maxpool")) return e;
    ApplyPoolArgs a;
    a.H = d->H; a.W = d->W; a.P = d->P; a.Q = d->Q; a.ldx = d->ldx; a.ldy = d->ldy;
    a.sh = 2; a.sw = 2; a.ph = d->pad_h; a.pw = d->pad_w; a.relu = relu;
    a.cpr = d->C / dtype_chunk(d->dtype);
    a.total = (uint32_t)((int64_t)d->N * d->P * d->Q * a.cpr);
    a.f_cpr = make_fastdiv(a.cpr); a.f_q = make_fastdiv(d->Q); a.f_p = make_fastdiv(d->P);
    if (a.total == 0) return IFCBK_OK;
    const size_t shm = (size_t)2 * d->C * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == IFCBK_F32) hipLaunchKernelGGL(bn_apply_maxpool_kernel<float>, dim3(cdiv(a.total, 256)), dim3(256), shm, st, (const float*)x, scale, shift, (float*)y, argmax, a);
    else hipLaunchKernelGGL(bn_apply_maxpool_kernel<bf16_t>, dim3(cdiv(a.total, 256)), dim3(256), shm, st, (const bf16_t*)x, scale, shift, (bf16_t*)y, argmax, a);
    IFCBK_LAUNCH_CHECK(ctx, "bn_apply_maxpool");
    return 0;
}

extern "C" int ifcbk_bn_bwd_partials_ld(ifcbk_ctx* ctx, const ifcbk_bn_desc* d, const void* x, const void* dy, int lddy,
                                        const float* gamma, const float* mean, const float* invstd, const float* scale,
                                        const float* shift, const float* part, int ntiles, int part_ld, void* dx, int lddx,
                                        float* dgamma, float* dbeta, int param_accumulate, void* stream) {
    if (!d || !part || ntiles <= 0 || part_ld < 0) IFCBK_FAIL(ctx, IFCBK_EINVAL, "bn_bwd_partials: bad args");
    if (d->dtype == IFCBK_F32)
        return bwd_t<float>(ctx, d, x, nullptr, dy, lddy, gamma, mean, invstd, dx, lddx, nullptr, 0, 0, dgamma, dbeta,
                            param_accumulate, scale, shift, (hipStream_t)stream, nullptr, part, ntiles, part_ld);
    if (d->dtype == IFCBK_BF16)
        return bwd_t<bf16_t>(ctx, d, x, nullptr, dy, lddy, gamma, mean, invstd, dx, lddx, nullptr, 0, 0, dgamma, dbeta,
                             param_accumulate, scale, shift, (hipStream_t)stream, nullptr, part, ntiles, part_ld);
    IFCBK_FAIL(ctx, IFCBK_EINVAL, "bn_bwd_partials: bad dtype");
}

extern "C" int ifcbk_bn_bwd_partials(ifcbk_ctx* ctx, const ifcbk_bn_desc* d, const void* x, const void* dy, int lddy,
                                     const float* gamma, const float* mean, const float* invstd, const float* scale,
                                     const float* shift, const float* part, int ntiles, void* dx, int lddx, float* dgamma,
                                     float* dbeta, int param_accumulate, void* stream) {
    return ifcbk_bn_bwd_partials_ld(ctx, d, x, dy, lddy, gamma, mean, invstd, scale, shift, part, ntiles, 0, dx, lddx, dgamma, dbeta,
                                    param_accumulate, stream);
}

extern "C" int ifcbk_bn_bwd_maxpool(ifcbk_ctx* ctx, const ifcbk_pool_desc* d, const void* x, const void* dpool,
                                    const uint8_t* argmax, const float* gamma, const float* mean, const float* invstd,
                                    const float* scale, const float* shift, int relu, void* dx, int lddx, float* dgamma,
                                    float* dbeta, int param_accumulate, void* stream) {
    if (int e = pooled_check(ctx, d, "bn_bwd_maxpool")) return e;
    if (!argmax || !dpool) IFCBK_FAIL(ctx, IFCBK_EINVAL, "bn_bwd_maxpool: pooled gradient and arg-max required");
    ifcbk_bn_desc b;
    b.M = (int64_t)d->N * d->H * d->W; b.C = d->C; b.ldx = d->ldx; b.ldy = d->ldx; b.relu = relu; b.dtype = d->dtype;
    b.eps = 0.f; b.momentum = 0.f;
    PoolGather pg;
    pg.arg = argmax; pg.H = d->H; pg.W = d->W; pg.P = d->P; pg.Q = d->Q; pg.ph = d->pad_h; pg.pw = d->pad_w;
    pg.ldp = d->ldy; pg.C = d->C;
    pg.fHW = make_fastdiv(d->H * d->W); pg.fW = make_fastdiv(d->W);
    if (d->dtype == IFCBK_F32)
        return bwd_t<float>(ctx, &b, x, nullptr, dpool, 0, gamma, mean, invstd, dx, lddx, nullptr, 0, 0, dgamma, dbeta,
                            param_accumulate, scale, shift, (hipStream_t)stream, &pg);
    return bwd_t<bf16_t>(ctx, &b, x, nullptr, dpool, 0, gamma, mean, invstd, dx, lddx, nullptr, 0, 0, dgamma, dbeta,
                         param_accumulate, scale, shift, (hipStream_t)stream, &pg);
}
