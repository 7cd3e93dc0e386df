// Pooling, GAP->dropout->FC head, softmax / cross-entropy, Adam/SGD, layout conversion.  HBM-bound kernels:
// NHWC bf16, one 16-byte chunk (8 channels) per lane, consecutive lanes on consecutive chunks.
#include "common.h"
#include <math.h>
#include <stdlib.h>

namespace {

struct PoolArgs {
    int H, W, C, ldx, R, S, sh, sw, ph, pw, P, Q, ldy;
    int64_t total;   // work items
    int cpr;
};

PoolArgs make_pool(const ifcbk_pool_desc* d, bool bwd) {
    PoolArgs a;
    a.H = d->H; a.W = d->W; a.C = d->C; a.ldx = d->ldx; a.R = d->R; a.S = d->S;
    a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w; a.P = d->P; a.Q = d->Q; a.ldy = d->ldy;
    a.cpr = d->C / dtype_chunk(d->dtype);
    a.total = (int64_t)d->N * (bwd ? (int64_t)d->H * d->W : (int64_t)d->P * d->Q) * a.cpr;
    return a;
}

template <class T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* x, T* y, uint8_t* arg, PoolArgs a) {
    constexpr int E = Chunk<T>::N;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.total) return;
    int c = (int)(i % a.cpr) * E;
    int64_t pix = i / a.cpr;
    int q = (int)(pix % a.Q);
    int64_t t2 = pix / a.Q;
    int p = (int)(t2 % a.P);
    int64_t n = t2 / a.P;
    float best[E];
    int bi[E];
#pragma unroll
    for (int j = 0; j < E; ++j) { best[j] = -INFINITY; bi[j] = 0; }
    bool first = true;
    for (int r = 0; r < a.R; ++r) {
        int h = p * a.sh - a.ph + r;
        if (h < 0 || h >= a.H) continue;
        for (int s = 0; s < a.S; ++s) {
            int w = q * a.sw - a.pw + s;
            if (w < 0 || w >= a.W) continue;
            float f[E];
            Chunk<T>::load(x + ((n * a.H + h) * a.W + w) * a.ldx + c, f);
#pragma unroll
            for (int j = 0; j < E; ++j)
                if (first || f[j] > best[j] || f[j] != f[j]) { best[j] = f[j]; bi[j] = r * a.S + s; }
            first = false;
        }
    }
    Chunk<T>::store(y + pix * a.ldy + c, best);
    if (arg) ArgPack<E>::store(arg + pix * a.C + c, bi);
}

template <class T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* dy, const uint8_t* arg, T* dx, PoolArgs a, int accumulate) {
    constexpr int E = Chunk<T>::N;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.total) return;
    int c = (int)(i % a.cpr) * E;
    int64_t pix = i / a.cpr;
    int w = (int)(pix % a.W);
    int64_t t2 = pix / a.W;
    int h = (int)(t2 % a.H);
    int64_t n = t2 / a.H;
    float g[E];
#pragma unroll
    for (int j = 0; j < E; ++j) g[j] = 0.f;
    int plo = h + a.ph - a.R + 1; plo = plo <= 0 ? 0 : (plo + a.sh - 1) / a.sh;
    int phi = (h + a.ph) / a.sh; if (phi >= a.P) phi = a.P - 1;
    int qlo = w + a.pw - a.S + 1; qlo = qlo <= 0 ? 0 : (qlo + a.sw - 1) / a.sw;
    int qhi = (w + a.pw) / a.sw; if (qhi >= a.Q) qhi = a.Q - 1;
    for (int p = plo; p <= phi; ++p)
        for (int q = qlo; q <= qhi; ++q) {
            int want = (h - (p * a.sh - a.ph)) * a.S + (w - (q * a.sw - a.pw));
            int64_t opix = (n * a.P + p) * a.Q + q;
            int idx[E];
            ArgPack<E>::load(arg + opix * a.C + c, idx);
            float f[E];
            Chunk<T>::load(dy + opix * a.ldy + c, f);
#pragma unroll
            for (int j = 0; j < E; ++j)
                if (idx[j] == want) g[j] += f[j];
        }
    T* dp = dx + pix * a.ldx + c;
    if (accumulate) {
        float o[E];
        Chunk<T>::load(dp, o);
#pragma unroll
        for (int j = 0; j < E; ++j) g[j] += o[j];
    }
    Chunk<T>::store(dp, g);
}

template <class T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* x, T* y, PoolArgs a) {
    constexpr int E = Chunk<T>::N;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.total) return;
    int c = (int)(i % a.cpr) * E;
    int64_t pix = i / a.cpr;
    int q = (int)(pix % a.Q);
    int64_t t2 = pix / a.Q;
    int p = (int)(t2 % a.P);
    int64_t n = t2 / a.P;
    float acc[E];
#pragma unroll
    for (int j = 0; j < E; ++j) acc[j] = 0.f;
    for (int r = 0; r < a.R; ++r) {
        int h = p * a.sh - a.ph + r;
        if (h < 0 || h >= a.H) continue;
        for (int s = 0; s < a.S; ++s) {
            int w = q * a.sw - a.pw + s;
            if (w < 0 || w >= a.W) continue;
            float f[E];
            Chunk<T>::load(x + ((n * a.H + h) * a.W + w) * a.ldx + c, f);
#pragma unroll
            for (int j = 0; j < E; ++j) acc[j] += f[j];
        }
    }
    const float inv = 1.f / (float)(a.R * a.S);     // count_include_pad=True
#pragma unroll
    for (int j = 0; j < E; ++j) acc[j] *= inv;
    Chunk<T>::store(y + pix * a.ldy + c, acc);
}

template <class T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* dy, T* dx, PoolArgs a, int accumulate) {
    constexpr int E = Chunk<T>::N;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.total) return;
    int c = (int)(i % a.cpr) * E;
    int64_t pix = i / a.cpr;
    int w = (int)(pix % a.W);
    int64_t t2 = pix / a.W;
    int h = (int)(t2 % a.H);
    int64_t n = t2 / a.H;
    float g[E];
#pragma unroll
    for (int j = 0; j < E; ++j) g[j] = 0.f;
    int plo = h + a.ph - a.R + 1; plo = plo <= 0 ? 0 : (plo + a.sh - 1) / a.sh;
    int phi = (h + a.ph) / a.sh; if (phi >= a.P) phi = a.P - 1;
    int qlo = w + a.pw - a.S + 1; qlo = qlo <= 0 ? 0 : (qlo + a.sw - 1) / a.sw;
    int qhi = (w + a.pw) / a.sw; if (qhi >= a.Q) qhi = a.Q - 1;
    for (int p = plo; p <= phi; ++p)
        for (int q = qlo; q <= qhi; ++q) {
            float f[E];
            Chunk<T>::load(dy + ((n * a.P + p) * a.Q + q) * a.ldy + c, f);
#pragma unroll
            for (int j = 0; j < E; ++j) g[j] += f[j];
        }
    const float inv = 1.f / (float)(a.R * a.S);
#pragma unroll
    for (int j = 0; j < E; ++j) g[j] *= inv;
    T* dp = dx + pix * a.ldx + c;
    if (accumulate) {
        float o[E];
        Chunk<T>::load(dp, o);
#pragma unroll
        for (int j = 0; j < E; ++j) g[j] += o[j];
    }
    Chunk<T>::store(dp, g);
}

// ---------------------------------------------------------------- 3x3 fast paths (every pool of inception_v3 / resnet)
// 32-bit work index decoded with multiply-high "fastdiv", all taps loaded before the first use, and the block id
// remapped so that one XCD (one L2) owns a contiguous run of rows -- vertically adjacent outputs re-read the same
// input rows, which otherwise are fetched once per XCD (rocprof FETCH_SIZE was ~4x the tensor).
struct Pool3Args {
    int H, W, P, Q, cpr, ldx, ldy, sh, sw, ph, pw, nstrip, remap;
    uint32_t total;
    fastdiv_t f_cpr, f_a, f_b;      // chunk, then (strip | q | w), then (h | p)
};

// 3x3 / stride 1 / pad 1 average, count_include_pad: forward AND backward (the operator is symmetric; backward passes
// accumulate=1 when dx already holds another branch's gradient).  A thread owns 4 adjacent output columns of one 16-byte
// channel chunk and walks down up to AVG_SEG output rows: every input row is loaded once (6 chunks), reduced to 4
// horizontal 3-sums, and used by three output rows from registers -- 1.9 loads per output instead of 9 (generic
// kernel) or 4.5 (first fast path); these kernels are bound by L2 requests, not HBM.
constexpr int AVG_SEG = 8;
template <class T>
__global__ __launch_bounds__(256) void avgpool3x3s1_kernel(const T* in, T* out, Pool3Args a, int accumulate,
                                                           const float* ep_scale = nullptr, const float* ep_shift = nullptr, int ep_relu = 0) {
    constexpr int E = Chunk<T>::N;
    constexpr int TW = 4;
    const uint32_t i = (a.remap ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x) * 256u + threadIdx.x;
    if (i >= a.total) return;
    const uint32_t t = fdiv(i, a.f_cpr);
    const int c = (int)(i - t * a.cpr) * E;
    const uint32_t t2 = fdiv(t, a.f_a);
    const int w0 = (int)(t - t2 * a.nstrip) * TW;
    const uint32_t n = fdiv(t2, a.f_b);                       // f_b: row segments per image
    const int hbeg = (int)(t2 - n * a.f_b.d) * AVG_SEG;
    const int hend = min(hbeg + AVG_SEG, a.H);                // output rows [hbeg, hend)
    const float inv = 1.f / 9.f;
    // optional epilogue y = act(round(avg) * scale[c] + shift[c]): the eval-mode BatchNorm of a pool branch that runs as
    // avgpool(conv1x1(x)); the average is rounded to the storage type first, like the pooled tensor training stores
    float esc[E], esh[E];
    if (ep_scale) {
#pragma unroll
        for (int j = 0; j < E; ++j) { esc[j] = ep_scale[c + j]; esh[j] = ep_shift[c + j]; }
    }
    float p2[TW][E], p1[TW][E];                               // horizontal 3-sums of input rows hh-2, hh-1
#pragma unroll
    for (int k = 0; k < TW; ++k)
#pragma unroll
        for (int j = 0; j < E; ++j) p2[k][j] = p1[k][j] = 0.f;
    // the six chunks of input row hh+1 are requested before row hh is reduced and stored: one memory latency per row is
    // overlapped with the previous row's math and stores instead of being exposed (the kernel holds only ~3 waves per SIMD)
    typedef typename Chunk<T>::raw_t raw_t;
    raw_t nxt[TW + 2];
    auto fetch = [&](int hh) {
        const bool rv = hh >= 0 && hh < a.H;
        const T* row = in + ((int64_t)(n * a.H + (rv ? hh : 0)) * a.W) * a.ldx + c;
#pragma unroll
        for (int k = 0; k < TW + 2; ++k) {
            const int ww = w0 - 1 + k;
            if (rv && ww >= 0 && ww < a.W) nxt[k] = Chunk<T>::load_raw(row + (int64_t)ww * a.ldx);
            else nxt[k] = raw_t{};
        }
    };
    fetch(hbeg - 1);
    for (int hh = hbeg - 1; hh <= hend; ++hh) {
        float cur[TW][E];
        raw_t now[TW + 2];
#pragma unroll
        for (int k = 0; k < TW + 2; ++k) now[k] = nxt[k];
        if (hh < hend) fetch(hh + 1);
        {
            float v[TW + 2][E];
#pragma unroll
            for (int k = 0; k < TW + 2; ++k) Chunk<T>::widen(now[k], v[k]);
#pragma unroll
            for (int k = 0; k < TW; ++k)
#pragma unroll
                for (int j = 0; j < E; ++j) cur[k][j] = (v[k][j] + v[k + 1][j]) + v[k + 2][j];
        }
        if (hh >= hbeg + 1) {
            T* orow = out + ((int64_t)(n * a.H + (hh - 1)) * a.W) * a.ldy + c;
#pragma unroll
            for (int k = 0; k < TW; ++k) {
                const int ww = w0 + k;
                if (ww >= a.W) break;
                float o[E];
#pragma unroll
                for (int j = 0; j < E; ++j) o[j] = ((p2[k][j] + p1[k][j]) + cur[k][j]) * inv;
                if (ep_scale) {
#pragma unroll
                    for (int j = 0; j < E; ++j) {
                        o[j] = Chunk<T>::round(o[j]) * esc[j] + esh[j];
                        if (ep_relu) o[j] = fmaxf(o[j], 0.f);
                    }
                }
                T* op = orow + (int64_t)ww * a.ldy;
                if (accumulate) {
                    float g[E];
                    Chunk<T>::load(op, g);
#pragma unroll
                    for (int j = 0; j < E; ++j) o[j] += g[j];
                }
                Chunk<T>::store(op, o);
            }
        }
#pragma unroll
        for (int k = 0; k < TW; ++k)
#pragma unroll
            for (int j = 0; j < E; ++j) { p2[k][j] = p1[k][j]; p1[k][j] = cur[k][j]; }
    }
}

// 3x3 max, any stride / padding: same result and first-max tie rule as maxpool_fwd_kernel
template <class T>
__global__ __launch_bounds__(256) void maxpool3x3_fwd_kernel(const T* x, T* y, uint8_t* arg, Pool3Args a) {
    constexpr int E = Chunk<T>::N;
    const uint32_t i = (a.remap ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x) * 256u + threadIdx.x;
    if (i >= a.total) return;
    const uint32_t pix = fdiv(i, a.f_cpr);
    const int c = (int)(i - pix * a.cpr) * E;
    const uint32_t t2 = fdiv(pix, a.f_a);
    const int q = (int)(pix - t2 * a.Q);
    const uint32_t n = fdiv(t2, a.f_b);
    const int p = (int)(t2 - n * a.P);
    const int h0 = p * a.sh - a.ph, w0 = q * a.sw - a.pw;
    float f[9][E];
    bool ok[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int h = h0 + r, w = w0 + s;
            ok[r * 3 + s] = h >= 0 && h < a.H && w >= 0 && w < a.W;
            if (ok[r * 3 + s]) Chunk<T>::load(x + ((int64_t)(n * a.H + h) * a.W + w) * a.ldx + c, f[r * 3 + s]);
        }
    float best[E];
    int bi[E];
#pragma unroll
    for (int j = 0; j < E; ++j) { best[j] = -INFINITY; bi[j] = 0; }
    bool first = true;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        if (!ok[k]) continue;
#pragma unroll
        for (int j = 0; j < E; ++j)
            if (first || f[k][j] > best[j] || f[k][j] != f[k][j]) { best[j] = f[k][j]; bi[j] = k; }
        first = false;
    }
    Chunk<T>::store(y + (int64_t)pix * a.ldy + c, best);
    if (arg) ArgPack<E>::store(arg + (int64_t)pix * (a.cpr * E) + c, bi);
}

// backward of a 3x3 / stride 2 max pool in gather form: an input pixel lies in at most 2x2 windows
template <class T>
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_kernel(const T* dy, const uint8_t* arg, T* dx, Pool3Args a, int accumulate) {
    constexpr int E = Chunk<T>::N;
    const uint32_t i = (a.remap ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x) * 256u + threadIdx.x;
    if (i >= a.total) return;
    const uint32_t pix = fdiv(i, a.f_cpr);
    const int c = (int)(i - pix * a.cpr) * E;
    const uint32_t t2 = fdiv(pix, a.f_a);
    const int w = (int)(pix - t2 * a.W);
    const uint32_t n = fdiv(t2, a.f_b);
    const int h = (int)(t2 - n * a.H);
    const int C = a.cpr * E;
    int plo = h + a.ph - 2; plo = plo <= 0 ? 0 : (plo + 1) >> 1;
    int phi = (h + a.ph) >> 1; if (phi >= a.P) phi = a.P - 1;
    int qlo = w + a.pw - 2; qlo = qlo <= 0 ? 0 : (qlo + 1) >> 1;
    int qhi = (w + a.pw) >> 1; if (qhi >= a.Q) qhi = a.Q - 1;
    float f[4][E];
    int idx[4][E];
    int want[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int p = plo + u, q = qlo + v, k = u * 2 + v;
            ok[k] = p <= phi && q <= qhi;
            want[k] = (h - (p * 2 - a.ph)) * 3 + (w - (q * 2 - a.pw));
            if (ok[k]) {
                const int64_t opix = (int64_t)(n * a.P + p) * a.Q + q;
                ArgPack<E>::load(arg + opix * C + c, idx[k]);
                Chunk<T>::load(dy + opix * a.ldy + c, f[k]);
            }
        }
    float g[E];
#pragma unroll
    for (int j = 0; j < E; ++j) g[j] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!ok[k]) continue;
#pragma unroll
        for (int j = 0; j < E; ++j)
            if (idx[k][j] == want[k]) g[j] += f[k][j];
    }
    T* dp = dx + (int64_t)pix * a.ldx + c;
    if (accumulate) {
        float o[E];
        Chunk<T>::load(dp, o);
#pragma unroll
        for (int j = 0; j < E; ++j) g[j] += o[j];
    }
    Chunk<T>::store(dp, g);
}

// mode 0: avg 3x3s1p1 strips (fwd or bwd), 1: max fwd (items = outputs), 2: max bwd (items = inputs)
bool make_pool3(const ifcbk_pool_desc* d, int mode, Pool3Args* a) {
    // measured on MI355X (scripts/pool_bench.py): the strip average and the 2x2-gather max backward beat the generic
    // kernels by 1.1-1.7x; the hoisted 9-tap max forward does not (85 VGPRs, 3.3 vs 3.9 TB/s) and stays opt-in
    { const char* e = getenv("IFCBK_POOL_FAST"); const int m = e ? atoi(e) : 5; if (!((m >> mode) & 1)) return false; }
    if (d->R != 3 || d->S != 3) return false;
    if (mode == 0 && !(d->stride_h == 1 && d->stride_w == 1 && d->pad_h == 1 && d->pad_w == 1 && d->P == d->H && d->Q == d->W)) return false;
    if (mode == 2 && !(d->stride_h == 2 && d->stride_w == 2 && d->pad_h <= 1 && d->pad_w <= 1)) return false;
    a->H = d->H; a->W = d->W; a->P = d->P; a->Q = d->Q; a->ldx = d->ldx; a->ldy = d->ldy;
    a->sh = d->stride_h; a->sw = d->stride_w; a->ph = d->pad_h; a->pw = d->pad_w;
    a->cpr = d->C / dtype_chunk(d->dtype);
    a->nstrip = (d->W + 3) / 4;
    { const char* e = getenv("IFCBK_POOL_REMAP"); a->remap = e ? atoi(e) : 1; }
    const int inner = mode == 0 ? a->nstrip : (mode == 1 ? d->Q : d->W);
    const int outer = mode == 0 ? (d->H + AVG_SEG - 1) / AVG_SEG : (mode == 1 ? d->P : d->H);   // mode 0: row segments
    const int64_t total = (int64_t)d->N * outer * inner * a->cpr;
    if (total <= 0 || total >= (1ll << 31) - 256) return false;
    a->total = (uint32_t)total;
    a->f_cpr = make_fastdiv(a->cpr); a->f_a = make_fastdiv(inner); a->f_b = make_fastdiv(outer);
    return true;
}

int pool_check(ifcbk_ctx* ctx, const ifcbk_pool_desc* d) {
    if (!d || (d->dtype != IFCBK_BF16 && d->dtype != IFCBK_F32)) IFCBK_FAIL(ctx, IFCBK_EINVAL, "pool: bad desc");
    const int e = dtype_chunk(d->dtype);
    if (d->C % e || d->ldx % e || d->ldy % e) IFCBK_FAIL(ctx, IFCBK_EINVAL, "pool: channels must be multiples of %d", e);
    if (d->R * d->S > 255) IFCBK_FAIL(ctx, IFCBK_EINVAL, "pool: window too large");
    return 0;
}

// ---------------------------------------------------------------- head
// feat[n][c] = mean_hw x[n,hw,c] * (mask ? mask*keep_scale : 1)
// GAP_SPLIT threads share one (sample, 16-byte channel chunk): thread s sums pixels s, s+GAP_SPLIT, ... (4 loads in flight),
// the partial sums are combined in a fixed order -- a thread per chunk walking all 64 pixels alone left the 67 MB read of
// the 8x8x2048 head at 0.45 TB/s (one wave per SIMD, one load in flight)
constexpr int GAP_SPLIT = 8;
template <class T>
__global__ __launch_bounds__(256) void gap_kernel(const T* x, int ldx, int HW, int C, int64_t total, const uint8_t* mask,
                                                  float keep_scale, float* feat) {
    constexpr int E = Chunk<T>::N;
    const int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int sub = (int)(gi & (GAP_SPLIT - 1));
    int64_t i = gi / GAP_SPLIT;
    const bool live = i < total;
    if (!live) i = total - 1;                       // keep the whole wave in the shuffles
    int cpr = C / E;
    int c = (int)(i % cpr) * E;
    int64_t n = i / cpr;
    float acc[E];
#pragma unroll
    for (int j = 0; j < E; ++j) acc[j] = 0.f;
    const T* px = x + n * HW * (int64_t)ldx + c;
    int hw = sub;
    for (; hw + 3 * GAP_SPLIT < HW; hw += 4 * GAP_SPLIT) {
        typename Chunk<T>::raw_t r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = Chunk<T>::load_raw(px + (int64_t)(hw + u * GAP_SPLIT) * ldx);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float f[E];
            Chunk<T>::widen(r[u], f);
#pragma unroll
            for (int j = 0; j < E; ++j) acc[j] += f[j];
        }
    }
    for (; hw < HW; hw += GAP_SPLIT) {
        float f[E];
        Chunk<T>::load(px + (int64_t)hw * ldx, f);
#pragma unroll
        for (int j = 0; j < E; ++j) acc[j] += f[j];
    }
#pragma unroll
    for (int off = 1; off < GAP_SPLIT; off <<= 1)
#pragma unroll
        for (int j = 0; j < E; ++j) acc[j] += __shfl_xor(acc[j], off);
    if (!live || sub) return;
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        float v = acc[j] * inv;
        if (mask) v *= mask[n * C + c + j] ? keep_scale : 0.f;
        feat[n * C + c + j] = v;
    }
}

// logits[n][j] = feat[n] . W[j] + b[j].  Block = FC_NB samples x one quarter of the classes: the FC_NB feature rows
// sit in LDS, each wave owns a class and streams its weight row ONCE for all FC_NB samples (fixed summation order:
// lane-strided partial sums, then a butterfly -- independent of the grid).
// (FC_NB = 8 samples per block up to 2048 features, 4 up to 4096 -- the alexnet / vgg classifier -- for the 64 KiB of LDS)
template <int FC_NB>
__global__ __launch_bounds__(256) void fc_fwd_kernel(const float* feat, const float* W, const float* b, float* logits,
                                                     int N, int C, int NC) {
    extern __shared__ float sf[];                 // [FC_NB][C]
    const int n0 = blockIdx.x * FC_NB, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int nb = N - n0 < FC_NB ? N - n0 : FC_NB;
    for (int i = t; i < FC_NB * C; i += 256) {
        const int r = i / C;
        sf[i] = r < nb ? feat[(size_t)n0 * C + i] : 0.f;
    }
    __syncthreads();
    for (int j = blockIdx.y * 4 + wave; j < NC; j += 4 * gridDim.y) {
        const float* w = W + (size_t)j * C;
        float s[FC_NB];
#pragma unroll
        for (int r = 0; r < FC_NB; ++r) s[r] = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float wv = w[c];
#pragma unroll
            for (int r = 0; r < FC_NB; ++r) s[r] += sf[r * C + c] * wv;
        }
#pragma unroll
        for (int r = 0; r < FC_NB; ++r) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s[r] += __shfl_xor(s[r], off);
        }
        if (lane == 0) {
            const float bj = b[j];
#pragma unroll
            for (int r = 0; r < FC_NB; ++r)
                if (r < nb) logits[(size_t)(n0 + r) * NC + j] = s[r] + bj;
        }
    }
}

// dW[j][c] (+)= sum_n dl[n][j]*feat[n][c];   thread per (j,c)
__global__ __launch_bounds__(256) void fc_wgrad_kernel(const float* dl, const float* feat, float* dW, int N, int C, int NC,
                                                       int accumulate) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)NC * C) return;
    int c = (int)(i % C), j = (int)(i / C);
    float s = 0.f;
#pragma unroll 8
    for (int n = 0; n < N; ++n) s += dl[(size_t)n * NC + j] * feat[(size_t)n * C + c];      // 8 independent loads in flight, one fixed sum order
    dW[i] = accumulate ? dW[i] + s : s;
}
// one wave per class: lane-strided partial sums over the batch, then a fixed butterfly (a thread per class walking all N samples
// alone took 62 us at batch 256 -- at the one point of a step where nothing else can run: between the loss and the backward)
__global__ __launch_bounds__(64) void fc_bgrad_kernel(const float* dl, float* db, int N, int NC, int accumulate) {
    const int j = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int n = lane; n < N; n += 64) s += dl[(size_t)n * NC + j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) db[j] = accumulate ? db[j] + s : s;
}
// dx[n,hw,c] = (sum_j dl[n][j] W[j][c]) * mask*scale / HW ; thread per (n, 16-byte chunk of channels)
template <class T>
__global__ __launch_bounds__(256) void head_dx_kernel(const float* dl, const float* W, const uint8_t* mask, float keep_scale,
                                                      T* dx, int lddx, int HW, int C, int NC, int64_t total) {
    constexpr int E = Chunk<T>::N;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int cpr = C / E;
    int c = (int)(i % cpr) * E;
    int64_t n = i / cpr;
    float g[E];
#pragma unroll
    for (int j = 0; j < E; ++j) g[j] = 0.f;
    if (!W) {                                       // pooled-logits form: d(pooled channel c) = dl[n][c] for c < NC
#pragma unroll
        for (int j = 0; j < E; ++j) g[j] = c + j < NC ? dl[n * NC + c + j] : 0.f;
    } else {
#pragma unroll 4
        for (int k = 0; k < NC; ++k) {
            float d = dl[n * NC + k];
            const float* w = W + (size_t)k * C + c;
#pragma unroll
            for (int j = 0; j < E; ++j) g[j] += d * w[j];
        }
    }
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        float v = g[j] * inv;
        if (mask) v *= mask[n * C + c + j] ? keep_scale : 0.f;
        g[j] = v;
    }
    for (int hw = 0; hw < HW; ++hw) Chunk<T>::store(dx + (n * HW + hw) * lddx + c, g);
}

// counter-based Bernoulli mask (splitmix64 of seed, offset+i)
__global__ void dropout_mask_kernel(uint8_t* mask, int64_t n, float p, uint64_t seed, uint64_t offset) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (offset + (uint64_t)i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    float u = (float)(z >> 40) * (1.0f / 16777216.0f);
    mask[i] = u >= p ? 1 : 0;
}

// ---------------------------------------------------------------- loss (single block: deterministic mean)
// 4 lanes per sample (classes j = sub, sub+4, ...: adjacent lanes read adjacent logits), fixed butterfly for max and
// sum, per-sample losses summed in a fixed order by thread 0 -> bitwise reproducible.
__global__ __launch_bounds__(1024) void softmax_xent_kernel(const float* logits, const int64_t* target, int N, int NC,
                                                            float weight, float* loss_out, int loss_acc, float* dlogits) {
    __shared__ float sl[256];
    const int sub = threadIdx.x & 3, slot = threadIdx.x >> 2;
    const float invN = 1.f / (float)N;
    float local = 0.f;
    for (int n0 = 0; n0 < N; n0 += 256) {
        const int n = n0 + slot;
        const bool ok = n < N;
        const float* l = logits + (size_t)(ok ? n : 0) * NC;
        float mx = -INFINITY;
        for (int j = sub; j < NC; j += 4) mx = fmaxf(mx, l[j]);
        mx = fmaxf(mx, __shfl_xor(mx, 1));
        mx = fmaxf(mx, __shfl_xor(mx, 2));
        float s = 0.f;
        for (int j = sub; j < NC; j += 4) s += expf(l[j] - mx);
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        const int tg = ok ? (int)target[n] : 0;
        if (ok && sub == 0) local += mx + logf(s) - l[tg];
        if (ok && dlogits) {
            float* d = dlogits + (size_t)n * NC;
            const float is = 1.f / s;
            for (int j = sub; j < NC; j += 4) d[j] = weight * invN * (expf(l[j] - mx) * is - (j == tg ? 1.f : 0.f));
        }
    }
    if (sub == 0) sl[slot] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < 256; ++i) s += sl[i];
        s = s * invN * weight;
        loss_out[0] = loss_acc ? loss_out[0] + s : s;
    }
}
__global__ void softmax_kernel(const float* logits, int N, int NC, float* probs) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float* l = logits + (size_t)n * NC;
    float mx = -INFINITY;
    for (int j = 0; j < NC; ++j) mx = fmaxf(mx, l[j]);
    float s = 0.f;
    for (int j = 0; j < NC; ++j) s += expf(l[j] - mx);
    float is = 1.f / s;
    for (int j = 0; j < NC; ++j) probs[(size_t)n * NC + j] = expf(l[j] - mx) * is;
}

// ---------------------------------------------------------------- optimizers (flat)
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                                                   float b1, float b2, float eps, float wd, float bc1, float sbc2,
                                                   float gscale) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 4 <= n) {
        float4 pp = *reinterpret_cast<float4*>(p + i), gg = *reinterpret_cast<const float4*>(g + i);
        float4 mm = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
        float* P = &pp.x; float* G = &gg.x; float* Mm = &mm.x; float* V = &vv.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gr = G[j] * gscale + wd * P[j];
            Mm[j] = b1 * Mm[j] + (1.f - b1) * gr;
            V[j] = b2 * V[j] + (1.f - b2) * gr * gr;
            float denom = sqrtf(V[j]) / sbc2 + eps;
            P[j] -= (lr / bc1) * (Mm[j] / denom);
        }
        *reinterpret_cast<float4*>(p + i) = pp;
        *reinterpret_cast<float4*>(m + i) = mm;
        *reinterpret_cast<float4*>(v + i) = vv;
    } else {
        for (int64_t k = i; k < n; ++k) {
            float gr = g[k] * gscale + wd * p[k];
            m[k] = b1 * m[k] + (1.f - b1) * gr;
            v[k] = b2 * v[k] + (1.f - b2) * gr * gr;
            float denom = sqrtf(v[k]) / sbc2 + eps;
            p[k] -= (lr / bc1) * (m[k] / denom);
        }
    }
}
__global__ __launch_bounds__(256) void sgd_kernel(float* p, const float* g, float* mom, int64_t n, float lr, float mu,
                                                  float wd, float gscale) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gr = g[i] * gscale + wd * p[i];
    if (mom) {
        float b = mu * mom[i] + gr;
        mom[i] = b;
        gr = b;
    }
    p[i] -= lr * gr;
}

// ---------------------------------------------------------------- layout
template <class T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* x, int C, int64_t HW, int64_t total, int Cpad,
                                                           float s0, float s1, float s2, float t0, float t1, float t2,
                                                           T* y) {
    constexpr int E = Chunk<T>::N;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // pixel index n*HW + hw
    if (i >= total) return;
    int64_t n = i / HW, hw = i - n * HW;
    const float* src = x + n * C * HW + hw;
    float sc[3] = {s0, s1, s2}, sh[3] = {t0, t1, t2};
    for (int c0 = 0; c0 < Cpad; c0 += E) {
        float f[E];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            int c = c0 + j;
            float v = c < C ? src[c * HW] : 0.f;
            if (c < 3 && c < C) v = v * sc[c] + sh[c];
            f[j] = v;
        }
        Chunk<T>::store(y + i * Cpad + c0, f);
    }
}
template <class T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* x, int C, int64_t HW, int64_t total, int ldx, float* y) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over n*C*HW, hw fastest
    if (i >= total) return;
    int64_t hw = i % HW;
    int64_t nc = i / HW;
    int c = (int)(nc % C);
    int64_t n = nc / C;
    y[i] = to_f32(x[(n * HW + hw) * ldx + c]);
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int ifcbk_maxpool_fwd(ifcbk_ctx* ctx, const ifcbk_pool_desc* d, const void* x, void* y, uint8_t* argmax, void* stream) {
    if (int e = pool_check(ctx, d)) return e;
    Pool3Args f;
    if (make_pool3(d, 1, &f)) {
        if (d->dtype == IFCBK_F32) hipLaunchKernelGGL(maxpool3x3_fwd_kernel<float>, dim3(cdiv(f.total, 256)), dim3(256), 0, ST, (const float*)x, (float*)y, argmax, f);
        else hipLaunchKernelGGL(maxpool3x3_fwd_kernel<bf16_t>, dim3(cdiv(f.total, 256)), dim3(256), 0, ST, (const bf16_t*)x, (bf16_t*)y, argmax, f);
        IFCBK_LAUNCH_CHECK(ctx, "maxpool3x3_fwd");
        return 0;
    }
    PoolArgs a = make_pool(d, false);
    if (d->dtype == IFCBK_F32) hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(cdiv(a.total, 256)), dim3(256), 0, ST, (const float*)x, (float*)y, argmax, a);
    else hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(cdiv(a.total, 256)), dim3(256), 0, ST, (const bf16_t*)x, (bf16_t*)y, argmax, a);
    IFCBK_LAUNCH_CHECK(ctx, "maxpool_fwd");
    return 0;
}
extern "C" int ifcbk_maxpool_bwd(ifcbk_ctx* ctx, const ifcbk_pool_desc* d, const void* dy, const uint8_t* argmax, void* dx,
                                 int accumulate, void* stream) {
    if (int e = pool_check(ctx, d)) return e;
    Pool3Args f;
    if (make_pool3(d, 2, &f)) {
        if (d->dtype == IFCBK_F32) hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel<float>, dim3(cdiv(f.total, 256)), dim3(256), 0, ST, (const float*)dy, argmax, (float*)dx, f, accumulate);
        else hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel<bf16_t>, dim3(cdiv(f.total, 256)), dim3(256), 0, ST, (const bf16_t*)dy, argmax, (bf16_t*)dx, f, accumulate);
        IFCBK_LAUNCH_CHECK(ctx, "maxpool3x3s2_bwd");
        return 0;
    }
    PoolArgs a = make_pool(d, true);
    if (d->dtype == IFCBK_F32) hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(cdiv(a.total, 256)), dim3(256), 0, ST, (const float*)dy, argmax, (float*)dx, a, accumulate);
    else hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(cdiv(a.total, 256)), dim3(256), 0, ST, (const bf16_t*)dy, argmax, (bf16_t*)dx, a, accumulate);
    IFCBK_LAUNCH_CHECK(ctx, "maxpool_bwd");
    return 0;
}
extern "C" int ifcbk_avgpool_fwd(ifcbk_ctx* ctx, const ifcbk_pool_desc* d, const void* x, void* y, void* stream) {
    if (int e = pool_check(ctx, d)) return e;
    Pool3Args f;
    if (make_pool3(d, 0, &f)) {
        if (d->dtype == IFCBK_F32) hipLaunchKernelGGL(avgpool3x3s1_kernel<float>, dim3(cdiv(f.total, 256)), dim3(256), 0, ST, (const float*)x, (float*)y, f, 0);
        else hipLaunchKernelGGL(avgpool3x3s1_kernel<bf16_t>, dim3(cdiv(f.total, 256)), dim3(256), 0, ST, (const bf16_t*)x, (bf16_t*)y, f, 0);
        IFCBK_LAUNCH_CHECK(ctx, "avgpool3x3s1");
        return 0;
    }
    PoolArgs a = make_pool(d, false);
    if (d->dtype == IFCBK_F32) hipLaunchKernelGGL(avgpool_fwd_kernel<float>, dim3(cdiv(a.total, 256)), dim3(256), 0, ST, (const float*)x, (float*)y, a);
    else hipLaunchKernelGGL(avgpool_fwd_kernel<bf16_t>, dim3(cdiv(a.total, 256)), dim3(256), 0, ST, (const bf16_t*)x, (bf16_t*)y, a);
    IFCBK_LAUNCH_CHECK(ctx, "avgpool_fwd");
    return 0;
}
extern "C" int ifcbk_avgpool3x3_affine(ifcbk_ctx* ctx, const ifcbk_pool_desc* d, const void* x, const float* scale, const float* shift,
                                       int relu, void* y, void* stream) {
    if (int e = pool_check(ctx, d)) return e;
    if (!scale || !shift) IFCBK_FAIL(ctx, IFCBK_EINVAL, "avgpool3x3_affine: scale/shift required");
    Pool3Args f;
    if (!make_pool3(d, 0, &f)) IFCBK_FAIL(ctx, IFCBK_EUNSUPPORTED, "avgpool3x3_affine: 3x3 / stride 1 / pad 1 pools only");
    if (d->dtype == IFCBK_F32) hipLaunchKernelGGL(avgpool3x3s1_kernel<float>, dim3(cdiv(f.total, 256)), dim3(256), 0, ST, (const float*)x, (float*)y, f, 0, scale, shift, relu);
    else hipLaunchKernelGGL(avgpool3x3s1_kernel<bf16_t>, dim3(cdiv(f.total, 256)), dim3(256), 0, ST, (const bf16_t*)x, (bf16_t*)y, f, 0, scale, shift, relu);
    IFCBK_LAUNCH_CHECK(ctx, "avgpool3x3_affine");
    return 0;
}
extern "C" int ifcbk_avgpool_bwd(ifcbk_ctx* ctx, const ifcbk_pool_desc* d, const void* dy, void* dx, int accumulate, void* stream) {
    if (int e = pool_check(ctx, d)) return e;
    Pool3Args f;
    if (make_pool3(d, 0, &f)) {
        // symmetric operator: dx = avg3x3(dy); the roles of (ldx, ldy) swap
        const int t = f.ldx; f.ldx = f.ldy; f.ldy = t;
        if (d->dtype == IFCBK_F32) hipLaunchKernelGGL(avgpool3x3s1_kernel<float>, dim3(cdiv(f.total, 256)), dim3(256), 0, ST, (const float*)dy, (float*)dx, f, accumulate);
        else hipLaunchKernelGGL(avgpool3x3s1_kernel<bf16_t>, dim3(cdiv(f.total, 256)), dim3(256), 0, ST, (const bf16_t*)dy, (bf16_t*)dx, f, accumulate);
        IFCBK_LAUNCH_CHECK(ctx, "avgpool3x3s1(bwd)");
        return 0;
    }
    PoolArgs a = make_pool(d, true);
    if (d->dtype == IFCBK_F32) hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(cdiv(a.total, 256)), dim3(256), 0, ST, (const float*)dy, (float*)dx, a, accumulate);
    else hipLaunchKernelGGL(avgpool_bwd_kernel<bf16_t>, dim3(cdiv(a.total, 256)), dim3(256), 0, ST, (const bf16_t*)dy, (bf16_t*)dx, a, accumulate);
    IFCBK_LAUNCH_CHECK(ctx, "avgpool_bwd");
    return 0;
}

extern "C" int ifcbk_head_fwd(ifcbk_ctx* ctx, const ifcbk_head_desc* d, const void* x, const uint8_t* mask, const float* W,
                              const float* b, float* feat, float* logits, void* stream) {
    if (!d || (d->dtype != IFCBK_BF16 && d->dtype != IFCBK_F32)) IFCBK_FAIL(ctx, IFCBK_EINVAL, "head_fwd: bad desc");
    const int e = dtype_chunk(d->dtype);
    if (d->C % e || d->ldx % e) IFCBK_FAIL(ctx, IFCBK_EINVAL, "head_fwd: channels must be multiples of %d", e);
    if (W && (size_t)d->C * 4 * 4 > 64 * 1024) IFCBK_FAIL(ctx, IFCBK_EINVAL, "head_fwd: C too large");
    if (!W && d->NC > d->C) IFCBK_FAIL(ctx, IFCBK_EINVAL, "head_fwd: the pooled-logits form needs NC <= C");
    int64_t total = (int64_t)d->N * (d->C / e);
    if (d->dtype == IFCBK_F32) hipLaunchKernelGGL(gap_kernel<float>, dim3(cdiv(total * GAP_SPLIT, 256)), dim3(256), 0, ST, (const float*)x, d->ldx, d->HW, d->C, total, mask, d->keep_scale, feat);
    else hipLaunchKernelGGL(gap_kernel<bf16_t>, dim3(cdiv(total * GAP_SPLIT, 256)), dim3(256), 0, ST, (const bf16_t*)x, d->ldx, d->HW, d->C, total, mask, d->keep_scale, feat);
    IFCBK_LAUNCH_CHECK(ctx, "gap");
    if (!W) {
        // squeezenet's classifier ends in the pool itself (Dropout -> Conv2d(512, NC, 1) -> ReLU -> AdaptiveAvgPool2d(1)): the
        // logits are the first NC pooled channels (the conv's output channels are padded to a whole 16-byte chunk)
        IFCBK_HIP(ctx, hipMemcpy2DAsync(logits, (size_t)d->NC * 4, feat, (size_t)d->C * 4, (size_t)d->NC * 4, (size_t)d->N,
                                        hipMemcpyDeviceToDevice, ST));
        return 0;
    }
    if (d->C <= 2048) hipLaunchKernelGGL(fc_fwd_kernel<8>, dim3(cdiv(d->N, 8), cdiv(d->NC, 8)), dim3(256), (size_t)8 * d->C * sizeof(float), ST, (const float*)feat, W, b, logits, d->N, d->C, d->NC);
    else hipLaunchKernelGGL(fc_fwd_kernel<4>, dim3(cdiv(d->N, 4), cdiv(d->NC, 8)), dim3(256), (size_t)4 * d->C * sizeof(float), ST, (const float*)feat, W, b, logits, d->N, d->C, d->NC);
    IFCBK_LAUNCH_CHECK(ctx, "fc_fwd");
    return 0;
}
extern "C" int ifcbk_head_bwd(ifcbk_ctx* ctx, const ifcbk_head_desc* d, const float* dlogits, const float* feat,
                              const uint8_t* mask, const float* W, float* dW, float* db, void* dx, int lddx,
                              int param_accumulate, void* stream) {
    if (!d || (d->dtype != IFCBK_BF16 && d->dtype != IFCBK_F32)) IFCBK_FAIL(ctx, IFCBK_EINVAL, "head_bwd: bad desc");
    const int e = dtype_chunk(d->dtype);
    if (d->C % e || lddx % e) IFCBK_FAIL(ctx, IFCBK_EINVAL, "head_bwd: channels must be multiples of %d", e);
    if (W) {
        hipLaunchKernelGGL(fc_wgrad_kernel, dim3(cdiv((int64_t)d->NC * d->C, 256)), dim3(256), 0, ST, dlogits, feat, dW, d->N, d->C, d->NC, param_accumulate);
        IFCBK_LAUNCH_CHECK(ctx, "fc_wgrad");
        hipLaunchKernelGGL(fc_bgrad_kernel, dim3(d->NC), dim3(64), 0, ST, dlogits, db, d->N, d->NC, param_accumulate);
        IFCBK_LAUNCH_CHECK(ctx, "fc_bgrad");
    }
    if (dx) {
        int64_t total = (int64_t)d->N * (d->C / e);
        if (d->dtype == IFCBK_F32) hipLaunchKernelGGL(head_dx_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, ST, dlogits, W, mask, d->keep_scale, (float*)dx, lddx, d->HW, d->C, d->NC, total);
        else hipLaunchKernelGGL(head_dx_kernel<bf16_t>, dim3(cdiv(total, 256)), dim3(256), 0, ST, dlogits, W, mask, d->keep_scale, (bf16_t*)dx, lddx, d->HW, d->C, d->NC, total);
        IFCBK_LAUNCH_CHECK(ctx, "head_dx");
    }
    return 0;
}
extern "C" int ifcbk_dropout_mask(ifcbk_ctx* ctx, uint8_t* mask, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream) {
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST, mask, n, p, seed, offset);
    IFCBK_LAUNCH_CHECK(ctx, "dropout_mask");
    return 0;
}

extern "C" int ifcbk_softmax_xent(ifcbk_ctx* ctx, const float* logits, const int64_t* target, int N, int NC, float weight,
                                  float* loss_out, int loss_accumulate, float* dlogits, void* stream) {
    if (N <= 0 || NC <= 0) IFCBK_FAIL(ctx, IFCBK_EINVAL, "softmax_xent: empty");
    hipLaunchKernelGGL(softmax_xent_kernel, dim3(1), dim3(1024), 0, ST, logits, target, N, NC, weight, loss_out, loss_accumulate, dlogits);
    IFCBK_LAUNCH_CHECK(ctx, "softmax_xent");
    return 0;
}
namespace {
__global__ void step_counters_kernel(int64_t* nbt, int n, float* loss_sum, const float* loss) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (nbt && i < n) nbt[i] += 1;
    if (i == 0 && loss_sum) loss_sum[0] += loss[0];
}
}  // namespace
extern "C" int ifcbk_step_counters(ifcbk_ctx* ctx, int64_t* nbt, int n, float* loss_sum, const float* loss, void* stream) {
    if (n < 0 || (loss_sum && !loss)) IFCBK_FAIL(ctx, IFCBK_EINVAL, "step_counters: bad arguments");
    hipLaunchKernelGGL(step_counters_kernel, dim3(cdiv(n > 0 ? n : 1, 256)), dim3(256), 0, ST, nbt, n, loss_sum, loss);
    IFCBK_LAUNCH_CHECK(ctx, "step_counters");
    return 0;
}
extern "C" int ifcbk_softmax(ifcbk_ctx* ctx, const float* logits, int N, int NC, float* probs, void* stream) {
    if (N <= 0) return 0;
    hipLaunchKernelGGL(softmax_kernel, dim3(cdiv(N, 64)), dim3(64), 0, ST, logits, N, NC, probs);
    IFCBK_LAUNCH_CHECK(ctx, "softmax");
    return 0;
}

extern "C" int ifcbk_adam_flat(ifcbk_ctx* ctx, float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) {
    if (step < 1) IFCBK_FAIL(ctx, IFCBK_EINVAL, "adam: step must be >= 1");
    float bc1 = 1.f - powf(beta1, (float)step);
    float sbc2 = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3(cdiv(cdiv(n, 4), 256)), dim3(256), 0, ST, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, sbc2, grad_scale);
    IFCBK_LAUNCH_CHECK(ctx, "adam");
    return 0;
}
extern "C" int ifcbk_sgd_flat(ifcbk_ctx* ctx, float* p, const float* g, float* mom, int64_t n, float lr, float momentum,
                              float weight_decay, float grad_scale, void* stream) {
    hipLaunchKernelGGL(sgd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST, p, g, mom, n, lr, momentum, weight_decay, grad_scale);
    IFCBK_LAUNCH_CHECK(ctx, "sgd");
    return 0;
}

extern "C" int ifcbk_nchw_to_nhwc(ifcbk_ctx* ctx, const float* x, int N, int C, int H, int W, int Cpad, int dtype,
                                  const float* scale3, const float* shift3, void* y, void* stream) {
    if ((dtype != IFCBK_BF16 && dtype != IFCBK_F32) || Cpad % dtype_chunk(dtype) || C > Cpad) IFCBK_FAIL(ctx, IFCBK_EINVAL, "nchw_to_nhwc: bad args");
    int64_t total = (int64_t)N * H * W;
    float s[3] = {1, 1, 1}, t[3] = {0, 0, 0};
    if (scale3) for (int i = 0; i < 3; ++i) s[i] = scale3[i];
    if (shift3) for (int i = 0; i < 3; ++i) t[i] = shift3[i];
    if (dtype == IFCBK_F32) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, ST, x, C, (int64_t)H * W, total, Cpad, s[0], s[1], s[2], t[0], t[1], t[2], (float*)y);
    else hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(cdiv(total, 256)), dim3(256), 0, ST, x, C, (int64_t)H * W, total, Cpad, s[0], s[1], s[2], t[0], t[1], t[2], (bf16_t*)y);
    IFCBK_LAUNCH_CHECK(ctx, "nchw_to_nhwc");
    return 0;
}
extern "C" int ifcbk_nhwc_to_nchw_f32(ifcbk_ctx* ctx, const void* x, int N, int C, int H, int W, int ldx, int dtype, float* y,
                                      void* stream) {
    if (dtype != IFCBK_BF16 && dtype != IFCBK_F32) IFCBK_FAIL(ctx, IFCBK_EINVAL, "nhwc_to_nchw: bad args");
    int64_t total = (int64_t)N * C * H * W;
    if (dtype == IFCBK_F32) hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, ST, (const float*)x, C, (int64_t)H * W, total, ldx, y);
    else hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(cdiv(total, 256)), dim3(256), 0, ST, (const bf16_t*)x, C, (int64_t)H * W, total, ldx, y);
    IFCBK_LAUNCH_CHECK(ctx, "nhwc_to_nchw");
    return 0;
}
