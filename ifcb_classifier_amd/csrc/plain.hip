// Layers of the backbones WITHOUT BatchNorm behind every conv: alexnet, vgg*, squeezenet1_1 and the classifier stacks /
// dense concatenations of vgg*_bn and densenet* (reference call sites: neuston_models.py:27-36,40-42 -> torchvision
// alexnet.py, vgg.py, squeezenet.py, densenet.py).  The convolutions themselves (with their bias (+ReLU) in the epilogue) are
// the implicit-GEMM kernels of conv_igemm.hip; here are the elementwise / reduction passes around them:
//   * bias_relu_bwd: dz = dy * (y > 0), dbias = column sums of dz       (autograd of relu_(conv(x) + b))
//   * dropout_apply: y (+)= x * mask * scale                            (nn.Dropout forward and backward)
//   * flatten_chw:   NHWC [N,HW,C] <-> torch.flatten(NCHW, 1) order     (the classifier's Linear weights index (c, h, w))
// All HBM-bound; 16-byte chunks, deterministic two-stage column sums (no float atomics: training steps stay bitwise
// reproducible).
#include "common.h"

namespace {



// Block b owns rows [b*rows, (b+1)*rows) and ALL K channels.  A thread keeps one 16-byte chunk column; 256 / W row lanes share
// a column group of W = min(256, chunks per row) columns and are combined through LDS in a fixed order.
template <class T>
__global__ __launch_bounds__(256) void bias_relu_bwd_kernel(const T* y, int ldy, const T* dy, int lddy, T* dz, int lddz, int64_t M,
                                                            int K, int rows, int relu, float* part) {
    constexpr int E = Chunk<T>::N;
    __shared__ float sred[256 * E];
    const int t = threadIdx.x;
    const int cpr = K / E;
    const int64_t r0 = (int64_t)blockIdx.x * rows;
    const int64_t r1 = r0 + rows < M ? r0 + rows : M;
    for (int cg = 0; cg < cpr; cg += 256) {
        const int Wc = cpr - cg < 256 ? cpr - cg : 256;
        const int rl = 256 / Wc;
        const int col = t % Wc, lane_r = t / Wc;
        float acc[E];
#pragma unroll
        for (int j = 0; j < E; ++j) acc[j] = 0.f;
        if (lane_r < rl) {
            const int c = (cg + col) * E;
            for (int64_t r = r0 + lane_r; r < r1; r += rl) {
                float g[E];
                Chunk<T>::load(dy + r * lddy + c, g);
                if (relu) {
                    float a[E];
                    Chunk<T>::load(y + r * ldy + c, a);
#pragma unroll
                    for (int j = 0; j < E; ++j) g[j] = a[j] > 0.f ? g[j] : 0.f;
                }
                if (dz) Chunk<T>::store(dz + r * lddz + c, g);
#pragma unroll
                for (int j = 0; j < E; ++j) acc[j] += g[j];
            }
        }
        if (part) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < E; ++j) sred[t * E + j] = acc[j];
            __syncthreads();
            if (t < Wc) {
                float s[E];
#pragma unroll
                for (int j = 0; j < E; ++j) s[j] = 0.f;
                for (int q = 0; q < rl; ++q)
#pragma unroll
                    for (int j = 0; j < E; ++j) s[j] += sred[(q * Wc + t) * E + j];
#pragma unroll
                for (int j = 0; j < E; ++j) part[(size_t)blockIdx.x * K + (cg + t) * E + j] = s[j];
            }
        }
    }
}

__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* part, int nblk, int K, float* out, int accumulate) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += part[(size_t)b * K + k];
    out[k] = accumulate ? out[k] + s : s;
}

template <class T>
__global__ __launch_bounds__(256) void dropout_apply_kernel(const T* x, const uint8_t* mask, float scale, T* y, int64_t nchunks,
                                                            int accumulate) {
    constexpr int E = Chunk<T>::N;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nchunks) return;
    float f[E];
    Chunk<T>::load(x + i * E, f);
    if (mask) {
        int m[E];
        ArgPack<E>::load(mask + i * E, m);
#pragma unroll
        for (int j = 0; j < E; ++j) f[j] = m[j] ? f[j] * scale : 0.f;
    }
    if (accumulate) {
        float o[E];
        Chunk<T>::load(y + i * E, o);
#pragma unroll
        for (int j = 0; j < E; ++j) f[j] += o[j];
    }
    Chunk<T>::store(y + i * E, f);
}

// to_chw = 1: flat[n][c*HW + hw] = x[n][hw][c];  to_chw = 0: x[n][hw][c] (+)= flat[n][c*HW + hw].  Thread per element of the
// NHWC side's 16-byte chunk (the tensors are N x 9216 / N x 25088 elements: nothing to optimise)
template <class T>
__global__ __launch_bounds__(256) void flatten_chw_kernel(T* x, int ldx, T* flat, int HW, int C, int64_t total, int to_chw, int accumulate) {
    constexpr int E = Chunk<T>::N;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int cpr = C / E;
    const int c = (int)(i % cpr) * E;
    const int64_t pix = i / cpr;
    const int hw = (int)(pix % HW);
    const int64_t n = pix / HW;
    T* px = x + pix * ldx + c;
    T* pf = flat + n * (int64_t)HW * C + (int64_t)c * HW + hw;
    float f[E];
    if (to_chw) {
        Chunk<T>::load(px, f);
#pragma unroll
        for (int j = 0; j < E; ++j) pf[(int64_t)j * HW] = from_f32<T>(f[j]);
    } else {
#pragma unroll
        for (int j = 0; j < E; ++j) f[j] = to_f32(pf[(int64_t)j * HW]);
        if (accumulate) {
            float o[E];
            Chunk<T>::load(px, o);
#pragma unroll
            for (int j = 0; j < E; ++j) f[j] += o[j];
        }
        Chunk<T>::store(px, f);
    }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int ifcbk_bias_relu_bwd_rows(int64_t M) {
    int64_t rows = (M + 1023) / 1024;
    return (int)(rows < 16 ? 16 : rows);
}

extern "C" size_t ifcbk_bias_relu_bwd_workspace(int64_t M, int K) {
    const int rows = ifcbk_bias_relu_bwd_rows(M);
    return (size_t)cdiv(M, rows) * K * sizeof(float);
}

extern "C" int ifcbk_bias_relu_bwd(ifcbk_ctx* ctx, int64_t M, int K, int dtype, const void* y, int ldy, const void* dy, int lddy,
                                   void* dz, int lddz, int relu, float* dbias, int param_accumulate, void* stream) {
    if (M <= 0 || K <= 0 || !dy) IFCBK_FAIL(ctx, IFCBK_EINVAL, "bias_relu_bwd: empty");
    if (dtype != IFCBK_BF16 && dtype != IFCBK_F32) IFCBK_FAIL(ctx, IFCBK_EUNSUPPORTED, "bias_relu_bwd: dtype must be bf16 or f32");
    const int e = dtype_chunk(dtype);
    if (K % e || lddy % e || (relu && (!y || ldy % e)) || (dz && lddz % e))
        IFCBK_FAIL(ctx, IFCBK_EINVAL, "bias_relu_bwd: K=%d and the row strides must be multiples of %d", K, e);
    if (!dz && !dbias) return 0;
    const int rows = ifcbk_bias_relu_bwd_rows(M);
    const int nblk = cdiv(M, rows);
    float* part = nullptr;
    if (dbias) {
        const size_t need = (size_t)nblk * K * sizeof(float);
        if (need > ctx->ws_bytes) IFCBK_FAIL(ctx, IFCBK_ENOMEM, "bias_relu_bwd: workspace %zu > reserved %zu", need, ctx->ws_bytes);
        part = (float*)ctx->ws;
    }
    if (dtype == IFCBK_F32)
        hipLaunchKernelGGL(bias_relu_bwd_kernel<float>, dim3(nblk), dim3(256), 0, ST, (const float*)y, ldy, (const float*)dy, lddy, (float*)dz, lddz, M, K, rows, relu, part);
    else
        hipLaunchKernelGGL(bias_relu_bwd_kernel<bf16_t>, dim3(nblk), dim3(256), 0, ST, (const bf16_t*)y, ldy, (const bf16_t*)dy, lddy, (bf16_t*)dz, lddz, M, K, rows, relu, part);
    IFCBK_LAUNCH_CHECK(ctx, "bias_relu_bwd");
    if (dbias) {
        hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cdiv(K, 256)), dim3(256), 0, ST, (const float*)part, nblk, K, dbias, param_accumulate);
        IFCBK_LAUNCH_CHECK(ctx, "colsum_finalize");
    }
    return 0;
}

extern "C" int ifcbk_dropout_apply(ifcbk_ctx* ctx, int64_t n, int dtype, const void* x, const uint8_t* mask, float scale, void* y,
                                   int accumulate, void* stream) {
    if (n <= 0 || !x || !y) IFCBK_FAIL(ctx, IFCBK_EINVAL, "dropout_apply: empty");
    if (dtype != IFCBK_BF16 && dtype != IFCBK_F32) IFCBK_FAIL(ctx, IFCBK_EUNSUPPORTED, "dropout_apply: dtype must be bf16 or f32");
    const int e = dtype_chunk(dtype);
    if (n % e) IFCBK_FAIL(ctx, IFCBK_EINVAL, "dropout_apply: n=%lld must be a multiple of %d", (long long)n, e);
    const int64_t nch = n / e;
    if (dtype == IFCBK_F32)
        hipLaunchKernelGGL(dropout_apply_kernel<float>, dim3(cdiv(nch, 256)), dim3(256), 0, ST, (const float*)x, mask, scale, (float*)y, nch, accumulate);
    else
        hipLaunchKernelGGL(dropout_apply_kernel<bf16_t>, dim3(cdiv(nch, 256)), dim3(256), 0, ST, (const bf16_t*)x, mask, scale, (bf16_t*)y, nch, accumulate);
    IFCBK_LAUNCH_CHECK(ctx, "dropout_apply");
    return 0;
}

extern "C" int ifcbk_flatten_chw(ifcbk_ctx* ctx, int N, int HW, int C, int dtype, void* x, int ldx, void* flat, int to_chw,
                                 int accumulate, void* stream) {
    if (N <= 0 || HW <= 0 || C <= 0 || !x || !flat) IFCBK_FAIL(ctx, IFCBK_EINVAL, "flatten_chw: empty");
    if (dtype != IFCBK_BF16 && dtype != IFCBK_F32) IFCBK_FAIL(ctx, IFCBK_EUNSUPPORTED, "flatten_chw: dtype must be bf16 or f32");
    const int e = dtype_chunk(dtype);
    if (C % e || ldx % e) IFCBK_FAIL(ctx, IFCBK_EINVAL, "flatten_chw: C=%d ldx=%d must be multiples of %d", C, ldx, e);
    const int64_t total = (int64_t)N * HW * (C / e);
    if (dtype == IFCBK_F32)
        hipLaunchKernelGGL(flatten_chw_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, ST, (float*)x, ldx, (float*)flat, HW, C, total, to_chw, accumulate);
    else
        hipLaunchKernelGGL(flatten_chw_kernel<bf16_t>, dim3(cdiv(total, 256)), dim3(256), 0, ST, (bf16_t*)x, ldx, (bf16_t*)flat, HW, C, total, to_chw, accumulate);
    IFCBK_LAUNCH_CHECK(ctx, "flatten_chw");
    return 0;
}
