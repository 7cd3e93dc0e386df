// On-GPU ROI preprocessing: ragged u8 ROIs -> PIL-exact bilinear resize (antialiased, 8-bit fixed point,
// horizontal pass then vertical pass with a u8 intermediate) -> /255 -> Normalize -> NHWC bf16.
//
// Restates Pillow's ImagingResample (libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
// ImagingResampleHorizontal_8bpc / Vertical_8bpc) -- the arithmetic behind transforms.Resize([S,S]) at
// /root/reference/neuston_data.py:345 and :460.  The coefficient maths runs in IEEE double on the device
// with contraction off, so the 22-bit fixed-point taps are bit-identical to the CPU's.
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// one thread per (image, axis, output index): bounds + int taps
__global__ void roi_coeffs_kernel(const int32_t* hs, const int32_t* ws, int n_img, int S, int kmax, int32_t* tab) {
#pragma clang fp contract(off)
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_img * 2 * S) return;
    int xx = i % S;
    int axis = (i / S) & 1;
    int img = i / (2 * S);
    int inSize = axis == 0 ? ws[img] : hs[img];
    int32_t* row = tab + (size_t)i * (2 + kmax);
    double scale = (double)((float)inSize - 0.0f) / (double)S;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    double support = 1.0 * filterscale;              // bilinear support = 1.0
    double center = 0.0 + ((double)xx + 0.5) * scale;
    double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > inSize) xmax = inSize;
    xmax -= xmin;
    if (xmax > kmax) xmax = kmax;                     // cannot happen when kmax is sized from max dims
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
        double a = ((double)(x + xmin) - center + 0.5) * ss;
        if (a < 0.0) a = -a;
        double w = a < 1.0 ? 1.0 - a : 0.0;
        ww += w;
    }
    for (int x = 0; x < kmax; ++x) {
        int kq = 0;
        if (x < xmax) {
            double a = ((double)(x + xmin) - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            double w = a < 1.0 ? 1.0 - a : 0.0;
            if (ww != 0.0) w = w / ww;
            kq = w < 0.0 ? (int)(-0.5 + w * (double)(1 << PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << PRECISION_BITS));
        }
        row[2 + x] = kq;
    }
    row[0] = xmin;
    row[1] = xmax;
}

struct RoiArgs {
    const uint8_t* pixels;
    const int64_t* offs;
    const int32_t* hs;
    const int32_t* ws;
    const uint8_t* flips;
    const int32_t* tab;
    void* out;
    int f32;
    uint8_t* out_u8;
    int n_img, S, cin, cout, kmax;
    float mean[3], std[3], tsc[3], tsh[3];
};

__global__ __launch_bounds__(256) void roi_resize_kernel(RoiArgs a) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)a.n_img * a.S * a.S;
    if (i >= total) return;
    int x = (int)(i % a.S);
    int y = (int)((i / a.S) % a.S);
    int img = (int)(i / ((int64_t)a.S * a.S));
    const int h = a.hs[img], w = a.ws[img];
    const uint8_t* src = a.pixels + a.offs[img];
    const int fl = a.flips ? a.flips[img] : 0;
    const bool vflip = fl & 1, hflip = fl & 2;
    const int32_t* th = a.tab + ((size_t)(img * 2 + 0) * a.S + x) * (2 + a.kmax);
    const int32_t* tv = a.tab + ((size_t)(img * 2 + 1) * a.S + y) * (2 + a.kmax);
    const int xmin = th[0], xn = th[1], ymin = tv[0], yn = tv[1];
    int res[3];
    for (int c = 0; c < a.cin; ++c) {
        int accv = 1 << (PRECISION_BITS - 1);
        for (int j = 0; j < yn; ++j) {
            int row = ymin + j;
            if (vflip) row = h - 1 - row;
            int acch = 1 << (PRECISION_BITS - 1);
            for (int k = 0; k < xn; ++k) {
                int col = xmin + k;
                if (hflip) col = w - 1 - col;
                acch += (int)src[((size_t)row * w + col) * a.cin + c] * th[2 + k];
            }
            accv += clip8(acch) * tv[2 + j];
        }
        res[c] = clip8(accv);
    }
    if (a.cin == 1) res[1] = res[2] = res[0];
    if (a.out_u8)
        for (int c = 0; c < a.cin; ++c) a.out_u8[i * a.cin + c] = (uint8_t)res[c];
    if (a.out) {
        for (int c0 = 0; c0 < a.cout; c0 += 8) {
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int c = c0 + j;
                float v = 0.f;
                if (c < 3) {
                    v = (float)res[c] / 255.0f;
                    v = (v - a.mean[c]) / a.std[c];
                    v = v * a.tsc[c] + a.tsh[c];
                }
                f[j] = v;
            }
            if (a.f32) {
                float* o = (float*)a.out + i * a.cout + c0;
                *reinterpret_cast<float4*>(o) = make_float4(f[0], f[1], f[2], f[3]);
                *reinterpret_cast<float4*>(o + 4) = make_float4(f[4], f[5], f[6], f[7]);
            } else {
                *reinterpret_cast<uint4*>((bf16_t*)a.out + i * a.cout + c0) = pack8(f);
            }
        }
    }
}

int kmax_for(int max_h, int max_w, int S) {
    int m = max_h > max_w ? max_h : max_w;
    double scale = (double)m / S;
    if (scale < 1.0) scale = 1.0;
    return (int)ceil(scale) * 2 + 1;
}

}  // namespace

extern "C" size_t ifcbk_roi_preprocess_workspace(const ifcbk_roi_desc* d, int max_h, int max_w) {
    int kmax = kmax_for(max_h, max_w, d->S);
    return (size_t)d->n_img * 2 * d->S * (2 + kmax) * sizeof(int32_t);
}

extern "C" int ifcbk_roi_preprocess(ifcbk_ctx* ctx, const ifcbk_roi_desc* d, const uint8_t* pixels, const int64_t* offs,
                                    const int32_t* hs, const int32_t* ws, const uint8_t* flips, int max_h, int max_w,
                                    void* out, uint8_t* out_u8, void* stream) {
    if (!d || d->n_img <= 0) return IFCBK_OK;   // empty bin: nothing to do
    if ((d->dtype != IFCBK_BF16 && d->dtype != IFCBK_F32) || (d->in_channels != 1 && d->in_channels != 3) || d->out_channels % 8 || d->out_channels < 8)
        IFCBK_FAIL(ctx, IFCBK_EINVAL, "roi_preprocess: bad desc");
    if (max_h < 1 || max_w < 1) IFCBK_FAIL(ctx, IFCBK_EINVAL, "roi_preprocess: max dims");
    size_t need = ifcbk_roi_preprocess_workspace(d, max_h, max_w);
    if (need > ctx->ws_bytes) IFCBK_FAIL(ctx, IFCBK_ENOMEM, "roi_preprocess: workspace %zu > reserved %zu", need, ctx->ws_bytes);
    int kmax = kmax_for(max_h, max_w, d->S);
    hipStream_t st = (hipStream_t)stream;
    int nco = d->n_img * 2 * d->S;
    hipLaunchKernelGGL(roi_coeffs_kernel, dim3(cdiv(nco, 256)), dim3(256), 0, st, hs, ws, d->n_img, d->S, kmax, (int32_t*)ctx->ws);
    IFCBK_LAUNCH_CHECK(ctx, "roi_coeffs");
    RoiArgs a;
    a.pixels = pixels; a.offs = offs; a.hs = hs; a.ws = ws; a.flips = d->flip_bits_valid ? flips : nullptr;
    a.tab = (const int32_t*)ctx->ws; a.out = out; a.f32 = d->dtype == IFCBK_F32; a.out_u8 = out_u8;
    a.n_img = d->n_img; a.S = d->S; a.cin = d->in_channels; a.cout = d->out_channels; a.kmax = kmax;
    for (int i = 0; i < 3; ++i) { a.mean[i] = d->mean[i]; a.std[i] = d->std[i]; a.tsc[i] = d->tin_scale[i]; a.tsh[i] = d->tin_shift[i]; }
    int64_t total = (int64_t)d->n_img * d->S * d->S;
    hipLaunchKernelGGL(roi_resize_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, a);
    IFCBK_LAUNCH_CHECK(ctx, "roi_resize");
    return 0;
}
