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
    // table layout [image][axis][field][S] (field 0 = first input index, 1 = tap count, 2.. = taps): the resize kernel's threads
    // (consecutive output x) read consecutive words of each field
    int32_t* row = tab + ((size_t)(img * 2 + axis) * (2 + kmax)) * S + xx;
    const int RS_ = S;
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
        row[(size_t)(2 + x) * RS_] = kq;
    }
    row[0] = xmin;
    row[RS_] = xmax;
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

// One block per (image, output row): the image's size / offset and the row's vertical taps are block-uniform (scalar loads), a
// thread's dependent chain is table -> pixels -> store.  As one flat grid of output pixels every thread first had to derive
// (image, row) and fetch those per-image values itself: four dependent memory round trips per output pixel, 0.30 ms per batch of 256
// with 366 MB written (1.2 TB/s).
__global__ __launch_bounds__(320) void roi_resize_kernel(RoiArgs a) {
    const int img = (int)(blockIdx.x / (unsigned)a.S);
    const int y = (int)(blockIdx.x - (unsigned)img * (unsigned)a.S);
    const int x0 = blockIdx.y * blockDim.x + threadIdx.x;
    const bool live = x0 < a.S;                          // (no early return: every thread reaches the barrier of the fast path)
    const int x = live ? x0 : a.S - 1;
    const int64_t i = ((int64_t)img * a.S + y) * a.S + x;
    const int h = a.hs[img], w = a.ws[img];
    const uint8_t* src = a.pixels + a.offs[img];
    const int fl = a.flips ? a.flips[img] : 0;
    const bool vflip = fl & 1, hflip = fl & 2;
    const int TS = a.S;                                  // field stride of the tap table
    const int32_t* th = a.tab + ((size_t)(img * 2 + 0) * (2 + a.kmax)) * a.S + x;
    const int32_t* tv = a.tab + ((size_t)(img * 2 + 1) * (2 + a.kmax)) * a.S + y;
    const int xmin = th[0], xn = th[TS], ymin = tv[0], yn = tv[TS];
    int res[3];
    // Fast path (grey ROIs no larger than the output: at most three taps per axis, the whole batch of the benchmark): the up to
    // three source rows of this output row are brought to LDS once with coalesced byte loads -- 3 vector-memory loads per thread
    // instead of 9 pixel loads + 9 tap loads, which kept the texture addresser busy with 64-byte transactions; same integer
    // arithmetic, taps beyond a row's count are zeros in the table (clamped addresses)
    constexpr int LR = 5, LW = 640;                     // LDS row images: up to 5 taps per axis (inputs up to 2x the output size)
    __shared__ uint8_t srow[LR][LW];
    const bool fast3 = a.cin == 1 && a.kmax == 3 && w <= (int)blockDim.x;          // block-uniform
    const bool lds_ok = !fast3 && a.cin == 1 && a.kmax <= LR && w <= LW;
    if (fast3) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            int row = ymin + (j < yn ? j : yn - 1);
            if (vflip) row = h - 1 - row;
            if ((int)threadIdx.x < w) srow[j][threadIdx.x] = src[(size_t)row * w + threadIdx.x];
        }
        __syncthreads();
        const int t0 = th[2 * TS], t1 = th[3 * TS], t2 = th[4 * TS];
        int c0 = xmin, c1 = xmin + (xn > 1 ? 1 : 0), c2 = xmin + (xn > 2 ? 2 : xn - 1);
        if (hflip) { c0 = w - 1 - c0; c1 = w - 1 - c1; c2 = w - 1 - c2; }
        int accv = 1 << (PRECISION_BITS - 1);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int acch = (1 << (PRECISION_BITS - 1)) + (int)srow[j][c0] * t0 + (int)srow[j][c1] * t1 + (int)srow[j][c2] * t2;
            accv += clip8(acch) * tv[(2 + j) * TS];
        }
        res[0] = clip8(accv);
    } else if (lds_ok) {
        for (int j = 0; j < yn; ++j) {
            int row = ymin + j;
            if (vflip) row = h - 1 - row;
            for (int c = threadIdx.x; c < w; c += blockDim.x) srow[j][c] = src[(size_t)row * w + c];
        }
        __syncthreads();
        int accv = 1 << (PRECISION_BITS - 1);
        for (int j = 0; j < yn; ++j) {
            int acch = 1 << (PRECISION_BITS - 1);
            for (int k = 0; k < xn; ++k) {
                int col = xmin + k;
                if (hflip) col = w - 1 - col;
                acch += (int)srow[j][col] * th[(size_t)(2 + k) * TS];
            }
            accv += clip8(acch) * tv[(size_t)(2 + j) * TS];
        }
        res[0] = clip8(accv);
    } else
    for (int c = 0; c < a.cin; ++c) {
        int accv = 1 << (PRECISION_BITS - 1);
        for (int j = 0; j < yn; ++j) {
            int row = ymin + j;
            if (vflip) row = h - 1 - row;
            int acch = 1 << (PRECISION_BITS - 1);
            for (int k = 0; k < xn; ++k) {
                int col = xmin + k;
                if (hflip) col = w - 1 - col;
                acch += (int)src[((size_t)row * w + col) * a.cin + c] * th[(size_t)(2 + k) * TS];
            }
            accv += clip8(acch) * tv[(size_t)(2 + j) * TS];
        }
        res[c] = clip8(accv);
    }
    if (!live) return;
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

// The benchmark's case as its own kernel -- grey ROIs no larger than the output (three taps per axis): one block per RPB consecutive
// output rows of one image.  With one row per block (the kernel above) a block lived for four dependent memory round trips
// (sizes / offset -> tap tables -> source rows -> store) to produce 299 bytes: 76,544 short-lived blocks per batch of 256, 306 us
// even when only the u8 plane is written.  Here the image's scalars and a thread's horizontal taps are fetched once per RPB rows
// and the 3 x RPB source-row loads are in flight together.  Same integer arithmetic as the fast path above.
constexpr int RPB = 8;
__global__ __launch_bounds__(320) void roi_resize3_kernel(RoiArgs a) {
    const unsigned nrb = (unsigned)(a.S + RPB - 1) / RPB;
    const int img = (int)(blockIdx.x / nrb);
    const int y0 = (int)(blockIdx.x - (unsigned)img * nrb) * RPB;
    const bool live = (int)threadIdx.x < a.S;
    const int x = live ? (int)threadIdx.x : a.S - 1;
    const int h = a.hs[img] > 0 ? a.hs[img] : 1, w = a.ws[img] > 0 ? a.ws[img] : 1;
    // this kernel is chosen from the caller's max_h / max_w (kmax == 3: no ROI larger than the output).  A table entry that
    // breaks that promise (stale maxima) must not read unstaged LDS: columns are clamped to what the staging below holds --
    // such a ROI comes out wrong (its coefficient table was sized for three taps), never out of bounds
    const int wl = w < 320 ? w : 320;
    const uint8_t* src = a.pixels + a.offs[img];
    const int fl = a.flips ? a.flips[img] : 0;
    const bool vflip = fl & 1, hflip = fl & 2;
    const int TS = a.S;
    const int32_t* th = a.tab + ((size_t)(img * 2 + 0) * 5) * a.S + x;
    const int32_t* tv = a.tab + ((size_t)(img * 2 + 1) * 5) * a.S;
    __shared__ uint8_t srow[RPB][3][320];
    int tvv[RPB][3];
#pragma unroll
    for (int r = 0; r < RPB; ++r) {
        const int y = y0 + r < a.S ? y0 + r : a.S - 1;                 // block-uniform
        const int ymin = tv[y], yn = tv[TS + y];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            tvv[r][j] = tv[(2 + j) * TS + y];
            int row = ymin + (j < yn ? j : yn - 1);
            if (vflip) row = h - 1 - row;
            row = row < 0 ? 0 : (row >= h ? h - 1 : row);
            // (strided: a ROI wider than the block -- maxima understated by the caller -- still stages every column it reads)
            for (int xx = (int)threadIdx.x; xx < wl; xx += (int)blockDim.x) srow[r][j][xx] = src[(size_t)row * w + xx];
        }
    }
    const int xmin = th[0], xn = th[TS];
    const int t0 = th[2 * TS], t1 = th[3 * TS], t2 = th[4 * TS];
    int c0 = xmin, c1 = xmin + (xn > 1 ? 1 : 0), c2 = xmin + (xn > 2 ? 2 : xn - 1);
    if (hflip) { c0 = w - 1 - c0; c1 = w - 1 - c1; c2 = w - 1 - c2; }
    c0 = c0 < 0 ? 0 : (c0 >= wl ? wl - 1 : c0);
    c1 = c1 < 0 ? 0 : (c1 >= wl ? wl - 1 : c1);
    c2 = c2 < 0 ? 0 : (c2 >= wl ? wl - 1 : c2);
    __syncthreads();
    if (!live) return;
#pragma unroll
    for (int r = 0; r < RPB; ++r) {
        if (y0 + r >= a.S) break;
        int accv = 1 << (PRECISION_BITS - 1);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int acch = (1 << (PRECISION_BITS - 1)) + (int)srow[r][j][c0] * t0 + (int)srow[r][j][c1] * t1 + (int)srow[r][j][c2] * t2;
            accv += clip8(acch) * tvv[r][j];
        }
        const int res = clip8(accv);
        const int64_t i = ((int64_t)img * a.S + (y0 + r)) * a.S + x;
        if (a.out_u8) a.out_u8[i] = (uint8_t)res;
        if (a.out) {
            for (int cc = 0; cc < a.cout; cc += 8) {
                float f[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = cc + j;
                    float v = 0.f;
                    if (c < 3) {
                        v = (float)res / 255.0f;
                        v = (v - a.mean[c]) / a.std[c];
                        v = v * a.tsc[c] + a.tsh[c];
                    }
                    f[j] = v;
                }
                if (a.f32) {
                    float* o = (float*)a.out + i * a.cout + cc;
                    *reinterpret_cast<float4*>(o) = make_float4(f[0], f[1], f[2], f[3]);
                    *reinterpret_cast<float4*>(o + 4) = make_float4(f[4], f[5], f[6], f[7]);
                } else {
                    *reinterpret_cast<uint4*>((bf16_t*)a.out + i * a.cout + cc) = pack8(f);
                }
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
    const int bx = d->S <= 64 ? 64 : d->S <= 128 ? 128 : d->S <= 192 ? 192 : d->S <= 256 ? 256 : 320;      // threads per output row
    if (d->in_channels == 1 && kmax == 3 && d->S <= 320)      // (kmax == 3: no ROI is larger than the output, so w <= S <= 320 threads)
        hipLaunchKernelGGL(roi_resize3_kernel, dim3((unsigned)(d->n_img * cdiv(d->S, RPB))), dim3(bx), 0, st, a);
    else
        hipLaunchKernelGGL(roi_resize_kernel, dim3((unsigned)(d->n_img * d->S), (unsigned)cdiv(d->S, bx)), dim3(bx), 0, st, a);
    IFCBK_LAUNCH_CHECK(ctx, "roi_resize");
    return 0;
}
