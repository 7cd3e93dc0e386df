/* libifcbk -- C-ABI of the MI355X (gfx950) kernels behind the ifcb_classifier train/infer hot path.
 *
 * The reference (WHOIGit/ifcb_classifier v0.3.1) has NO FFI: its hot path is Python calling
 * torch/torchvision/PIL.  Each entry point below therefore cites the reference call site whose
 * third-party primitive it replaces (file:line under /root/reference); INTEGRATION.md shows the
 * ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - plain C, no torch types: raw device pointers + explicit dims; the caller owns every buffer.
 *   - every call is asynchronous and ordered on `stream` (a hipStream_t passed as void*); no host
 *     sync, no allocation in steady state (hipGraph-capturable) once ifcbk_ctx_reserve has been called.
 *   - return 0 on success, negative IFCBK_E* on error; text via ifcbk_last_error(); no C++ exception
 *     crosses the ABI.
 *   - a ctx is thread-compatible (one caller at a time), one per (process, GPU).
 *   - activations are NHWC; `ld*` is the element stride between consecutive pixels so a tensor may be a
 *     channel slice of a wider (concat) buffer.  Channel counts, ld* and slice offsets are multiples of 8.
 *   - dtype: IFCBK_BF16 (2-byte storage, fp32 accumulate, v_mfma_f32_16x16x32_bf16) is the performance mode;
 *     IFCBK_F32 (4-byte storage, v_mfma_f32_16x16x4_f32 = exact fp32 fma chain) is the parity mode.  Channel counts,
 *     ld* and slice offsets are multiples of 16 bytes / element size (8 for bf16, 4 for f32).
 */
#ifndef IFCBK_H
#define IFCBK_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden: only the entry points declared here leave its dynamic symbol table */
#define IFCBK_API __attribute__((visibility("default")))

#define IFCBK_OK          0
#define IFCBK_EINVAL     -1   /* bad descriptor / unsupported shape */
#define IFCBK_EHIP       -2   /* HIP runtime error */
#define IFCBK_ENOMEM     -3   /* workspace too small: call ifcbk_ctx_reserve outside capture */
#define IFCBK_EUNSUPPORTED -4

#define IFCBK_BF16 0
#define IFCBK_F32  1

typedef struct ifcbk_ctx ifcbk_ctx;

IFCBK_API const char* ifcbk_version(void);
IFCBK_API int  ifcbk_ctx_create(int device, ifcbk_ctx** out);
IFCBK_API int  ifcbk_ctx_destroy(ifcbk_ctx* ctx);
/* grow the ctx-owned workspace (split-K slabs, resize coefficient tables) to >= bytes; syncs the device */
IFCBK_API int  ifcbk_ctx_reserve(ifcbk_ctx* ctx, size_t bytes);
IFCBK_API size_t ifcbk_ctx_workspace_bytes(ifcbk_ctx* ctx);
/* number of program lanes (see ifcbk_op.flags; 1..8, default 4) that get a workspace arena from ifcbk_ctx_reserve; call before the
 * first reserve.  A program that names a lane beyond it is refused. */
IFCBK_API int  ifcbk_ctx_set_lanes(ifcbk_ctx* ctx, int lanes);
/* graphs captured through this ctx and not yet destroyed (the ctx owns them: ifcbk_ctx_destroy destroys what is left) */
IFCBK_API int  ifcbk_ctx_live_graphs(ifcbk_ctx* ctx);
IFCBK_API const char* ifcbk_last_error(ifcbk_ctx* ctx);

/* ------------------------------------------------------------------ convolution (implicit GEMM, MFMA)
 * replaces aten::conv2d fwd/bwd reached from  neuston_models.py:66-68 (forward) and the autograd
 * backward of [PL] automatic optimisation (neuston_models.py:63-64,81-86).                          */
typedef struct {
    int32_t N, H, W, C;      /* input  [N,H,W,C], C = channels the kernel reads (padded to %8)       */
    int32_t ldx;             /* input pixel stride (elements)                                        */
    int32_t K;               /* output channels                                                      */
    int32_t R, S;            /* filter height, width                                                 */
    int32_t stride_h, stride_w, pad_h, pad_w;
    int32_t P, Q;            /* output height, width                                                 */
    int32_t ldy;             /* output pixel stride (elements)                                       */
    int32_t Cw;              /* channels of the MASTER weight/grad (3 for the stem conv, else == C)   */
    int32_t dtype;
} ifcbk_conv_desc;

/* y[n,p,q,k] = sum_{r,s,c} x[n, p*sh-ph+r, q*sw-pw+s, c] * w[k,r,s,c]   (w: bf16 [K][R][S][C])
 * bn_part (nullable): fp32 [ceil(N*P*Q/128)][2][K] per-M-block partial (sum, sum of squares) of the
 * ROUNDED outputs, consumed by ifcbk_bn_finalize.                                                    */
IFCBK_API int ifcbk_conv2d_fwd(ifcbk_ctx*, const ifcbk_conv_desc*, const void* x, const void* w, void* y,
                     float* bn_part, void* stream);
/* inference form: the per-channel affine of an eval-mode BatchNorm (scale/shift from ifcbk_bn_finalize with
 * part==NULL), the optional resnet residual add and the ReLU are applied in the conv epilogue, so neither the raw
 * conv output nor a separate bn_apply pass touches HBM:  y = act(conv(x,w)*scale[k] + shift[k] (+ residual)).     */
IFCBK_API int ifcbk_conv2d_fwd_affine(ifcbk_ctx*, const ifcbk_conv_desc*, const void* x, const void* w, void* y,
                            const float* scale, const float* shift, const void* residual, int ldr, int relu,
                            void* stream);
/* ... and, where that activation feeds ONLY a 3x3 / stride-2 / unpadded max pool (inception Conv2d_2b_3x3 -> maxpool1), the pool as well:
 * y_pooled [N, (P-3)/2+1, (Q-3)/2+1, ldp] = maxpool(act(conv(x,w)*scale[k] + shift[k])), bit-identical to ifcbk_conv2d_fwd_affine followed
 * by ifcbk_maxpool_fwd, the activation is never stored.  Served for the shapes of the row-streaming kernel (3x3 / stride 1, 32 -> 64
 * channels, bf16, Q <= 160): ..._ok returns 1, otherwise the caller runs the two calls.  Replaces [TV] `F.relu(bn(conv(x)))` +
 * `F.max_pool2d(x, 3, 2)` of Inception3._forward in eval mode (reference call site neuston_models.py:94-103, 152-157).              */
IFCBK_API int ifcbk_conv2d_fwd_affine_maxpool_ok(const ifcbk_conv_desc*);
IFCBK_API int ifcbk_conv2d_fwd_affine_maxpool(ifcbk_ctx*, const ifcbk_conv_desc*, const void* x, const void* w, void* y_pooled, int ldp,
                                    const float* scale, const float* shift, int relu, void* stream);
/* Eval-mode sibling GEMM: ONE convolution whose d->K output channels belong to nseg (<= 4) consecutive segments with their own
 * destination tensors ys[s] (pixel stride ldys[s], ksegs[s] channels; sizes sum to d->K; d->ldy is ignored).  affine[s] = 1:
 * y = relu(conv * scale[k] + shift[k]) with the per-channel arrays indexed by the merged channel k (the folded BatchNorm of
 * [TV] BasicConv2d); affine[s] = 0: the raw convolution (a pool branch whose average pool applies the affine afterwards).
 * Replaces the 3-4 1x1 convolutions that read one Inception block input (inception.py InceptionA/C/E.forward), reference
 * call site neuston_models.py:94-103,152-157 (eval forward).                                                             */
IFCBK_API int ifcbk_conv2d_fwd_affine_segments(ifcbk_ctx*, const ifcbk_conv_desc* d, const void* x, const void* w, int nseg,
                                     void* const* ys, const int32_t* ldys, const int32_t* ksegs, const int32_t* affine,
                                     const float* scale, const float* shift, void* stream);
/* dx[n,h,w,c] (+)= sum_{k,r,s} dy[n,p,q,k] * w[k,r,s,c];  wT = bf16 [C][R][S][K] with r,s FLIPPED
 * (made by ifcbk_weight_pack).  accumulate!=0 adds into dx.                                          */
IFCBK_API int ifcbk_conv2d_dgrad(ifcbk_ctx*, const ifcbk_conv_desc*, const void* dy, const void* wT, void* dx,
                       int accumulate, void* stream);
/* dw[k,r,s,c] (+)= sum_{n,p,q} dy[n,p,q,k] * x[n, p*sh-ph+r, q*sw-pw+s, c];  dw fp32 [K][R][S][Cw];
 * deterministic split-K through the ctx workspace.                                                  */
IFCBK_API int ifcbk_conv2d_wgrad(ifcbk_ctx*, const ifcbk_conv_desc*, const void* x, const void* dy, float* dw,
                       int accumulate, void* stream);
/* horizontally fused convs (siblings reading the same input, filters concatenated along K): ONE gradient GEMM over
 * the concatenated dy, whose row segments kseg[i] are written to their own master-gradient tensors dws[i]
 * (host arrays, nseg <= 8).                                                                                        */
IFCBK_API int ifcbk_conv2d_wgrad_segments(ifcbk_ctx*, const ifcbk_conv_desc*, const void* x, const void* dy, int nseg,
                                float* const* dws, const int32_t* kseg, int accumulate, void* stream);
IFCBK_API size_t ifcbk_conv2d_wgrad_workspace(const ifcbk_conv_desc*);
/* the weight gradients of n <= 8 INDEPENDENT layers (each as ifcbk_conv2d_wgrad: its own x, dy, dw and descriptor) as ONE
 * split-K grid + ONE fixed-order reduce: the layers of an Inception block finish their dy one after the other, but no weight
 * gradient is consumed before the optimizer (autograd computes them at the same place: aten::convolution_backward under
 * neuston_models.py:81-86) -- together they fill the chip with 1/n of the pixel splits, i.e. 1/n of the fp32 slab traffic and
 * reduce work per layer and n times the K-steps per block.  All members must take the same wide-tile template
 * (bf16, C == Cw, equal channel tile): ifcbk_conv2d_wgrad_group_workspace returns 0 when they do not (then call
 * ifcbk_conv2d_wgrad per layer).  descs / xs / dys / dws are host arrays.  Bitwise reproducible.                          */
IFCBK_API int ifcbk_conv2d_wgrad_group(ifcbk_ctx*, int n, const ifcbk_conv_desc* descs, const void* const* xs, const void* const* dys,
                             float* const* dws, int accumulate, void* stream);
IFCBK_API size_t ifcbk_conv2d_wgrad_group_workspace(int n, const ifcbk_conv_desc* descs);
/* the plan of such a group: channel tile (kh * 32 output channels), blocks of the grid, pixel splits per member          */
IFCBK_API int ifcbk_conv2d_wgrad_group_info(int n, const ifcbk_conv_desc* descs, int* kh, int* blocks, int* nsplit);
/* channel tile (4, 5, 6 = 128 / 160 / 192 output channels per block) a layer would take inside a group, 0 = not a candidate
 * (dtype, padded channels, or a tiling that would multiply more than 1.3x the layer's true K x RSC)                        */
IFCBK_API int ifcbk_conv2d_wgrad_group_member_kh(const ifcbk_conv_desc*);
IFCBK_API int  ifcbk_conv2d_fwd_mblocks(const ifcbk_conv_desc*);   /* rows of bn_part */
/* master fp32 [K][R][S][Cw] -> bf16 shadow [K][R][S][C] (zero padded) and, if wT!=NULL, the flipped
 * transposed dgrad shadow [C][R][S][K].                                                             */
IFCBK_API int ifcbk_weight_pack(ifcbk_ctx*, const ifcbk_conv_desc*, const float* w_master, void* w, void* wT,
                      void* stream);

/* all convs of a model in ONE launch: `items` is a DEVICE array built once by the host */
typedef struct {
    const float* w_master;   /* fp32 [K][R][S][Cw]                                                    */
    void*   w;               /* shadow [K][R][S][C]                                                   */
    void*   wT;              /* flipped/transposed shadow [C][R][S][K] or NULL                        */
    int32_t K, RS, C, Cw;
    int64_t first_block;     /* index of this item's first block; an item has cdiv(K,32)*RS*cdiv(C,32) blocks  */
    int32_t wT_ld;           /* 0: K; >0: row stride of wT (this conv is a K-slice of a fused filter)  */
    int32_t pad_;
} ifcbk_pack_item;
IFCBK_API int ifcbk_weight_pack_multi(ifcbk_ctx*, const ifcbk_pack_item* items_dev, int n_items, int64_t total_blocks, int dtype,
                            void* stream);

/* ------------------------------------------------------------------ BatchNorm (+ReLU, +residual)
 * replaces aten::batch_norm / relu_ inside [TV] BasicConv2d and resnet blocks (neuston_models.py:66-68) */
typedef struct {
    int32_t M;               /* N*H*W pixels                                                          */
    int32_t C;               /* channels                                                              */
    int32_t ldx, ldy;        /* pixel strides of raw conv output / activated output                  */
    int32_t relu;            /* 1: y = max(0, .)                                                      */
    int32_t dtype;
    float   eps, momentum;
} ifcbk_bn_desc;

/* train: part[mblocks][2][C] -> mean/invstd -> scale = g*invstd, shift = b - mean*scale;
 * running_mean/var updated with momentum (unbiased var), as torch.nn.BatchNorm2d does.
 * eval (part==NULL): scale/shift from the running statistics.                                       */
/* part_ld (ifcbk_bn_finalize_ld): row stride of `part` when the BN owns a channel slice of a fused conv's partials */
IFCBK_API int ifcbk_bn_finalize_ld(ifcbk_ctx*, const ifcbk_bn_desc*, const float* part, int mblocks, int part_ld,
                         const float* gamma, const float* beta, float* running_mean, float* running_var,
                         float* mean, float* invstd, float* scale, float* shift, void* stream);
IFCBK_API int ifcbk_bn_finalize(ifcbk_ctx*, const ifcbk_bn_desc*, const float* part, int mblocks,
                      const float* gamma, const float* beta, float* running_mean, float* running_var,
                      float* mean, float* invstd, float* scale, float* shift, void* stream);
/* y = act(x*scale + shift (+ residual)) */
IFCBK_API int ifcbk_bn_apply(ifcbk_ctx*, const ifcbk_bn_desc*, const void* x, const float* scale,
                   const float* shift, const void* residual, int ldr, void* y, void* stream);
/* dz = dy * (y>0 if relu);  dgamma = sum dz*xhat;  dbeta = sum dz;
 * dx = gamma*invstd*(dz - dbeta/M - xhat*dgamma/M); if dres!=NULL: dres (+)= dz (residual branch).
 * dy may alias dx.  Partial sums go through the ctx workspace (deterministic two-stage reduction).
 * dres_accumulate: bit 0 = dres += dz, bit 1 = dx accumulates too (a BatchNorm in FRONT of a conv reading a slice of a
 * concatenation, densenet: the slice's gradient collects the share of every later layer).                */
IFCBK_API int ifcbk_bn_bwd(ifcbk_ctx*, const ifcbk_bn_desc*, const void* x, const void* y, const void* dy, int lddy,
                 const float* gamma, const float* mean, const float* invstd,
                 void* dx, int lddx, void* dres, int lddres, int dres_accumulate,
                 float* dgamma, float* dbeta, int param_accumulate,
                 const float* scale, const float* shift /* nullable: bn_apply's affine; lets the ReLU mask be
                 recomputed from x (y is then not read) when there is no residual */, void* stream);

/* ------------------------------------------------------------------ pooling
 * replaces F.max_pool2d / F.avg_pool2d(count_include_pad=True) / adaptive_avg_pool2d in [TV] graphs  */
typedef struct {
    int32_t N, H, W, C, ldx;
    int32_t R, S, stride_h, stride_w, pad_h, pad_w;
    int32_t P, Q, ldy;
    int32_t dtype;
} ifcbk_pool_desc;
IFCBK_API int ifcbk_maxpool_fwd(ifcbk_ctx*, const ifcbk_pool_desc*, const void* x, void* y, uint8_t* argmax, void* stream);
IFCBK_API int ifcbk_maxpool_bwd(ifcbk_ctx*, const ifcbk_pool_desc*, const void* dy, const uint8_t* argmax, void* dx,
                      int accumulate, void* stream);
IFCBK_API int ifcbk_avgpool_fwd(ifcbk_ctx*, const ifcbk_pool_desc*, const void* x, void* y, void* stream);
IFCBK_API int ifcbk_avgpool_bwd(ifcbk_ctx*, const ifcbk_pool_desc*, const void* dy, void* dx, int accumulate, void* stream);

/* ------------------------------------------------------------------ head: GAP -> dropout -> FC
 * replaces adaptive_avg_pool2d + dropout(0.5) + fc in [TV] Inception3.forward / ResNet.forward       */
typedef struct {
    int32_t N, HW, C, ldx;   /* input [N,HW,C]                                                        */
    int32_t NC;              /* classes                                                               */
    int32_t dtype;
    float   keep_scale;      /* 1/(1-p) when mask!=NULL (2.0 for p=0.5)                               */
} ifcbk_head_desc;
/* feat[N][C] fp32 = mean_HW(x) * (mask? mask*keep_scale : 1); logits[N][NC] = feat @ W^T + b          */
IFCBK_API int ifcbk_head_fwd(ifcbk_ctx*, const ifcbk_head_desc*, const void* x, const uint8_t* mask,
                   const float* W, const float* b, float* feat, float* logits, void* stream);
/* dW (+)= dlogits^T feat; db (+)= sum dlogits; dx = (dlogits @ W) * mask*keep_scale / HW broadcast    */
IFCBK_API int ifcbk_head_bwd(ifcbk_ctx*, const ifcbk_head_desc*, const float* dlogits, const float* feat,
                   const uint8_t* mask, const float* W, float* dW, float* db, void* dx, int lddx,
                   int param_accumulate, void* stream);
/* Bernoulli(keep = 1-p) mask bytes from a counter-based generator (seed, offset)                     */
IFCBK_API int ifcbk_dropout_mask(ifcbk_ctx*, uint8_t* mask, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream);

/* ------------------------------------------------------------------ loss
 * replaces nn.CrossEntropyLoss (mean) neuston_models.py:55,70-78 and softmax(dim=1) :99,:156          */
/* loss_out[0] (+)= weight * mean_i CE(logits_i, target_i); dlogits = weight*(softmax - onehot)/N     */
IFCBK_API int ifcbk_softmax_xent(ifcbk_ctx*, const float* logits, const int64_t* target, int N, int NC, float weight,
                       float* loss_out, int loss_accumulate, float* dlogits, void* stream);
IFCBK_API int ifcbk_softmax(ifcbk_ctx*, const float* logits, int N, int NC, float* probs, void* stream);
/* the bookkeeping of one fused train step, in the step's own op table (no framework kernel between the first and the last
 * launch of a step): num_batches_tracked[0..n) += 1 of every BatchNorm ([PL]/torch: nn.BatchNorm2d.forward in training) and
 * loss_sum += loss (the reference's train_loss is the SUM of the batch losses, neuston_models.py:85).  Either part may be NULL. */
IFCBK_API int ifcbk_step_counters(ifcbk_ctx*, int64_t* num_batches_tracked, int n, float* loss_sum, const float* loss, void* stream);

/* ------------------------------------------------------------------ optimizer
 * replaces torch.optim.Adam(lr=1e-3) neuston_models.py:63-64 (one flat launch instead of 292 loops)  */
IFCBK_API int ifcbk_adam_flat(ifcbk_ctx*, float* p, const float* g, float* m, float* v, int64_t n, float lr,
                    float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                    void* stream);
IFCBK_API int ifcbk_sgd_flat(ifcbk_ctx*, float* p, const float* g, float* mom, int64_t n, float lr, float momentum,
                   float weight_decay, float grad_scale, void* stream);

/* ------------------------------------------------------------------ input path
 * replaces PIL convert('RGB') + transforms.Resize([S,S]) (PIL bilinear, antialiased, 8-bit fixed point)
 * + ToTensor + Normalize: neuston_data.py:342-371, :456-464 (IfcbBinDataset.__getitem__)              */
typedef struct {
    int32_t n_img;           /* ROIs in the batch                                                     */
    int32_t S;               /* output side (299 / 224)                                               */
    int32_t in_channels;     /* 1 (grayscale ROI -> 3 identical channels) or 3 (interleaved RGB)      */
    int32_t out_channels;    /* padded channel count of the NHWC output (8)                            */
    int32_t flip_bits_valid; /* !=0: flips[i] bit0 = vertical flip ('x'), bit1 = horizontal ('y')     */
    int32_t dtype;
    float   mean[3], std[3]; /* Normalize; std = 1, mean = 0 for none                                 */
    float   tin_scale[3], tin_shift[3]; /* [TV] transform_input affine (1,0 when off)                 */
} ifcbk_roi_desc;
/* pixels: concatenated u8 ROIs; offs[i] byte offset, hs[i]/ws[i] dims; out (nullable when out_u8 is given):
 * [n_img,S,S,out_channels]                                                                             */
IFCBK_API int ifcbk_roi_preprocess(ifcbk_ctx*, const ifcbk_roi_desc*, const uint8_t* pixels, const int64_t* offs,
                         const int32_t* hs, const int32_t* ws, const uint8_t* flips, int max_h, int max_w,
                         void* out, uint8_t* out_u8 /*nullable: resized u8 [n,S,S,in_channels]*/, void* stream);
IFCBK_API size_t ifcbk_roi_preprocess_workspace(const ifcbk_roi_desc*, int max_h, int max_w);
/* The stem conv straight from the resized u8 plane (grey ROIs: in_channels = 1).  convert('RGB') + ToTensor + Normalize
 * (+ [TV] transform_input) make three copies of one plane g under per-channel affines x_c = ab[c] * g + ab[3 + c]; the
 * 3x3 / stride-2 / unpadded / 32-channel Conv2d_1a of those is a one-plane conv plus a constant, so the [N,S,S,8] input
 * tensor is never written: g = out_u8 of ifcbk_roi_preprocess ([N][H][W] bytes), w_master = the fp32 master filter
 * [32][3][3][3], ab = 6 floats in device memory.  fp32 arithmetic on exact pixels and master weights; y rounded to d->dtype.
 *   scale == shift == NULL: y = raw conv output, bn_part (nullable) [ifcbk_stem_u8_rows][2][32] partial sums of the rounded
 *   outputs for ifcbk_bn_finalize;  otherwise y = act(conv * scale[k] + shift[k]) (eval-mode folded BatchNorm, relu flag).
 * d: N, H, W of the plane, K = 32, R = S = 3, stride 2, pad 0, P, Q, ldy, Cw = 3 (C / ldx ignored); anything else:
 * IFCBK_EUNSUPPORTED (ifcbk_stem_u8_rows returns 0) and the caller uses ifcbk_conv2d_* on the [N,S,S,8] tensor.
 * wgrad: dw[k][r][s][c] (+)= ab[c] * sum dy[k] g[r,s] + ab[3 + c] * sum dy[k]; deterministic (per-block partials in the ctx
 * workspace, fixed-order fp64 reduce).  dy: [N*P*Q] rows of 32, row stride d->ldy.
 * Replaces aten::conv2d forward / weight gradient of [TV] Inception3.Conv2d_1a_3x3 on the output of the transform chain
 * neuston_data.py:342-371, :456-464 (reference call sites neuston_models.py:66-68, 81-86).                               */
IFCBK_API int ifcbk_stem_u8_rows(const ifcbk_conv_desc*);
IFCBK_API int ifcbk_stem_u8_fwd(ifcbk_ctx*, const ifcbk_conv_desc* d, const uint8_t* g, const float* w_master, const float* ab, void* y,
                      float* bn_part, const float* scale, const float* shift, int relu, void* stream);
IFCBK_API size_t ifcbk_stem_u8_wgrad_workspace(const ifcbk_conv_desc*);
IFCBK_API int ifcbk_stem_u8_wgrad(ifcbk_ctx*, const ifcbk_conv_desc* d, const uint8_t* g, const void* dy, const float* ab, float* dw,
                        int accumulate, void* stream);
/* fp32 NCHW [N,3,H,W] -> NHWC [N,H,W,Cpad] (zero padded), optional per-channel affine (transform_input) */
IFCBK_API int ifcbk_nchw_to_nhwc(ifcbk_ctx*, const float* x, int N, int C, int H, int W, int Cpad, int dtype,
                       const float* scale3, const float* shift3, void* y, void* stream);
IFCBK_API int ifcbk_nhwc_to_nchw_f32(ifcbk_ctx*, const void* x, int N, int C, int H, int W, int ldx, int dtype,
                           float* y, void* stream);

/* BN apply (+ReLU) fused into the 3x3 / stride-2 max pool that is the activation's only consumer (inception
 * Conv2d_2b -> maxpool1, Conv2d_4a -> maxpool2; the resnet stem): y = maxpool(act(x*scale+shift)), values and
 * arg-max exactly those of ifcbk_bn_apply followed by ifcbk_maxpool_fwd, the activation is never written.
 * d: N,H,W,C of the conv output x (ldx = its pixel stride), P,Q,ldy of the pooled output, R=S=3, stride 2.
 * Replaces [TV] `F.relu(bn(conv(x)))` + `F.max_pool2d(x, 3, 2)` (inception.py forward; reference call site
 * neuston_models.py:66-68).                                                                             */
IFCBK_API int ifcbk_bn_apply_maxpool(ifcbk_ctx*, const ifcbk_pool_desc* d, const void* x, const float* scale,
                           const float* shift, int relu, void* y, uint8_t* argmax, void* stream);
/* Batch statistics of a tensor that no conv epilogue produced (an average pool moved BEHIND its 1x1 conv: the BatchNorm
 * input is then the pool's output): partial sums (sum x, sum x^2 of the stored values) per 1024-row tile into
 * part [ifcbk_bn_stats_rows(M)][2][C], to be reduced by ifcbk_bn_finalize like a conv epilogue's partials.
 * Replaces the statistics half of [TV] BasicConv2d's nn.BatchNorm2d in training mode (reference call site
 * neuston_models.py:66-68).                                                                                              */
/* Eval-mode twin of the same reordering: y = act(avgpool3x3(s1,p1)(x) * scale[c] + shift[c]), the pool of a branch that runs
 * as avgpool(conv1x1(x)) with the BatchNorm affine (+ReLU) of [TV] BasicConv2d applied to the pooled value (rounded to the
 * storage type first, as training stores it).  3x3 / stride 1 / pad 1 only.                                             */
IFCBK_API int ifcbk_avgpool3x3_affine(ifcbk_ctx*, const ifcbk_pool_desc* d, const void* x, const float* scale, const float* shift,
                            int relu, void* y, void* stream);
IFCBK_API int ifcbk_bn_stats_rows(int64_t M);
IFCBK_API int ifcbk_bn_stats(ifcbk_ctx*, const ifcbk_bn_desc* d, const void* x, float* part, void* stream);

/* ... and its backward: BN backward whose upstream gradient is maxpool_bwd(dpool, argmax), gathered on the fly
 * (neither the activation nor its gradient exists in memory).  dx: gradient of the conv output (ld lddx). */
IFCBK_API int ifcbk_bn_bwd_maxpool(ifcbk_ctx*, const ifcbk_pool_desc* d, const void* x, const void* dpool,
                         const uint8_t* argmax, const float* gamma, const float* mean, const float* invstd,
                         const float* scale, const float* shift, int relu, void* dx, int lddx, float* dgamma,
                         float* dbeta, int param_accumulate, void* stream);

/* Input gradient of a stride-1 conv whose input is ONE BatchNorm+ReLU activation y = relu(bn(prev_raw)) with no other
 * consumer: the epilogue that writes dx (= d loss / d y) also reduces that BatchNorm's backward sums (sum dz, sum dz*xhat
 * per channel, dz = dx where y > 0) into `part` [mblocks][2][C] -- ifcbk_bn_bwd_partials then skips its reduction pass
 * (one read of prev_raw in the epilogue replaces a pass over prev_raw and dx).  mblocks from ..._mblocks (0: this
 * descriptor has no fused variant -- strided, or served by the row-streaming kernel).  Same results as
 * ifcbk_conv2d_dgrad + ifcbk_bn_bwd up to the summation order of the two per-channel sums.
 * Replaces autograd of [TV] BasicConv2d chains (inception.py), reference call site neuston_models.py:66-68,81-86.  */
IFCBK_API int ifcbk_conv2d_dgrad_bnstat_mblocks(const ifcbk_conv_desc*);
/* The same fusion when dx is the gradient of a CONCATENATION of several BatchNorm+ReLU outputs (an Inception block output whose only
 * consumer is the next block's sibling 1x1 GEMM): `table` (device memory) holds one entry per 8 channels of dx naming that chunk's
 * producer -- `raw`: the producing BatchNorm's input at this chunk's first channel, pixel 0 (pixel stride raw_ld elements); `stat`:
 * that channel's mean, with invstd / scale / shift at +stat_ld, +2 stat_ld, +3 stat_ld floats; raw == NULL: no BatchNorm producer
 * (a pooled slice), its sums are zero.  part [mblocks][2][C]: every producer reads its own column range with row stride C
 * (ifcbk_bn_bwd_partials_ld).  Replaces autograd through [TV] torch.cat(...) of InceptionA/B/C/D/E.forward (reference call site
 * neuston_models.py:66-68,81-86).                                                                                              */
typedef struct {
    const void*  raw;
    const float* stat;
    int32_t      raw_ld;
    int32_t      stat_ld;
} ifcbk_bs_chunk;
IFCBK_API int ifcbk_conv2d_dgrad_bnstat_table(ifcbk_ctx*, const ifcbk_conv_desc*, const void* dy, const void* wT, void* dx,
                                    const ifcbk_bs_chunk* table, float* part, void* stream);
IFCBK_API int ifcbk_conv2d_dgrad_bnstat(ifcbk_ctx*, const ifcbk_conv_desc*, const void* dy, const void* wT, void* dx,
                              const void* prev_raw, int prev_ld, const float* prev_mean, const float* prev_invstd,
                              const float* prev_scale, const float* prev_shift, float* part, void* stream);
/* BatchNorm(+ReLU) backward from those partial sums: finalize + dx only                                        */
IFCBK_API int ifcbk_bn_bwd_partials(ifcbk_ctx*, const ifcbk_bn_desc*, const void* x, const void* dy, int lddy,
                          const float* gamma, const float* mean, const float* invstd, const float* scale,
                          const float* shift, const float* part, int ntiles, void* dx, int lddx, float* dgamma,
                          float* dbeta, int param_accumulate, void* stream);
/* ... with an explicit row stride of `part` (floats; 0 = C): this layer's columns inside a wider partial-sum matrix            */
IFCBK_API int ifcbk_bn_bwd_partials_ld(ifcbk_ctx*, const ifcbk_bn_desc*, const void* x, const void* dy, int lddy,
                             const float* gamma, const float* mean, const float* invstd, const float* scale,
                             const float* shift, const float* part, int ntiles, int part_ld, void* dx, int lddx,
                             float* dgamma, float* dbeta, int param_accumulate, void* stream);

/* ------------------------------------------------------------------ layers without BatchNorm (alexnet, vgg*, squeezenet1_1,
 * densenet*: neuston_models.py:27-36,40-42 -> torchvision alexnet.py / vgg.py / squeezenet.py / densenet.py).  The conv with its
 * bias (+ReLU) is ifcbk_conv2d_fwd_affine with scale = 1, shift = bias; these are the passes around it.                           */
/* autograd of relu_(conv(x) + b) up to the conv: dz = dy * (y > 0) (relu = 0: dz = dy), dbias (+)= column sums of dz.  y, dy, dz:
 * [M] rows of K channels with row strides ldy / lddy / lddz (elements); dz may alias dy and may be NULL (sums only), dbias may be
 * NULL (mask only).  Deterministic two-stage sums through the ctx workspace (ifcbk_bias_relu_bwd_workspace bytes).              */
IFCBK_API int ifcbk_bias_relu_bwd(ifcbk_ctx*, int64_t M, int K, int dtype, const void* y, int ldy, const void* dy, int lddy, void* dz,
                        int lddz, int relu, float* dbias, int param_accumulate, void* stream);
IFCBK_API size_t ifcbk_bias_relu_bwd_workspace(int64_t M, int K);
IFCBK_API int ifcbk_bias_relu_bwd_rows(int64_t M);
/* nn.Dropout forward and backward: y (+)= x * (mask ? mask[i] * scale : 1) over n contiguous elements (mask: one byte each,
 * ifcbk_dropout_mask; NULL = eval mode, a copy)                                                                                    */
IFCBK_API int ifcbk_dropout_apply(ifcbk_ctx*, int64_t n, int dtype, const void* x, const uint8_t* mask, float scale, void* y, int accumulate,
                        void* stream);
/* torch.flatten(x, 1) of an NCHW tensor, on the NHWC activation x [N][HW] rows of C channels (row stride ldx):
 * to_chw = 1: flat[n][c*HW + hw] = x[n][hw][c];  to_chw = 0 (backward): x[n][hw][c] (+)= flat[n][c*HW + hw]                          */
IFCBK_API int ifcbk_flatten_chw(ifcbk_ctx*, int N, int HW, int C, int dtype, void* x, int ldx, void* flat, int to_chw, int accumulate,
                      void* stream);

/* ------------------------------------------------------------------ program runner
 * One call launches a whole forward / backward / update list: the host builds the op table once
 * (static graph), so the per-step host cost is one FFI crossing.                                     */
enum {
    IFCBK_OP_CONV_FWD = 1, IFCBK_OP_CONV_DGRAD, IFCBK_OP_CONV_WGRAD, IFCBK_OP_WEIGHT_PACK,
    IFCBK_OP_BN_FINALIZE, IFCBK_OP_BN_APPLY, IFCBK_OP_BN_BWD,
    IFCBK_OP_MAXPOOL_FWD, IFCBK_OP_MAXPOOL_BWD, IFCBK_OP_AVGPOOL_FWD, IFCBK_OP_AVGPOOL_BWD,
    IFCBK_OP_HEAD_FWD, IFCBK_OP_HEAD_BWD, IFCBK_OP_SOFTMAX_XENT, IFCBK_OP_SOFTMAX,
    IFCBK_OP_ADAM, IFCBK_OP_MEMSET, IFCBK_OP_COPY2D, IFCBK_OP_DROPOUT_MASK, IFCBK_OP_CONV_FWD_AFFINE,
    IFCBK_OP_WEIGHT_PACK_MULTI, IFCBK_OP_CONV_WGRAD_SEG, IFCBK_OP_BN_APPLY_MAXPOOL, IFCBK_OP_BN_BWD_MAXPOOL,
    IFCBK_OP_CONV_DGRAD_BNSTAT, IFCBK_OP_BN_BWD_PARTIALS, IFCBK_OP_BN_STATS, IFCBK_OP_AVGPOOL_AFFINE,
    IFCBK_OP_CONV_FWD_AFFINE_SEG,
    IFCBK_OP_SGD,            /* p: P, G, momentum buffer (nullable); i[0] = n; f: lr, momentum, weight decay, grad scale */
    IFCBK_OP_CONV_DGRAD_BNSTAT_TAB,  /* p: dy, wT, dx, table, part (ifcbk_conv2d_dgrad_bnstat_table)                           */
    IFCBK_OP_BIAS_RELU_BWD,  /* p: y, dy, dz, dbias; u.bn: M, C, ldx = ld(y), ldy = ld(dy), relu, dtype; i[0] = ld(dz)          */
    IFCBK_OP_DROPOUT,        /* p: x, mask (nullable), y; i[0] = n, i[1] = dtype; f[0] = scale; flags bit 0 accumulate           */
    IFCBK_OP_FLATTEN_CHW,    /* p: x, flat; i: N, HW, C, ldx | dtype << 32; flags bit 0 accumulate, bit 2 to_chw                 */
    IFCBK_OP_STEM_U8_FWD,    /* p: g, w_master, ab, y, bn_part (nullable), scale, shift (both NULL: raw + statistics); flags bit 2 relu */
    IFCBK_OP_STEM_U8_WGRAD,  /* p: g, dy, ab, dw; flags bit 0 accumulate                                                           */
    IFCBK_OP_CONV_FWD_AFFINE_MAXPOOL, /* p: x, w, y_pooled, scale, shift; i[0] = ld of y_pooled; flags bit 2 relu                  */
    IFCBK_OP_STEP_COUNTERS,  /* p: num_batches_tracked (i64, nullable), loss_sum (nullable), loss; i[0] = number of BatchNorms      */
    IFCBK_OP_CONV_WGRAD_GROUP /* p[0]: HOST array of i[0] ifcbk_wgrad_item entries, kept alive by the caller; p[1..]: the members' dw again
                               * (what the data-parallel bucket planner reads); flags bit 0 accumulate                            */
};
typedef struct {
    ifcbk_conv_desc d;
    const void* x;
    const void* dy;
    float* dw;
} ifcbk_wgrad_item;
typedef struct {
    int32_t kind;
    int32_t flags;           /* bit 0 accumulate, bit 1 param accumulate, bit 2 relu (per kind), bit 3 dx accumulate (BN_BWD);
                              * bits 8-10 LANE of this op (0 = the caller's stream, 1-7 = ctx-owned streams),
                              * bits 12-19 WAIT mask: lanes whose queued work must finish before this op starts.
                              * All lanes start after the caller's prior work and join lane 0 at program end. */
    void*   p[12];           /* pointer operands in the order of the typed entry point               */
    int64_t i[4];            /* scalar ints (per kind)                                                */
    float   f[8];            /* scalar floats (per kind)                                              */
    union {
        ifcbk_conv_desc conv;
        ifcbk_bn_desc   bn;
        ifcbk_pool_desc pool;
        ifcbk_head_desc head;
    } u;
} ifcbk_op;
/* op_ms (nullable, host array of n floats): when given, every op is bracketed by HIP events on
 * `stream`, the stream is synchronised at the end and per-op milliseconds are returned.             */
IFCBK_API int ifcbk_run_program(ifcbk_ctx*, const ifcbk_op* ops, int n, void* stream, float* op_ms);
/* Non-blocking timing: same launches, with HIP events recorded (on the op's lane) around every op whose flags
 * have bit 7 set, into event slot `slot` (0..255, one slot per in-flight program run); nothing is synchronised.  After the caller has
 * synchronised the stream, ifcbk_program_times returns the n per-op milliseconds of that slot (0 for ops
 * that were not bracketed).                                                                              */
IFCBK_API int ifcbk_run_program_ev(ifcbk_ctx*, const ifcbk_op* ops, int n, void* stream, int slot);
IFCBK_API int ifcbk_program_times(ifcbk_ctx*, int slot, int n, float* op_ms);
/* hipGraph replay of a program (BASELINE config 4: "hipGraph-captured batches").  ifcbk_program_capture records the
 * program's launches -- all lanes, with their fork / wait / join edges -- into a hipGraph through stream capture on a
 * private stream (nothing executes) and instantiates it; ifcbk_graph_launch replays it stream-ordered on `stream`.
 * Every pointer and scalar of the ops is baked in at capture time: the caller keeps the buffers alive and unmoved, and
 * re-captures when a scalar changes (the engine keeps Adam, whose step count changes every step, outside the graph).
 * The per-lane workspace arenas are baked in as well: ifcbk_ctx_reserve growing the workspace invalidates every graph
 * of the ctx (ifcbk_graph_launch then fails with IFCBK_EINVAL instead of replaying stale pointers).  Lifetime: a graph belongs
 * to the ctx it was captured through -- launch / destroy it through that ctx only (anything else is IFCBK_EINVAL), and
 * ifcbk_ctx_destroy destroys the graphs the caller left, BEFORE the arenas / streams / events they were built from.             */
typedef struct ifcbk_graph ifcbk_graph;
IFCBK_API int ifcbk_program_capture(ifcbk_ctx*, const ifcbk_op* ops, int n, ifcbk_graph** out);
IFCBK_API int ifcbk_graph_launch(ifcbk_ctx*, ifcbk_graph*, void* stream);
IFCBK_API int ifcbk_graph_destroy(ifcbk_ctx*, ifcbk_graph*);
/* name of the (dominant) device kernel an op launches, e.g. "conv_igemm_bf16<4>" (as rocprofv3 prints it
 * inside its mangled/demangled symbol); returns 0 and "" for ops without a compute kernel               */
IFCBK_API int ifcbk_op_kernel(const ifcbk_op* op, char* name, size_t cap);
/* algorithmic work of one op: flops (MAC*2 of conv/FC only) and minimum HBM bytes                   */
IFCBK_API int ifcbk_op_cost(const ifcbk_op* op, double* flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif
