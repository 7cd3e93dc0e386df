// Shared device/host helpers for libifcbk (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include "../../include/ifcbk.h"

constexpr int IFCBK_MAX_LANES = 8;
struct ifcbk_ctx {
    int device;
    void* ws;            // workspace arena of the lane that is launching (split-K slabs, BN partials, resize tables)
    size_t ws_bytes;     // per lane
    void* ws_base;       // ws_lanes arenas of ws_bytes each
    int ws_lanes;        // program lanes that own an arena (ifcbk_ctx_set_lanes; programs that use a lane beyond it are refused)
    unsigned ws_epoch;   // bumped when the arenas move: graphs captured before are stale
    hipStream_t lane_st[IFCBK_MAX_LANES];   // lanes 1.. of ifcbk_run_program (lane 0 is the caller's stream)
    hipStream_t cap_st[IFCBK_MAX_LANES];    // the lanes' streams inside a stream capture (never used to launch)
    struct ifcbk_graph* graphs;             // every live graph captured through this ctx (the ctx owns them: ifcbk_ctx_destroy
    int n_graphs;                           // destroys what the caller left -- a graph never outlives the arenas, streams and events it was built from)
    hipEvent_t xev[64];  // cross-lane ordering events, used round-robin
    int n_xev, xev_next;
    hipEvent_t* cev;     // ordering events of stream captures: one per edge, never reused inside a capture
    int n_cev, cev_next, capturing;
    void* zeros;         // 4 KiB of zeros: source address of padded / out-of-range LDS-DMA chunks
    hipEvent_t* ev;      // profiling events for ifcbk_run_program
    int n_ev;
    hipEvent_t* slot_ev[256];   // ifcbk_run_program_ev: (start, stop) per op
    int slot_n[256];
    unsigned char* slot_rec[256];   // which ops of the slot's last run were bracketed (flags bit 7)
    char err[512];
};

#define IFCBK_FAIL(ctx, code, ...)                                   \
    do {                                                             \
        if (ctx) snprintf((ctx)->err, sizeof((ctx)->err), __VA_ARGS__); \
        return (code);                                               \
    } while (0)

#define IFCBK_HIP(ctx, expr)                                                              \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) IFCBK_FAIL(ctx, IFCBK_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define IFCBK_LAUNCH_CHECK(ctx, name)                                                     \
    do {                                                                                  \
        hipError_t e_ = hipGetLastError();                                                \
        if (e_ != hipSuccess) IFCBK_FAIL(ctx, IFCBK_EHIP, "launch %s: %s", name, hipGetErrorString(e_)); \
    } while (0)

typedef unsigned short bf16_t;   // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
// round-to-nearest-even, NaN preserved (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]);
    v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
    return v;
}

// ---- storage-type generic 16-byte chunk (8 bf16 or 4 fp32 channels): every HBM access of the elementwise kernels
template <class T> struct Chunk;
template <> struct Chunk<bf16_t> {
    static constexpr int N = 8;
    __device__ static __forceinline__ void load(const bf16_t* p, float* f) { unpack8(*reinterpret_cast<const uint4*>(p), f); }
    __device__ static __forceinline__ void store(bf16_t* p, const float* f) { *reinterpret_cast<uint4*>(p) = pack8(f); }
    __device__ static __forceinline__ float round(float v) { return bf2f(f2bf(v)); }      // the value as it would be stored
    typedef uint4 raw_t;                                                                    // a chunk as loaded, not yet widened
    __device__ static __forceinline__ raw_t load_raw(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
    __device__ static __forceinline__ void widen(const raw_t& r, float* f) { unpack8(r, f); }
};
template <> struct Chunk<float> {
    static constexpr int N = 4;
    __device__ static __forceinline__ void load(const float* p, float* f) { *reinterpret_cast<float4*>(f) = *reinterpret_cast<const float4*>(p); }
    __device__ static __forceinline__ void store(float* p, const float* f) { *reinterpret_cast<float4*>(p) = *reinterpret_cast<const float4*>(f); }
    __device__ static __forceinline__ float round(float v) { return v; }
    typedef float4 raw_t;
    __device__ static __forceinline__ raw_t load_raw(const float* p) { return *reinterpret_cast<const float4*>(p); }
    __device__ static __forceinline__ void widen(const raw_t& r, float* f) { *reinterpret_cast<float4*>(f) = r; }
};
__device__ __forceinline__ float to_f32(bf16_t v) { return bf2f(v); }
__device__ __forceinline__ float to_f32(float v) { return v; }
template <class T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f2bf(v); }
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
static inline int dtype_esize(int dtype) { return dtype == IFCBK_F32 ? 4 : 2; }
static inline int dtype_chunk(int dtype) { return dtype == IFCBK_F32 ? 4 : 8; }

// exact unsigned division by a runtime constant for n < 2^31 (host builds, device applies)
struct fastdiv_t {
    uint32_t mul, shift, d;
};
static inline fastdiv_t make_fastdiv(uint32_t d) {
    fastdiv_t f;
    f.d = d;
    uint32_t s = 0;
    while ((1ull << s) < d) ++s;
    f.shift = s;
    f.mul = (uint32_t)((((1ull << s) - d) << 32) / d + 1);
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const fastdiv_t& f) {
    return (__umulhi(n, f.mul) + n) >> f.shift;
}

// u8 arg-max of a max pool, one byte per channel, packed per 16-byte activation chunk (8 bf16 / 4 fp32 channels)
template <int E> struct ArgPack;
template <> struct ArgPack<8> {
    __device__ static __forceinline__ void store(uint8_t* p, const int* bi) {
        uint2 v;
        v.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
        v.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
        *reinterpret_cast<uint2*>(p) = v;
    }
    __device__ static __forceinline__ void load(const uint8_t* p, int* bi) {
        uint2 v = *reinterpret_cast<const uint2*>(p);
#pragma unroll
        for (int j = 0; j < 8; ++j) bi[j] = ((j < 4 ? v.x : v.y) >> (8 * (j & 3))) & 0xff;
    }
};
template <> struct ArgPack<4> {
    __device__ static __forceinline__ void store(uint8_t* p, const int* bi) {
        *reinterpret_cast<uint32_t*>(p) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    }
    __device__ static __forceinline__ void load(const uint8_t* p, int* bi) {
        uint32_t v = *reinterpret_cast<const uint32_t*>(p);
#pragma unroll
        for (int j = 0; j < 4; ++j) bi[j] = (v >> (8 * j)) & 0xff;
    }
};

// conv_rows.hip: direct 3x3/stride-1 convolution of the narrow stem layers
bool ifcbk_conv_rows_ok(int dtype, int cin, int cout, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int Wout);
int ifcbk_conv_rows_blocks(int N, int Pout);
int ifcbk_conv_rows_launch(ifcbk_ctx* ctx, int cin, int cout, int N, int H, int W, int ldx, int P, int Q, int ldy, int pad_h,
                           int pad_w, const void* x, const void* w, void* y, float* part, const float* scale,
                           const float* shift, int relu, hipStream_t st);
bool ifcbk_conv_rows_pool_ok(int dtype, int cin, int cout, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int P, int Q);
int ifcbk_conv_rows_pool_launch(ifcbk_ctx* ctx, int N, int H, int W, int ldx, int P, int Q, int pad_h, int pad_w, const void* x,
                                const void* w, void* y, int ldy, const float* scale, const float* shift, int relu, hipStream_t st);
int ifcbk_num_cus();
// conv_big.hip: wide-tile (256/320 pixels x 128..256 channels) ping-pong kernel; plan = does it serve this GEMM, and with which tile
bool ifcbk_conv_big_plan(int dtype, int M, int K, int Kg, int* mt, int* tn);
// persistent wide-tile kernel (conv_pp3.hip): epi 0 = raw store (+ statistics), 1 = eval affine (+ReLU)
bool ifcbk_conv_pp3_plan(int dtype, int M, int K, int Kg, int epi);
int ifcbk_conv_pp3_launch(ifcbk_ctx* ctx, void* conv_args, hipStream_t st);
void ifcbk_conv_pp3_name(int Kg, bool affine, bool plain, char* name, size_t cap);
int ifcbk_conv_big_launch(ifcbk_ctx* ctx, void* conv_args, int mt, int tn, hipStream_t st);
// conv_wgrad_pp.hip: wide-tile ping-pong weight gradient (plan: channel tile 32*kh, pixel splits)
bool ifcbk_wgrad_pp_plan(const ifcbk_conv_desc* d, int* kh, int* nsplit, int* split_len);
int ifcbk_wgrad_pp_launch(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const void* x, const void* dy, float* slab, int kh, int nsplit,
                          int split_len, hipStream_t st);
// conv_wgrad_flat.hip: flat-slot weight gradient (one filter row per block, x slab shared by the row's taps)
bool ifcbk_wgrad_flat_plan(const ifcbk_conv_desc* d, int* nsplit, int* split_len);
int ifcbk_wgrad_flat_launch(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const void* x, const void* dy, float* slab, int nsplit, int split_len,
                            hipStream_t st);
bool ifcbk_wgrad_flat_member(const ifcbk_conv_desc* d);
bool ifcbk_wgrad_flat_group_plan(int n, const ifcbk_conv_desc* ds, int* nsplit, int* split_len, int* tiles, size_t* slab_off, size_t* ws,
                                 int* blocks);
int ifcbk_wgrad_flat_group_launch(ifcbk_ctx* ctx, int n, const ifcbk_conv_desc* ds, const void* const* xs, const void* const* dys,
                                  const int* nsplit, const int* split_len, const size_t* slab_off, hipStream_t st);
// conv_flat.hip: flat-image kernel for stride-1 3x3 / 5x5 layers with 48..96 channels; returns its BatchNorm partial rows (= persistent grid; 0: not served)
int ifcbk_conv_flat_rows(int dtype, int N, int H, int W, int cin, int kout, int R, int S, int ph, int pw, int P, int Q);
int ifcbk_conv_flat_launch(ifcbk_ctx* ctx, void* conv_args, int N, hipStream_t st);
// conv_slab.hip: pixel-slab kernel for stride-1 multi-tap layers (17x17 1x7 / 7x1 ...); plan returns the pixel tile / 32 (0: not served)
int ifcbk_conv_slab_plan(int dtype, int N, int H, int W, int C, int K, int R, int S, int ph, int pw, int P, int Q);
int ifcbk_conv_slab_launch(ifcbk_ctx* ctx, void* conv_args, int N, hipStream_t st);
int ifcbk_conv_fwd_nt(int K, int M);
bool ifcbk_conv_ws_shape(int dtype, int M, int K, int Kg);     // the persistent warp-specialised kernel serves this GEMM shape
int ifcbk_conv_fwd_wm(int M, int K);
void ifcbk_conv_wgrad_shape(const ifcbk_conv_desc* d, int* mt, int* cols);
// Workgroups are dealt round-robin over the 8 XCDs (each with a private 4 MiB L2).  Bijective remap of the linear
// block id so that every XCD works on ONE contiguous range of logical tiles: neighbouring tiles (which share
// input rows / operand panels) then hit the same L2.  Speed only -- any placement is correct.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned xcd = bid & 7u, idx = bid >> 3;
    const unsigned q = nblk >> 3, r = nblk & 7u;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
