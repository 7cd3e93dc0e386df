// Direct 3x3 / stride-1 convolution for the narrow stem layers (32 or 64 input channels): forward of Conv2d_2a / 2b,
// input gradient of both (a 3x3 convolution of dy with the flipped filter).
//
// As an implicit GEMM these layers re-read every input pixel 9 times from L2 for a 128 x 32 tile (25 flop per loaded
// byte; 268-386 TF/s, 1.9 TB/s of L2->LDS traffic -- DESIGN.md 5-r1.1).  Here a block walks down the output rows of one
// image: each INPUT row is brought to LDS once (LDS-DMA, zero-filled outside the image by the buffer range check), stays
// for the three output rows that use it (4-slot ring, the next row is in flight while a row is multiplied), and all nine
// taps are MFMA operands read at shifted pixel addresses.  The filter lives in registers (MFMA A operand, loaded once
// per block), an input fragment feeds COUT/16 MFMAs.  Epilogue as in conv_igemm: output row through LDS, 16-byte stores,
// BatchNorm statistics of the ROUNDED outputs (one partial row per block), optional eval-BN affine + ReLU.
//
// Round 5: the row loop no longer serialises on memory.  It used `__syncthreads()` (which waits for EVERY outstanding memory operation
// of the wave, so each row waited for the write acknowledgements of the row before it and for the prefetch it had just issued) and
// plain LDS accesses (in front of which the compiler puts s_waitcnt vmcnt(0) while any LDS-DMA is pending): 6,000 cycles per row and
// block for 900 cycles of MFMA.  Now: raw s_barrier; inline-asm ds_read_b128 / ds_write_b64 for every LDS access inside the loop; the
// output row leaves through buffer stores with an out-of-range offset for idle lanes, so that every wave issues exactly KQ (POOL: NI,
// on the rows that complete a pooled row) store instructions per row and the wait for the next input row at the top of the loop is
// COUNTED -- s_waitcnt vmcnt(KQ): everything but the previous row's stores (loads and stores retire in order on one counter).
#include "common.h"
#include <stdlib.h>

namespace {

struct RowsArgs {
    const void* x;
    const void* w;          // [COUT][3][3][CIN]
    void* y;
    float* part;            // [blocks][2][COUT] or null
    const float* ep_scale;
    const float* ep_shift;
    int ep_relu;
    unsigned xbytes, ybytes;
    int N, H, W, ldx, P, Q, ldy, ph, pw;
    int rseg, nseg, mtiles;
    int Pp, Qp;             // POOL: the pooled output [N,Pp,Qp,ldy] is what is written
};

typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned u32x2r_t __attribute__((ext_vector_type(2)));
#define ROWS_DSREAD(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))
#define ROWS_DSREAD_OFF(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
#define ROWS_DSWRITE64_OFF(addr, val, OFF) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(OFF) : "memory")
constexpr int RSEG = 16;        // output rows per block
constexpr int MT_MAX = 10;      // 16-pixel tiles per output row (Q <= 160)
constexpr int NPX = 16 * MT_MAX + 4;   // pixels of an LDS row image: input columns -2 .. 16*MT_MAX+1

// POOL (eval only): the conv's folded-BatchNorm + ReLU output feeds ONLY a 3x3 / stride-2 / unpadded max pool (Conv2d_2b -> maxpool1):
// the activated row never leaves the block -- each thread keeps the running column-pooled maximum of its (pooled pixel, channel chunk)
// items in registers over the three conv rows of a pooled row and writes the pooled row; a block owns 8 pooled rows = 17 conv rows (the
// last one is the next block's first: recomputed).  Bit-identical to the affine epilogue followed by ifcbk_maxpool_fwd (rounding to the
// storage type is monotone), without the 708 MB activation write and read per batch of 256.
// EPI: 0 raw store (input gradients), 1 raw store + BatchNorm statistics (training forward), 2 folded-BatchNorm affine (+ReLU) (eval) --
// a template parameter so that the statistics / coefficient registers exist only where they are used (<64, 32> spilled otherwise)
template <int CIN, int COUT, int EPI, bool POOL = false>
__global__ __launch_bounds__(256, 2) void conv_rows3x3(RowsArgs a) {
    static_assert(!POOL || EPI == 2, "the pooled form is the eval epilogue");
    constexpr int CPP = CIN / 8;                     // 16-byte chunks per pixel
    constexpr int NTL = COUT / 16;                   // output-channel tiles
    constexpr int KG = CIN / 32;                     // MFMA k groups per tap
    constexpr int LDC = COUT + 8;
    constexpr int CPO = COUT / 8;                    // chunks per output pixel
    constexpr int NDMA = (NPX * CPP + 63) / 64;      // LDS-DMA wave-instructions per input row
    constexpr int JD = (NDMA + 3) / 4;               // per wave
    constexpr int SLOT = NDMA * 512;                 // elements per row slot: whole DMA pieces (the tail piece writes zeros)
    // 4 slots: the next input row is in flight while a row is multiplied (a fifth slot, two rows in flight, measured no better:
    // 21.21 vs 21.19 ms per step).  64 input channels: 3 slots (the next row is requested right after the multiply and lands during
    // the epilogue) so that two blocks still fit a CU
    constexpr int NSLOT = CIN == 32 ? 4 : 3;
    __shared__ __attribute__((aligned(16))) bf16_t sRow[NSLOT * SLOT];
    __shared__ __attribute__((aligned(16))) bf16_t sC[16 * MT_MAX * LDC];
    __shared__ float sRed[4][2][COUT];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int bid = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int n = bid / a.nseg;
    const int p0 = (bid - n * a.nseg) * a.rseg;
    const int p1 = min(p0 + a.rseg + (POOL ? 1 : 0), a.P);
    constexpr int NI = POOL ? (((16 * MT_MAX - 3) / 2 + 1) * (COUT / 8) + 255) / 256 : 1;      // pooled items of a thread
    uint4 run[NI];                // running maxima of a thread's pooled items, packed bf16 (the values are bf16 already): 4 registers each

    // ---- filter fragments (MFMA A operand): lane (row l&15, k group l>>4) of tile nt holds 8 input channels of a tap
    const int frow = lane & 15, fkg = lane >> 4;
    bf16x8_t wf[9][NTL][KG];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
            for (int kg = 0; kg < KG; ++kg)
                wf[tap][nt][kg] = *reinterpret_cast<const bf16x8_t*>((const bf16_t*)a.w + ((size_t)(nt * 16 + frow) * 9 + tap) * CIN + kg * 32 + fkg * 8);

    // ---- LDS-DMA of one input row: lane -> (pixel, chunk); LDS pixel index = input column + 2
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    int dcol[JD];                 // byte offset of this lane's chunk inside an input row, or -1
    bool dlive[JD];               // this wave issues instruction j
#pragma unroll
    for (int j = 0; j < JD; ++j) {
        const int inst = j * 4 + wave;
        dlive[j] = inst < NDMA;
        const int idx = inst * 64 + lane;            // position in the LDS image (chunk units)
        const int px = idx / CPP;
        const int phys = idx - px * CPP;
        // source-side swizzle: the lane that lands on (px, phys) fetches logical chunk phys ^ f(px)
        const int logical = CPP == 4 ? (phys ^ (((px >> 2) & 1) << 1)) : (phys ^ (px & 7));
        const int wcol = px - 2;
        dcol[j] = (px < NPX && wcol >= 0 && wcol < a.W) ? (wcol * a.ldx + logical * 8) * 2 : -1;
    }
    const int img = n * a.H;
#define ISSUE_ROW(hin)                                                                                          \
    {                                                                                                           \
        const int h_ = (hin);                                                                                   \
        const bool rowok = h_ >= 0 && h_ < a.H;                                                                 \
        const unsigned rbase = (unsigned)((img + h_) * a.W * a.ldx) * 2u;                                       \
        bf16_t* dst = sRow + ((h_ + 6) % NSLOT) * SLOT;                                                         \
        _Pragma("unroll") for (int j = 0; j < JD; ++j)                                                          \
            if (dlive[j]) {                                                                                     \
                const unsigned vo = (rowok && dcol[j] >= 0) ? rbase + (unsigned)dcol[j] : OOB;                  \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + (j * 4 + wave) * 512), 16, vo, 0, 0, 0); \
            }                                                                                                   \
    }

    // ---- per-lane LDS byte offsets: pixel fragments (tile mi, column shift s, k group kg) inside a row slot; the tile's C rows
    // pixel of lane = mt*16 + frow + s - pw + 2  (LDS index); chunk = kg*4 + fkg, swizzled by the pixel
    const unsigned sRowB = (unsigned)(size_t)(lptr_t)sRow, sCB = (unsigned)(size_t)(lptr_t)sC;
    // (tile mi of a wave is 64 pixels behind tile mi - 1 -- a multiple of the swizzle period: an immediate offset)
    unsigned foff[3][KG];
    const unsigned cwoff = sCB + (unsigned)(((wave * 16 + frow) * LDC + 4 * fkg) * 2);
#pragma unroll
    for (int sx = 0; sx < 3; ++sx) {
        const int px = wave * 16 + frow + sx - a.pw + 2;
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            const int ch = kg * 4 + fkg;
            const int phys = CPP == 4 ? (ch ^ (((px >> 2) & 1) << 1)) : (ch ^ (px & 7));
            foff[sx][kg] = (unsigned)((px * CIN + phys * 8) * 2);
        }
    }
    // ---- the output row: thread t owns channel chunk t % CPO of pixels t / CPO + k * (256 / CPO); KQ store instructions per row
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.ybytes, 0x00020000);
    constexpr int KQ = (16 * MT_MAX * CPO + 255) / 256;
    const int enn = (t % CPO) * 8;
    float esc[EPI == 2 ? 8 : 1], esh[EPI == 2 ? 8 : 1];         // eval coefficients of this thread's chunk: loaded once, not per row behind the row's stores
    if (EPI == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { esc[j] = a.ep_scale[enn + j]; esh[j] = a.ep_shift[enn + j]; }
    }
    float s1[EPI == 1 ? 8 : 1], s2[EPI == 1 ? 8 : 1];
    if (EPI == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
    }

    const int h_first = p0 - a.ph;
    ISSUE_ROW(h_first)
    ISSUE_ROW(h_first + 1)
    ISSUE_ROW(h_first + 2)
    for (int p = p0; p < p1; ++p) {
        // Row p+2 was requested one iteration ago, in front of that iteration's stores: it has landed when nothing but those stores
        // is outstanding (loads and stores retire in order on one counter)
        const int prel = p - p0 - 1;                  // POOL: the previous row stored iff it completed a pooled row (even, > 0)
        const bool prev_stored = POOL ? (prel > 0 && !(prel & 1)) : p > p0;
        if (prev_stored) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(POOL ? NI : KQ) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // rows p-ph .. p-ph+2 are in LDS; every wave is done reading sC of the previous row
        if (NSLOT == 4 && p + 1 < p1) ISSUE_ROW(p - a.ph + 3)        // prefetch: its slot held row p-ph-1, no longer needed
        unsigned sb[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) sb[r] = sRowB + (unsigned)(((p - a.ph + r + 6) % NSLOT) * SLOT * 2);
#pragma unroll
        for (int mi = 0; mi < 3; ++mi) {
            const int mt = wave + 4 * mi;
            if (mt >= a.mtiles) break;
            f32x4_t acc[NTL];
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) acc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            // fragments of filter row r + 1 are requested before the MFMAs of row r
            bf16x8_t xb[2][3][KG];
#pragma unroll
            for (int sx = 0; sx < 3; ++sx)
#pragma unroll
                for (int kg = 0; kg < KG; ++kg) ROWS_DSREAD_OFF(xb[0][sx][kg], sb[0] + foff[sx][kg], mi * 64 * CIN * 2);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                if (r < 2) {
#pragma unroll
                    for (int sx = 0; sx < 3; ++sx)
#pragma unroll
                        for (int kg = 0; kg < KG; ++kg) ROWS_DSREAD_OFF(xb[(r + 1) & 1][sx][kg], sb[r + 1] + foff[sx][kg], mi * 64 * CIN * 2);
                    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(3 * KG) : "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int sx = 0; sx < 3; ++sx)
#pragma unroll
                    for (int kg = 0; kg < KG; ++kg)
#pragma unroll
                        for (int nt = 0; nt < NTL; ++nt)
                            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[r * 3 + sx][nt][kg], xb[r & 1][sx][kg], acc[nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // lane holds channels nt*16 + 4*(lane>>4) .. +3 of pixel mt*16 + frow
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) {
                const u32x2r_t u = {pack2bf(acc[nt][0], acc[nt][1]), pack2bf(acc[nt][2], acc[nt][3])};
                ROWS_DSWRITE64_OFF(cwoff, u, mi * 64 * LDC * 2 + nt * 32);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // the C row is complete; every wave is past its multiply
        if (NSLOT == 3 && p + 1 < p1) ISSUE_ROW(p - a.ph + 3)        // the slot of row p-ph is free
        // ---- output row: 16-byte chunks, thread t always owns channel chunk t % CPO
        if (POOL) {
            const int rel = p - p0;
            u32x4_t cv[NI][3];
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int qq = (t + 256 * k) / CPO;
                const int qs = qq < a.Qp ? qq : 0;
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) ROWS_DSREAD(cv[k][sx], sCB + (unsigned)(((2 * qs + sx) * LDC + enn) * 2));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int qq = (t + 256 * k) / CPO;
                const bool live = qq < a.Qp;
                float h[8];
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) {
                    float fv[8];
                    unpack8(uint4{cv[k][sx][0], cv[k][sx][1], cv[k][sx][2], cv[k][sx][3]}, fv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float v = fv[j] * esc[j] + esh[j];
                        if (a.ep_relu) v = fmaxf(v, 0.f);
                        v = Chunk<bf16_t>::round(v);
                        h[j] = sx ? fmaxf(h[j], v) : v;
                    }
                }
                if (rel) {
                    float rv[8];
                    unpack8(run[k], rv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) rv[j] = fmaxf(rv[j], h[j]);
                    if (rel & 1) {
                        run[k] = pack8(rv);
                    } else {
                        const uint4 o = pack8(rv);
                        const unsigned off = live ? (unsigned)((((size_t)(n * a.Pp + (p >> 1) - 1) * a.Qp + qq) * a.ldy + enn) * 2) : OOB;
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{o.x, o.y, o.z, o.w}, rsY, off, 0, 0);
                        run[k] = pack8(h);
                    }
                } else {
                    run[k] = pack8(h);
                }
            }
        } else {
            const unsigned yrow = (unsigned)((((size_t)(n * a.P + p) * a.Q) * a.ldy + enn) * 2);
            u32x4_t cv[KQ];
#pragma unroll
            for (int k = 0; k < KQ; ++k) {
                const int q = t / CPO + k * (256 / CPO);
                ROWS_DSREAD(cv[k], sCB + (unsigned)((((q < a.Q) ? q : 0) * LDC + enn) * 2));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < KQ; ++k) {
                const int q = t / CPO + k * (256 / CPO);
                const bool live = q < a.Q;
                float fv[8];
                unpack8(uint4{cv[k][0], cv[k][1], cv[k][2], cv[k][3]}, fv);
                if (EPI == 1 && live) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { s1[j] += fv[j]; s2[j] += fv[j] * fv[j]; }
                }
                u32x4_t o = cv[k];
                if (EPI == 2) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        fv[j] = fv[j] * esc[j] + esh[j];
                        if (a.ep_relu) fv[j] = fmaxf(fv[j], 0.f);
                    }
                    const uint4 pk = pack8(fv);
                    o = u32x4_t{pk.x, pk.y, pk.z, pk.w};
                }
                __builtin_amdgcn_raw_buffer_store_b128(o, rsY, live ? yrow + (unsigned)(q * a.ldy * 2) : OOB, 0, 0);
            }
        }
    }
#undef ISSUE_ROW
    if (EPI == 1) {
        // threads with the same channel chunk (t % CPO) -> one partial row per block, fixed order
#pragma unroll
        for (int off = CPO; off < 64; off <<= 1)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s1[j] += __shfl_xor(s1[j], off);
                s2[j] += __shfl_xor(s2[j], off);
            }
        __syncthreads();
        if (lane < CPO) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sRed[wave][0][lane * 8 + j] = s1[j];
                sRed[wave][1][lane * 8 + j] = s2[j];
            }
        }
        __syncthreads();
        for (int i = t; i < 2 * COUT; i += 256) {
            const int which = i / COUT, c = i - which * COUT;
            a.part[((size_t)bid * 2 + which) * COUT + c] = (sRed[0][which][c] + sRed[1][which][c]) + (sRed[2][which][c] + sRed[3][which][c]);
        }
    }
}

bool enabled() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("IFCBK_CONV_ROWS"); on = e ? atoi(e) : 1; }
    return on != 0;
}

}  // namespace

// cin/cout: channels of the tensor that is READ / WRITTEN by this launch (swapped roles for the input gradient)
bool ifcbk_conv_rows_ok(int dtype, int cin, int cout, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int Wout) {
    if (!enabled() || dtype != IFCBK_BF16 || R != 3 || S != 3 || stride_h != 1 || stride_w != 1) return false;
    if (!((cin == 32 && cout == 32) || (cin == 32 && cout == 64) || (cin == 64 && cout == 32))) return false;
    return pad_h >= 0 && pad_h <= 2 && pad_w >= 0 && pad_w <= 2 && Wout <= 16 * MT_MAX && Wout >= 1;
}

int ifcbk_conv_rows_blocks(int N, int Pout) { return N * ((Pout + RSEG - 1) / RSEG); }

// x: [N,H,W,ldx] read with `cin` channels; y: [N,P,Q,ldy] written with `cout` channels; w: [cout][3][3][cin]
int ifcbk_conv_rows_launch(ifcbk_ctx* ctx, int cin, int cout, int N, int H, int W, int ldx, int P, int Q, int ldy, int pad_h,
                           int pad_w, const void* x, const void* w, void* y, float* part, const float* scale,
                           const float* shift, int relu, hipStream_t st) {
    RowsArgs a;
    a.x = x; a.w = w; a.y = y; a.part = part; a.ep_scale = scale; a.ep_shift = shift; a.ep_relu = relu;
    a.xbytes = (unsigned)((int64_t)N * H * W * ldx * 2);
    if ((int64_t)N * P * Q * ldy * 2 >= (1ll << 31)) IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv_rows3x3: the output exceeds the 2 GiB descriptor window");
    a.ybytes = (unsigned)((int64_t)N * P * Q * ldy * 2);
    a.N = N; a.H = H; a.W = W; a.ldx = ldx; a.P = P; a.Q = Q; a.ldy = ldy; a.ph = pad_h; a.pw = pad_w;
    a.rseg = RSEG; a.nseg = (P + RSEG - 1) / RSEG; a.mtiles = (Q + 15) / 16;
    a.Pp = a.Qp = 0;
    const dim3 grid(N * a.nseg), block(256);
    if ((part != nullptr) && (scale != nullptr)) IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv_rows3x3: statistics and the affine epilogue exclude each other");
    if ((scale != nullptr) != (shift != nullptr)) IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv_rows3x3: scale and shift come together");
    const int epi = part ? 1 : (scale ? 2 : 0);
#define IFCBK_ROWS_LAUNCH(CI, CO)                                                                       \
    {                                                                                                   \
        if (epi == 1) hipLaunchKernelGGL((conv_rows3x3<CI, CO, 1>), grid, block, 0, st, a);             \
        else if (epi == 2) hipLaunchKernelGGL((conv_rows3x3<CI, CO, 2>), grid, block, 0, st, a);        \
        else hipLaunchKernelGGL((conv_rows3x3<CI, CO, 0>), grid, block, 0, st, a);                      \
    }
    if (cin == 32 && cout == 32) IFCBK_ROWS_LAUNCH(32, 32)
    else if (cin == 32 && cout == 64) IFCBK_ROWS_LAUNCH(32, 64)
    else IFCBK_ROWS_LAUNCH(64, 32)
#undef IFCBK_ROWS_LAUNCH
    IFCBK_LAUNCH_CHECK(ctx, "conv_rows3x3");
    return 0;
}

// eval: conv (3x3 / stride 1) + folded BatchNorm affine (+ReLU) + max pool 3x3 / stride 2 / no padding in one pass (32 -> 64 channels)
bool ifcbk_conv_rows_pool_ok(int dtype, int cin, int cout, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int P, int Q) {
    return ifcbk_conv_rows_ok(dtype, cin, cout, R, S, stride_h, stride_w, pad_h, pad_w, Q) && cin == 32 && cout == 64 && P >= 3 && Q >= 3;
}

int ifcbk_conv_rows_pool_launch(ifcbk_ctx* ctx, int N, int H, int W, int ldx, int P, int Q, int pad_h, int pad_w, const void* x,
                                const void* w, void* y, int ldy, const float* scale, const float* shift, int relu, hipStream_t st) {
    RowsArgs a;
    a.x = x; a.w = w; a.y = y; a.part = nullptr; a.ep_scale = scale; a.ep_shift = shift; a.ep_relu = relu;
    a.xbytes = (unsigned)((int64_t)N * H * W * ldx * 2);
    a.N = N; a.H = H; a.W = W; a.ldx = ldx; a.P = P; a.Q = Q; a.ldy = ldy; a.ph = pad_h; a.pw = pad_w;
    a.Pp = (P - 3) / 2 + 1; a.Qp = (Q - 3) / 2 + 1;
    if ((int64_t)N * a.Pp * a.Qp * ldy * 2 >= (1ll << 31)) IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv_rows3x3(pool): the output exceeds the 2 GiB descriptor window");
    a.ybytes = (unsigned)((int64_t)N * a.Pp * a.Qp * ldy * 2);
    a.rseg = RSEG; a.nseg = (a.Pp + RSEG / 2 - 1) / (RSEG / 2); a.mtiles = (Q + 15) / 16;
    hipLaunchKernelGGL((conv_rows3x3<32, 64, 2, true>), dim3(N * a.nseg), dim3(256), 0, st, a);
    IFCBK_LAUNCH_CHECK(ctx, "conv_rows3x3(pool)");
    return 0;
}
