// Flat-image convolution for the stride-1 layers of the 35x35 stage (Mixed_5b/c/d, Mixed_6a's inputs: 3x3 and 5x5 filters over
// 48..96 channels), forward and input gradient (the input gradient of a stride-1 conv is a stride-1 conv of dy with the flipped,
// transposed filter), bf16 storage, fp32 accumulate.
//
// Why not the implicit GEMM: conv_igemm gathers every input pixel once per TAP (9x / 25x) through the CU's global->LDS path; with
// 64..96 output channels per tile that path, not the matrix pipe, sets the speed (56 flop per gathered byte; 14-19 % MFMA busy).
// Here the input is laid out FLAT: image n, input row h, column w live at slot
//     F = (n*Hp + h + ph)*Wp + (w + pw),   Hp = H + ph, Wp = W + pw
// (one shared band of zero slots between rows / images serves as right+left and bottom+top padding), an output pixel (p, q) is
// slot g = (n*Hp + p)*Wp + q, and tap (r, s) of ANY output slot reads slot g + r*Wp + s: every tap is a constant shift.  A block
// takes a SEGMENT of MS consecutive output slots, brings the MS + (R-1)*Wp + S-1 input slots it needs to LDS ONCE (LDS-DMA,
// zeros outside the image through the buffer range check) and reads all R*S*C/32 pixel fragments of the reduction from that
// resident image at shifted addresses; only the filter (L2-resident, k-contiguous rows) streams through a small ring.
// Global->LDS traffic per MAC: 1/5 of the implicit GEMM's (3x3, 96 -> 96: 266 KB against 1.5 MB per 448 pixels).
// Outputs at padding slots (q >= Q or p >= P: 5-11 % of the slots) are computed and dropped.
//
// Block: 512 threads = WM x WN waves, wave (wm, wn) owns MT pixel tiles x NT = K/16/WN channel tiles of 16x16 (filter = MFMA A
// operand, pixels = B operand: a lane ends with 4 consecutive channels of one pixel, as in conv_igemm).  Epilogue: the valid
// pixels of the segment are consecutive output rows; the tile goes to LDS compacted and conv_common.h's epilogue writes it
// (BatchNorm statistics, eval affine, accumulate, the BN-backward sums of MODE 3) with one partial row per SEGMENT.
#include "conv_common.h"
#include <stdlib.h>

namespace {

struct FlatArgs {
    int Hp, Wp, HpWp, ph, pw, N, G;   // G = N*Hp*Wp output slots
    fastdiv_t fHW, fW;
};

// physical 16-byte chunk of logical chunk ch of LDS pixel px: pixel strides that are odd multiples of 32 B (48, 80 channels)
// are conflict-free for ds_read_b128 as they stand; 128-byte and 192-byte pixels need an XOR (an involution: the LDS-DMA
// applies it on the source side, the fragment read on the address)
template <int CPP>
__device__ __forceinline__ int flat_swz(int px, int ch) {
    if constexpr (CPP == 8) return ch ^ (px & 7);
    else if constexpr (CPP == 12) return (ch & ~3) | ((ch & 3) ^ ((px >> 1) & 3));
    else return ch;
}

template <int CIN, int KOUT, int R, int S, int WM, int WN, int MT, int NSTAGE, int WPMAX, int MODE>
__global__ __launch_bounds__(512) void conv_flat(ConvArgs a, FlatArgs f) {
    typedef bf16_t T;
    typedef bf16x8_t frag_t;
    constexpr int CPP = CIN / 8;                      // 16-byte chunks per pixel
    constexpr int NW = WM * WN;
    static_assert(NW == 8 && KOUT % (16 * WN) == 0 && CIN % 8 == 0, "tile");
    constexpr int NT = KOUT / 16 / WN;
    constexpr int MS = 16 * MT * WM;                  // output slots per segment
    constexpr int JR = S * CPP;                       // chunks per filter row: contiguous in the flat image AND in the filter
    constexpr int NCH = R * JR;                       // chunks of the reduction
    constexpr int NKS = (NCH + 3) / 4;                // MFMA steps (32 k)
    constexpr int NK = (NKS + 1) / 2;                 // ring stages (64 k)
    constexpr int D = NSTAGE - 1;
    constexpr int HALO = (R - 1) * WPMAX + S - 1;
    constexpr int NPIECE = ((MS + HALO) * CPP + 63) / 64;
    constexpr int SEG_BYTES = NPIECE * 1024;
    constexpr int BSTAGE = KOUT * 128;                // bytes of a filter stage: KOUT rows x 64 k
    constexpr int NBP = KOUT / 8;                     // LDS-DMA pieces per stage (8 rows each)
    constexpr int JB = (NBP + NW - 1) / NW;           // per wave
    constexpr int RING_BYTES = NSTAGE * BSTAGE;
    constexpr int LDC = KOUT + 8;
    constexpr int CT_BYTES = MS * LDC * 2 + NW * 2 * KOUT * 4;
    constexpr int MAIN_BYTES = SEG_BYTES + RING_BYTES > CT_BYTES ? SEG_BYTES + RING_BYTES : CT_BYTES;
    constexpr int DUMMY_BYTES = (JB * NW > NBP) ? 1024 : 0;
    static_assert(MAIN_BYTES + DUMMY_BYTES <= 160 * 1024, "LDS");
    static_assert(JR >= 4, "a k-step spans at most two filter rows");
    __shared__ __attribute__((aligned(16))) unsigned char smem[MAIN_BYTES + DUMMY_BYTES];
    T* sSeg = reinterpret_cast<T*>(smem);
    T* sRing = reinterpret_cast<T*>(smem + SEG_BYTES);
    T* sC = reinterpret_cast<T*>(smem);
    float* sRed = reinterpret_cast<float*>(smem + MS * LDC * 2);
    T* sDummy = reinterpret_cast<T*>(smem + MAIN_BYTES);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int seg = (int)xcd_remap(blockIdx.x, gridDim.x);
    const unsigned g0 = (unsigned)seg * MS;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.wbytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // ---- the segment's input image: slots g0 .. g0 + MS + halo, one LDS-DMA piece = 64 chunks in image order
    {
        const int halo = (R - 1) * f.Wp + S - 1;
        const int np = ((MS + halo) * CPP + 63) >> 6;
        for (int pc = wave; pc < np; pc += NW) {
            const int idx = pc * 64 + lane;
            const int lpx = idx / CPP;
            const int phys = idx - lpx * CPP;
            const int logical = flat_swz<CPP>(lpx, phys);
            const unsigned F = g0 + (unsigned)lpx;
            const unsigned n = fdiv(F, f.fHW);
            const unsigned rem = F - n * (unsigned)f.HpWp;
            const unsigned hr = fdiv(rem, f.fW);
            const int w = (int)(rem - hr * (unsigned)f.Wp) - f.pw;
            const int h = (int)hr - f.ph;
            const bool v = n < (unsigned)f.N && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
            const unsigned voff = v ? (unsigned)((((int)n * a.H + h) * a.W + w) * a.ldx + logical * 8) * 2u : OOB;
            lds_dma16(rsA, (lptr_t)(smem + pc * 1024), voff, 0);
        }
    }

    // ---- filter ring: stage = [KOUT rows][8 chunks], chunk c of row r at physical chunk c ^ (r & 7) (source-side swizzle)
    const int lrow8 = lane >> 3;
    const int csrc = (lane & 7) ^ lrow8;
    unsigned woff[JB];
    bool glive[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        const int grp = j * NW + wave;
        glive[j] = grp < NBP;
        woff[j] = glive[j] ? (unsigned)((grp * 8 + lrow8) * a.Kg + csrc * 8) * 2u : OOB;
    }
    const bool btail_ok = (NK - 1) * 8 + csrc < NCH;   // this lane's chunk of the LAST stage lies inside the filter row
#define FLAT_ISSUE_B(kt, stage)                                                                             \
    {                                                                                                       \
        const bool cut = (kt) == NK - 1 && !btail_ok;                                                       \
        _Pragma("unroll") for (int j = 0; j < JB; ++j) {                                                    \
            T* dst = glive[j] ? sRing + (stage) * (BSTAGE / 2) + (j * NW + wave) * 512 : sDummy;            \
            lds_dma16(rsB, (lptr_t)dst, cut ? OOB : woff[j], (kt) * 128);                                   \
        }                                                                                                   \
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < NK) FLAT_ISSUE_B(d, d)

    f32x4_t acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    // filter fragments: row wn*NT*16 + nt*16 + frow of the stage, chunk (kk*4 + fq) ^ (frow & 7)
    const T* fB[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fB[kk] = sRing + (wn * NT * 16 + frow) * 64 + (((kk * 4 + fq) ^ (frow & 7)) * 8);
    // pixel fragments: chunk ci = 4*step + fq of the reduction = chunk jj of filter row rr, i.e. slot offset rr*Wp + jj / CPP
    const int pw0 = wm * MT * 16 + frow;
    int jj = fq, rr = 0;

#define FLAT_STEP(kk, soff, LAST)                                                                           \
    {                                                                                                       \
        const int dpx = jj / CPP;                                                                           \
        const int ch = jj - dpx * CPP;                                                                      \
        const bool dead = (LAST) && rr >= R;                                                                \
        const int px = dead ? 0 : pw0 + rr * f.Wp + dpx;                                                    \
        const T* pa = sSeg + (px * CPP + flat_swz<CPP>(px, ch)) * 8;                                        \
        frag_t fa[MT], fb[NT];                                                                              \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) fa[mt] = *reinterpret_cast<const frag_t*>(pa + mt * 16 * CPP * 8); \
        const T* pb = fB[kk] + (soff);                                                                      \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) fb[nt] = *reinterpret_cast<const frag_t*>(pb + nt * 16 * 64); \
        if (LAST) {                                                                                         \
            if (dead) {                                                                                     \
                _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) fa[mt] = frag_t{};  \
            }                                                                                               \
        }                                                                                                   \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                   \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                               \
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[nt][mt], 0, 0, 0); \
        jj += 4;                                                                                            \
        if (jj >= JR) { jj -= JR; ++rr; }                                                                   \
    }

    int stage = 0, istage = D % NSTAGE;
    for (int kt = 0; kt < NK; ++kt) {
        // stage kt (and, at kt = 0, the segment image issued before it) has landed once all but the newer stages are done
        const int newer = NK - 1 - kt < D - 1 ? NK - 1 - kt : D - 1;
        if (D >= 3 && newer >= 2) wait_vmcnt<2 * JB>();
        else if (D >= 2 && newer >= 1) wait_vmcnt<JB>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + D < NK) FLAT_ISSUE_B(kt + D, istage)
        const int soff = stage * (BSTAGE / 2);
        if (kt < NK - 1) {
            FLAT_STEP(0, soff, false)
            FLAT_STEP(1, soff, false)
        } else {
            FLAT_STEP(0, soff, true)
            if (NKS % 2 == 0) FLAT_STEP(1, soff, true)
        }
        stage = stage + 1 == NSTAGE ? 0 : stage + 1;
        istage = istage + 1 == NSTAGE ? 0 : istage + 1;
    }
#undef FLAT_STEP
#undef FLAT_ISSUE_B
    __syncthreads();            // image and ring are consumed: the epilogue reuses them as the C tile

    // ---- epilogue.  cnt(g) = valid output pixels at slots < g: the segment's valid pixels are rows cnt(g0) .. cnt(g0+MS)-1
    auto cnt = [&](unsigned g) -> int {
        if (g >= (unsigned)f.G) return a.M;
        const unsigned n = fdiv(g, f.fHW);
        const unsigned rem = g - n * (unsigned)f.HpWp;
        const unsigned p = fdiv(rem, f.fW);
        const unsigned q = rem - p * (unsigned)f.Wp;
        const int pp = (int)p < a.P ? (int)p : a.P;
        return (int)n * a.PQ + pp * a.Q + ((int)p < a.P ? ((int)q < a.Q ? (int)q : a.Q) : 0);
    };
    const int mfirst = cnt(g0), mend = cnt(g0 + MS);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const unsigned g = g0 + (unsigned)(pw0 + mt * 16);
        const unsigned n = fdiv(g, f.fHW);
        const unsigned rem = g - n * (unsigned)f.HpWp;
        const unsigned p = fdiv(rem, f.fW);
        const unsigned q = rem - p * (unsigned)f.Wp;
        if (g < (unsigned)f.G && (int)p < a.P && (int)q < a.Q) {
            const int row = (int)n * a.PQ + (int)p * a.Q + (int)q - mfirst;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) Mma<T>::pack4(sC + row * LDC + wn * NT * 16 + nt * 16 + 4 * fq, acc[nt][mt]);
        }
    }
    __syncthreads();
    ConvArgs b = a;
    b.M = mend;                 // rows of the tile past the segment's last valid pixel are not written
    conv_epilogue_store<T, MS, KOUT, 512, MODE>(b, sC, sRed, t, lane, wave, mfirst, 0, seg);
}

// IFCBK_CONV_FLAT: 0 = never, 1 = where the grid fills the chip (default), 2 = wherever the kernel applies (tests)
int flat_mode() {
    const char* e = getenv("IFCBK_CONV_FLAT");
    return e ? atoi(e) : 1;
}

constexpr int FLAT_WPMAX = 40;

struct FlatShape { int cin, kout, r, s, ms; };
// instantiations: the 35x35 stage of inception_v3 (forward and input-gradient roles)
const FlatShape kFlat[] = {
    {48, 64, 5, 5, 512},   // Mixed_5x.branch5x5_2 forward
    {64, 48, 5, 5, 512},   //                      input gradient
    {64, 96, 3, 3, 448},   // branch3x3dbl_2 forward
    {96, 64, 3, 3, 448},   //                input gradient
    {96, 96, 3, 3, 448},   // branch3x3dbl_3 forward and input gradient
};

int flat_find(int cin, int kout, int r, int s) {
    for (int i = 0; i < (int)(sizeof(kFlat) / sizeof(kFlat[0])); ++i)
        if (kFlat[i].cin == cin && kFlat[i].kout == kout && kFlat[i].r == r && kFlat[i].s == s) return i;
    return -1;
}

template <int CIN, int KOUT, int R, int S, int WM, int WN, int MT, int NSTAGE>
void launch_flat(const ConvArgs& a, const FlatArgs& f, int nseg, hipStream_t st) {
    const dim3 grid((unsigned)nseg), block(512);
    if (a.bs_raw) hipLaunchKernelGGL((conv_flat<CIN, KOUT, R, S, WM, WN, MT, NSTAGE, FLAT_WPMAX, 3>), grid, block, 0, st, a, f);
    else hipLaunchKernelGGL((conv_flat<CIN, KOUT, R, S, WM, WN, MT, NSTAGE, FLAT_WPMAX, 0>), grid, block, 0, st, a, f);
}

}  // namespace

// Segments (= grid size = rows of the BatchNorm partial sums) if the flat kernel serves a stride-1 gather of `cin` channels
// over an [N,H,W] map into `kout` channels, filter R x S, gather padding (ph, pw) -- 0: it does not.
int ifcbk_conv_flat_segments(int dtype, int N, int H, int W, int cin, int kout, int R, int S, int ph, int pw, int P, int Q) {
    const int mode = flat_mode();
    if (mode <= 0 || dtype != IFCBK_BF16) return 0;
    const int i = flat_find(cin, kout, R, S);
    if (i < 0) return 0;
    if (ph < 0 || ph > R - 1 || pw < 0 || pw > S - 1) return 0;
    if (P != H + 2 * ph - R + 1 || Q != W + 2 * pw - S + 1 || P < 1 || Q < 1) return 0;
    const int Wp = W + pw, Hp = H + ph;
    if (Wp > FLAT_WPMAX) return 0;
    const int64_t G = (int64_t)N * Hp * Wp;
    if (G + kFlat[i].ms + 4096 >= (1ll << 31)) return 0;
    const int nseg = cdiv(G, kFlat[i].ms);
    if (mode < 2 && nseg < ifcbk_num_cus() / 2) return 0;      // small grids: the implicit GEMM's 128-pixel tiles fill more CUs
    return nseg;
}

int ifcbk_conv_flat_launch(ifcbk_ctx* ctx, void* args, int N, hipStream_t st) {
    ConvArgs& a = *reinterpret_cast<ConvArgs*>(args);
    FlatArgs f;
    f.ph = -a.base_h; f.pw = -a.base_w; f.N = N;
    f.Hp = a.H + f.ph; f.Wp = a.W + f.pw; f.HpWp = f.Hp * f.Wp;
    f.G = N * f.HpWp;
    f.fHW = make_fastdiv((uint32_t)f.HpWp); f.fW = make_fastdiv((uint32_t)f.Wp);
    const int i = flat_find(a.C, a.K, a.R, a.S);
    if (i < 0) IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv_flat: no instantiation for C=%d K=%d %dx%d", a.C, a.K, a.R, a.S);
    const int nseg = cdiv(f.G, kFlat[i].ms);
    a.tilesN = 1;
    switch (i) {
        case 0: launch_flat<48, 64, 5, 5, 8, 1, 4, 3>(a, f, nseg, st); break;
        case 1: launch_flat<64, 48, 5, 5, 8, 1, 4, 3>(a, f, nseg, st); break;
        case 2: launch_flat<64, 96, 3, 3, 4, 2, 7, 3>(a, f, nseg, st); break;
        case 3: launch_flat<96, 64, 3, 3, 4, 2, 7, 3>(a, f, nseg, st); break;
        default: launch_flat<96, 96, 3, 3, 4, 2, 7, 3>(a, f, nseg, st); break;
    }
    IFCBK_LAUNCH_CHECK(ctx, "conv_flat");
    return 0;
}
