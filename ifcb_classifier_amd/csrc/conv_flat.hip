// Flat-image convolution for the stride-1 layers of the 35x35 stage (Mixed_5b/c/d, Mixed_6a's inputs: 3x3 and 5x5 filters over
// 48..96 channels), forward and input gradient (the input gradient of a stride-1 conv is a stride-1 conv of dy with the flipped,
// transposed filter), bf16 storage, fp32 accumulate.
//
// Why not the implicit GEMM: conv_igemm gathers every input pixel once per TAP (9x / 25x) through the CU's global->LDS path; with
// 64..96 output channels per tile that path, not the matrix pipe, sets the speed (56 flop per gathered byte; 14-19 % MFMA busy).
// Here the input is laid out FLAT: image n, input row h, column w live at slot
//     F = (n*Hp + h + ph)*Wp + (w + pw),   Hp = H + ph, Wp = W + pw
// (one shared band of zero slots between rows / images serves as right+left and bottom+top padding), an output pixel (p, q) is
// slot g = (n*Hp + p)*Wp + q, and tap (r, s) of ANY output slot reads slot g + r*Wp + s: every tap is a constant shift.  A
// SEGMENT of MS consecutive output slots needs the MS + (R-1)*Wp + S-1 input slots behind it: they come to LDS ONCE (LDS-DMA,
// zeros outside the image through the buffer range check) and all R*S*C/32 pixel fragments of the reduction are read from that
// resident image at shifted addresses; only the filter (L2-resident, k-contiguous rows) streams through a two-slot ring.
// Global->LDS traffic per MAC: 1/4 of the implicit GEMM's.  Outputs at padding slots (q >= Q or p >= P: 5-11 % of the slots)
// are computed and dropped.
//
// Structure (from s_memtime stamps and timing-only builds of two earlier forms): a block that loads its image,
// multiplies and writes its tile out in turn spends as long in the load and store phases as in the MFMAs (a CU streams
// 5-10 B/clk to memory, whoever else is busy: 37 KB of outputs = 8,500 cycles against 8,000 cycles of MFMA), a second resident
// block hides little of that, and an LDS-DMA instruction costs its wave ~200 issue cycles.  So: ONE PERSISTENT 512-thread block
// per CU walks over its segments with its waves SPECIALISED --
//   waves 4-7, loaders: every LDS-DMA of the block (filter stage s+1 and a share of the NEXT segment's image while stage s is
//     multiplied; counted vmcnt, one s_barrier per stage is the whole protocol); they never store, so no wait drains a store;
//   waves 0-3, consumers (one per SIMD): fragments + MFMAs only; at a segment's end the accumulators become packed bf16 rows in
//     REGISTERS (BatchNorm statistics of the rounded values / the BN-backward sums of MODE 3 / the eval affine are taken there)
//     and are stored a few per stage WHILE the next segment is multiplied: the store stream never exceeds what the CU drains.
// Consumer wave (wm, wn) owns MT pixel tiles x NT = K/16/WN channel tiles of 16x16 (filter = MFMA A operand, pixels = B operand:
// a lane ends with 4 consecutive channels of one pixel = one 8-byte store).  LDS read traffic per MFMA cycle is
// 4096*(WN/K + WM/MS) B/clk, which picks (WM, WN) = (2, 2) for 96 output channels and (4, 1) below.
// BatchNorm partial sums: one row per BLOCK (accumulated in registers over the block's segments, fixed order).
#include "conv_common.h"
#include <stdlib.h>

namespace {

struct FlatArgs {
    int Hp, Wp, HpWp, ph, pw, N, G, nseg;   // G = N*Hp*Wp output slots
    unsigned ybytes, rawbytes;              // buffer-descriptor extents of the output and of MODE 3's raw tensor
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

#ifdef IFCBK_EXPERIMENT_FLAT
#define FLAT_DBG(bit) ((a.dbg & (bit)) != 0)      // timing-only builds: 1 no image DMA, 2 no filter DMA, 4 no multiply, 8 no epilogue
#else
#define FLAT_DBG(bit) false
#endif

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

// KPS: 32-k MFMA steps per ring stage; NSTAGE: ring slots (the loaders run NSTAGE-1 stages ahead).  Shipped: 128-k stages, two slots;
// 64-k stages with four slots measured 10 % slower -- the loaders are bound by the number of LDS-DMA instructions (~200 cycles each
// for the issuing wave, at any depth), not by their latency
template <int CIN, int KOUT, int R, int S, int WM, int WN, int MT, int KPS, int NSTAGE, int WPMAX, int MODE>
__global__ __launch_bounds__(512) void conv_flat(ConvArgs a, FlatArgs f) {
    typedef bf16_t T;
    typedef bf16x8_t frag_t;
    constexpr int CPP = CIN / 8;                      // 16-byte chunks per pixel
    static_assert(WM * WN == 4 && KOUT % (16 * WN) == 0 && CIN % 8 == 0 && KPS % 2 == 0, "tile");
    constexpr int NT = KOUT / 16 / WN;
    constexpr int MS = 16 * MT * WM;                  // output slots per segment
    constexpr int JR = S * CPP;                       // chunks per filter row: contiguous in the flat image AND in the filter
    constexpr int NCH = R * JR;                       // chunks of the reduction
    constexpr int NKS = (NCH + 3) / 4;                // MFMA steps (32 k)
    constexpr int NK = (NKS + KPS - 1) / KPS;         // ring stages per segment
    constexpr int NSUB = KPS / 2;                     // 64-k sub-tiles per stage, each [KOUT rows][8 chunks]
    constexpr int HALO = (R - 1) * WPMAX + S - 1;
    constexpr int NPIECE = ((MS + HALO) * CPP + 63) / 64;
    constexpr int IMG_BYTES = NPIECE * 1024;
    constexpr int NLW = 4;                            // loader waves
    constexpr int D = NSTAGE - 1;                     // stages the filter ring runs ahead
    constexpr int IPS = (NPIECE + NLW * (NK - D) - 1) / (NLW * (NK - D));   // pieces of the next image per loader wave and stage
    constexpr int KI = (NPIECE + NLW * IPS - 1) / (NLW * IPS);     // ... in the first KI stages of a segment
    constexpr int SUB_BYTES = KOUT * 128;
    constexpr int BSTAGE = NSUB * SUB_BYTES;
    constexpr int NBP = NSUB * KOUT / 8;              // LDS-DMA pieces per stage (8 rows x 128 B each)
    constexpr int JB = (NBP + NLW - 1) / NLW;         // per loader wave
    constexpr int PAR_BYTES = 4 * KOUT * 4;           // per-channel parameters of the epilogue (affine / BN-backward), fp32
    constexpr int RED_BYTES = 4 * 2 * KOUT * 4;       // statistics hand-over between the consumer waves at the very end
    constexpr int SMEM_BYTES = 2 * IMG_BYTES + NSTAGE * BSTAGE + PAR_BYTES + RED_BYTES + 1024;
    static_assert(SMEM_BYTES <= 160 * 1024, "LDS");
    static_assert(JR >= 4 && D >= 1 && D <= 3 && NK > D && KI <= NK - D, "a segment's last D stages carry no image piece: stage 0 of the next waits for all of them");
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
    unsigned char* sRingB = smem + 2 * IMG_BYTES;
    float* sPar = reinterpret_cast<float*>(smem + 2 * IMG_BYTES + NSTAGE * BSTAGE);
    float* sRed = reinterpret_cast<float*>(smem + 2 * IMG_BYTES + NSTAGE * BSTAGE + PAR_BYTES);
    unsigned char* sDummy = smem + 2 * IMG_BYTES + NSTAGE * BSTAGE + PAR_BYTES + RED_BYTES;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int grid = (int)gridDim.x;
    const int bb = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int nmine = (f.nseg - bb + grid - 1) / grid;    // segments bb, bb + grid, ...: every XCD works on one contiguous range
    constexpr unsigned OOB = 0x80000000u;

    // per-channel epilogue parameters (tiny, read by the consumers at a segment's end)
    if (MODE == 3) {
        for (int c = t; c < KOUT; c += 512) {
            sPar[c] = a.bs_mean[c]; sPar[KOUT + c] = a.bs_invstd[c]; sPar[2 * KOUT + c] = a.bs_scale[c]; sPar[3 * KOUT + c] = a.bs_shift[c];
        }
    } else if (a.ep_scale) {
        for (int c = t; c < KOUT; c += 512) { sPar[c] = a.ep_scale[c]; sPar[KOUT + c] = a.ep_shift[c]; }
    }
    __syncthreads();

    if (wave >= 4) {
        // ================================================================ loaders
        const int lw = wave - 4;
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.wbytes, 0x00020000);
        const int np = ((MS + (R - 1) * f.Wp + S - 1) * CPP + 63) >> 6;      // pieces of a segment image (<= NPIECE: Wp <= WPMAX)
        // one LDS-DMA piece (64 chunks in image order) of the image of the segment that starts at slot g0
        // This wave's pieces of an image are pc = lw, lw + 4, ...: the lane's position (pixel lpx of the image, chunk phys, and
        // that pixel's image / row / column in the flat layout) is decoded ONCE per segment (two divisions) and then stepped
        // from piece to piece with adds and carries -- a decode per piece cost the loaders more VALU time than the DMAs themselves
        constexpr int DL = (NLW * 64) / CPP, PH = (NLW * 64) % CPP;
        int i_lpx, i_phys, i_n, i_hr, i_wc;
        auto piece_begin = [&](unsigned g0) {
            const int idx = lw * 64 + lane;
            i_lpx = idx / CPP;
            i_phys = idx - i_lpx * CPP;
            const unsigned F = g0 + (unsigned)i_lpx;
            const unsigned n = fdiv(F, f.fHW);
            const unsigned rem = F - n * (unsigned)f.HpWp;
            const unsigned hr = fdiv(rem, f.fW);
            i_n = (int)n; i_hr = (int)hr; i_wc = (int)(rem - hr * (unsigned)f.Wp);
        };
        auto piece_issue = [&](unsigned char* dst, bool live) {       // the current piece, then step to this wave's next one
            const int w = i_wc - f.pw, h = i_hr - f.ph;
            const bool v = live && i_n < f.N && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W && !FLAT_DBG(1);
            const unsigned voff = v ? (unsigned)(((i_n * a.H + h) * a.W + w) * a.ldx + flat_swz<CPP>(i_lpx, i_phys) * 8) * 2u : OOB;
            lds_dma16(rsA, (lptr_t)(live ? dst : sDummy), voff, 0);
            int dl = DL;
            i_phys += PH;
            if (i_phys >= CPP) { i_phys -= CPP; ++dl; }
            i_lpx += dl;
            i_wc += dl;
            while (i_wc >= f.Wp) { i_wc -= f.Wp; ++i_hr; }
            while (i_hr >= f.Hp) { i_hr -= f.Hp; ++i_n; }
        };
        // filter: a stage is NSUB sub-tiles [KOUT rows][8 chunks], chunk c of row r at physical chunk c ^ (r & 7) (source side)
        const int lrow8 = lane >> 3;
        const int csrc = (lane & 7) ^ lrow8;
        unsigned woff[JB];                                  // this lane's chunk of stage 0 (bytes); a stage advances KPS*64 bytes: scalar offset
        bool glive[JB], gtail[JB];                          // piece exists / this lane's chunk of a segment's LAST stage lies inside the filter row
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            const int grp = j * NLW + lw;                   // piece of the stage: sub-tile grp / (KOUT/8), row group grp % (KOUT/8)
            glive[j] = grp < NBP;
            const int psub = grp / (KOUT / 8);
            const int prow = (grp - psub * (KOUT / 8)) * 8 + lrow8;
            woff[j] = glive[j] ? (unsigned)(prow * a.Kg + (psub * 8 + csrc) * 8) * 2u : OOB;
            gtail[j] = ((NK - 1) * NSUB + psub) * 8 + csrc < NCH;
        }
        // stage kt of a segment: chunks kt*KPS*4 .. of every filter row; chunks past the row's end (last stage) read zeros
#define FLAT_ISSUE_B(kt, slot)                                                                              \
        _Pragma("unroll") for (int j = 0; j < JB; ++j) {                                                    \
            const bool cut = ((kt) == NK - 1 && !gtail[j]) || FLAT_DBG(2);                                  \
            unsigned char* dst = glive[j] ? sRingB + (slot) * BSTAGE + (j * NLW + lw) * 1024 : sDummy;      \
            lds_dma16(rsB, (lptr_t)dst, cut ? OOB : woff[j], (kt) * (KPS * 64));                            \
        }
        piece_begin((unsigned)bb * MS);
        for (int pc = lw; pc < np; pc += NLW) piece_issue(smem + pc * 1024, true);
#pragma unroll
        for (int d = 0; d < D; ++d) FLAT_ISSUE_B(d, d)
        wait_vmcnt<0>();
        int islot = D;                                      // ring slot the next fetched stage goes to
        bool first = true;
        for (int it = 0; it < nmine; ++it) {
            const bool has_next = it + 1 < nmine;
            unsigned char* bufn = smem + ((it + 1) & 1) * IMG_BYTES;
            piece_begin((unsigned)(bb + (it + 1) * grid) * MS);
            for (int kt = 0; kt < NK; ++kt) {
                // Stage kt has landed once only the loads issued BEHIND its pieces are outstanding: the D-1 later filter stages and
                // the image pieces of the last D stage bodies (a body issues filter stage +D, then its image pieces)
                if (!first) {
                    int c = 0;
#pragma unroll
                    for (int d = 1; d <= D; ++d) {
                        const int kp = kt - d < 0 ? kt - d + NK : kt - d;
                        c += kp < KI ? 1 : 0;
                    }
                    if (c == 0) wait_vmcnt<(D - 1) * JB>();
                    else if (c == 1) wait_vmcnt<(D - 1) * JB + IPS>();
                    else if (c == 2) wait_vmcnt<(D - 1) * JB + 2 * IPS>();
                    else wait_vmcnt<(D - 1) * JB + 3 * IPS>();
                }
                first = false;
                __builtin_amdgcn_s_barrier();       // stage kt is in LDS; the consumers have finished stage kt-1 (its slot is free)
                {
                    const int ktn = kt + D >= NK ? kt + D - NK : kt + D;      // running on into the next segment (the last one refetches early stages: unused)
                    FLAT_ISSUE_B(ktn, islot)
                    islot = islot + 1 == NSTAGE ? 0 : islot + 1;
                }
                if (kt < KI) {
#pragma unroll
                    for (int u = 0; u < IPS; ++u) {
                        const int pc = (kt * IPS + u) * NLW + lw;
                        piece_issue(bufn + pc * 1024, has_next && pc < np);
                    }
                }
            }
        }
#undef FLAT_ISSUE_B
        wait_vmcnt<0>();
        if (a.part) __syncthreads();                // the consumers' statistics hand-over
        return;
    }

    // ==================================================================== consumers
    const int wm = wave / WN, wn = wave - wm * WN;
    const int frow = lane & 15, fq = lane >> 4;
    const int pw0 = wm * MT * 16 + frow;              // this lane's pixel of tile mt = 0 inside the segment
    const int ch0 = wn * NT * 16 + 4 * fq;            // this lane's first channel of tile nt = 0
    // filter fragments: row wn*NT*16 + nt*16 + frow of a sub-tile, chunk (kk&1)*4 + fq, swizzled by the row
    int fBoff[2];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) fBoff[k2] = ((wn * NT * 16 + frow) * 64 + (((k2 * 4 + fq) ^ (frow & 7)) * 8)) * 2;
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, f.ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)(MODE == 3 ? a.bs_raw : a.y), 0, MODE == 3 ? f.rawbytes : 0u, 0x00020000);

    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    f32x4_t acc[NT][MT];
    uint2 outr[NT][MT];           // the previous segment's tile, packed bf16: stored while this segment is multiplied
    unsigned orow[MT];            // byte offset of its rows in y (this lane's pixel and first channel), OOB = padding slot
    f32x2_t s1[NT][2], s2[NT][2]; // statistics of this lane's 4 channels per channel tile, over all of the block's segments
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        s1[i][0] = s1[i][1] = s2[i][0] = s2[i][1] = f32x2_t{0.f, 0.f};
#pragma unroll
        for (int j = 0; j < MT; ++j) { acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; outr[i][j] = uint2{0u, 0u}; }
    }
#pragma unroll
    for (int j = 0; j < MT; ++j) orow[j] = OOB;
    constexpr int NTILE = NT * MT;
    constexpr int SPS = (NTILE + NK - 1) / NK;        // deferred stores per stage

    int slot = 0;
    for (int it = 0; it < nmine; ++it) {
        const int seg = bb + it * grid;
        const unsigned g0 = (unsigned)seg * MS;
        const T* sImg = reinterpret_cast<const T*>(smem + (it & 1) * IMG_BYTES);
        // rows of this segment's tile (output pixel index m), needed at its end
        unsigned crow[MT], rrow[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const unsigned g = g0 + (unsigned)(pw0 + mt * 16);
            const unsigned n = fdiv(g, f.fHW);
            const unsigned rem = g - n * (unsigned)f.HpWp;
            const unsigned p = fdiv(rem, f.fW);
            const unsigned q = rem - p * (unsigned)f.Wp;
            const bool v = g < (unsigned)f.G && (int)p < a.P && (int)q < a.Q;
            const unsigned m = n * (unsigned)a.PQ + p * (unsigned)a.Q + q;
            crow[mt] = v ? (m * (unsigned)a.ldy + (unsigned)ch0) * 2u : OOB;
            rrow[mt] = (MODE == 3 && v) ? (m * (unsigned)a.bs_ld + (unsigned)ch0) * 2u : OOB;
        }
        u32x2_t rawv[MODE == 3 ? NT : 1][MODE == 3 ? MT : 1];
        // pixel fragments: chunk ci = 4*step + fq of the reduction = chunk jj of filter row rr, i.e. slot offset rr*Wp + jj / CPP
        int jj = fq, rr = 0;
        for (int kt = 0; kt < NK; ++kt) {
            __builtin_amdgcn_s_barrier();       // the loaders' stage kt is in LDS
            if (MODE == 3 && kt == NK - 1) {
                // the producing BatchNorm's input at this tile's pixels: requested one stage ahead of its use
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        rawv[MODE == 3 ? nt : 0][MODE == 3 ? mt : 0] = __builtin_amdgcn_raw_buffer_load_b64(rsR, rrow[mt], nt * 32, 0);
            }
            const unsigned char* stg = sRingB + slot * BSTAGE;
            if (!FLAT_DBG(4)) {
#pragma unroll
                for (int kk = 0; kk < KPS; ++kk) {
                    if (kt * KPS + kk < NKS) {
                        const int dpx = jj / CPP;
                        const int ch = jj - dpx * CPP;
                        const bool dead = NCH % 4 != 0 && rr >= R;         // chunk past the reduction's end (last step only)
                        const int px = dead ? 0 : pw0 + rr * f.Wp + dpx;
                        const T* pa = sImg + (px * CPP + flat_swz<CPP>(px, ch)) * 8;
                        frag_t fa[MT], fb[NT];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) fa[mt] = *reinterpret_cast<const frag_t*>(pa + mt * 16 * CPP * 8);
                        const unsigned char* pb = stg + (kk >> 1) * SUB_BYTES + fBoff[kk & 1];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) fb[nt] = *reinterpret_cast<const frag_t*>(pb + nt * 16 * 128);
                        if (NCH % 4 != 0) {
                            if (dead) {
#pragma unroll
                                for (int mt = 0; mt < MT; ++mt) fa[mt] = frag_t{};
                            }
                        }
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
                                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[nt][mt], 0, 0, 0);
                        jj += 4;
                        if (jj >= JR) { jj -= JR; ++rr; }
                    }
                }
            }
            // the previous segment's rows leave a few per stage (out-of-range offsets = padding slots are dropped by the descriptor)
            if (!FLAT_DBG(8)) {
#pragma unroll
                for (int c = 0; c < NK; ++c)
                    if (kt == c) {
#pragma unroll
                        for (int u = 0; u < SPS; ++u) {
                            const int idx = c * SPS + u;
                            if (idx < NTILE) {
                                const int nt = idx / MT, mt = idx - nt * MT;
                                __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{outr[nt][mt].x, outr[nt][mt].y}, rsY, orow[mt], nt * 32, 0);
                            }
                        }
                    }
            }
            slot = slot + 1 == NSTAGE ? 0 : slot + 1;
        }
        // ---- this segment's tile: rounded to the storage type in registers, statistics of the rounded values
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float p0[4], p1[4], p2[4], p3[4];
            if (MODE == 3) {
                const f32x4_t mu = *reinterpret_cast<const f32x4_t*>(sPar + ch0 + nt * 16), is = *reinterpret_cast<const f32x4_t*>(sPar + KOUT + ch0 + nt * 16);
                const f32x4_t sc = *reinterpret_cast<const f32x4_t*>(sPar + 2 * KOUT + ch0 + nt * 16), sh = *reinterpret_cast<const f32x4_t*>(sPar + 3 * KOUT + ch0 + nt * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) { p0[j] = mu[j]; p1[j] = is[j]; p2[j] = sc[j]; p3[j] = sh[j]; }
            } else if (a.ep_scale) {
                const f32x4_t sc = *reinterpret_cast<const f32x4_t*>(sPar + ch0 + nt * 16), sh = *reinterpret_cast<const f32x4_t*>(sPar + KOUT + ch0 + nt * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) { p0[j] = sc[j]; p1[j] = sh[j]; }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4_t v = acc[nt][mt];
                acc[nt][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                if (MODE != 3 && a.ep_scale) {
                    // the affine acts on the conv output AS STORED in training (rounded to bf16), like conv_common.h's epilogue
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = bf2f(f2bf(v[j])) * p0[j] + p1[j];
                        if (a.ep_relu) v[j] = fmaxf(v[j], 0.f);
                    }
                }
                uint2 u;
                u.x = pack2bf(v[0], v[1]);
                u.y = pack2bf(v[2], v[3]);
                outr[nt][mt] = u;
                const bool valid = crow[mt] != OOB;
                f32x2_t lo = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u)};
                f32x2_t hi = {__uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
                if (!valid) lo = hi = f32x2_t{0.f, 0.f};
                if (MODE == 3) {
                    // BN-backward sums of the producing layer: dz = dx where its activation was positive, xhat from its input
                    const u32x2_t rw = rawv[MODE == 3 ? nt : 0][MODE == 3 ? mt : 0];
                    const float x0 = __uint_as_float(rw[0] << 16), x1 = __uint_as_float(rw[0] & 0xffff0000u);
                    const float x2 = __uint_as_float(rw[1] << 16), x3 = __uint_as_float(rw[1] & 0xffff0000u);
                    const f32x2_t dzl = {(x0 * p2[0] + p3[0]) > 0.f ? lo[0] : 0.f, (x1 * p2[1] + p3[1]) > 0.f ? lo[1] : 0.f};
                    const f32x2_t dzh = {(x2 * p2[2] + p3[2]) > 0.f ? hi[0] : 0.f, (x3 * p2[3] + p3[3]) > 0.f ? hi[1] : 0.f};
                    const f32x2_t xl = {(x0 - p0[0]) * p1[0], (x1 - p0[1]) * p1[1]}, xh = {(x2 - p0[2]) * p1[2], (x3 - p0[3]) * p1[3]};
                    s1[nt][0] += dzl; s1[nt][1] += dzh;
                    s2[nt][0] += dzl * xl; s2[nt][1] += dzh * xh;
                } else if (a.part) {
                    s1[nt][0] += lo; s1[nt][1] += hi;
                    s2[nt][0] += lo * lo; s2[nt][1] += hi * hi;
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) orow[mt] = crow[mt];
    }
    // ---- the last segment's rows
    if (!FLAT_DBG(8)) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{outr[nt][mt].x, outr[nt][mt].y}, rsY, orow[mt], nt * 32, 0);
    }
    // ---- statistics: over the 16 pixel lanes, then over the pixel groups (fixed order) -> one partial row per block
    if (a.part) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float x1 = s1[nt][h][j], x2 = s2[nt][h][j];
#pragma unroll
                    for (int off = 1; off < 16; off <<= 1) {
                        x1 += __shfl_xor(x1, off);
                        x2 += __shfl_xor(x2, off);
                    }
                    if (frow == 0) {
                        sRed[(wm * 2 + 0) * KOUT + ch0 + nt * 16 + h * 2 + j] = x1;
                        sRed[(wm * 2 + 1) * KOUT + ch0 + nt * 16 + h * 2 + j] = x2;
                    }
                }
        __syncthreads();
        for (int i = t; i < 2 * KOUT; i += 256) {
            const int which = i / KOUT, c = i - which * KOUT;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) s += sRed[(w * 2 + which) * KOUT + c];
            a.part[((size_t)bb * 2 + which) * KOUT + c] = s;
        }
    }
}

// IFCBK_CONV_FLAT: 0 = never, 1 = where the grid fills the chip (default), 2 = wherever the kernel applies (tests)
int flat_mode() {
    const char* e = getenv("IFCBK_CONV_FLAT");
    return e ? atoi(e) : 1;
}

constexpr int FLAT_WPMAX = 40;

struct FlatShape { int cin, kout, r, s, ms; };
// instantiations: the 35x35 stage of inception_v3 (forward and input-gradient roles); ms = 16 * MT * WM of launch_flat below
const FlatShape kFlat[] = {
    {64, 48, 5, 5, 256},   // Mixed_5x.branch5x5_2 input gradient (its forward, 48 -> 64, measured no faster than conv_igemm: not built)
    {64, 96, 3, 3, 192},   // branch3x3dbl_2 forward
    {96, 64, 3, 3, 192},   //                input gradient
    {96, 96, 3, 3, 192},   // branch3x3dbl_3 forward and input gradient
};

int flat_find(int cin, int kout, int r, int s) {
    for (int i = 0; i < (int)(sizeof(kFlat) / sizeof(kFlat[0])); ++i)
        if (kFlat[i].cin == cin && kFlat[i].kout == kout && kFlat[i].r == r && kFlat[i].s == s) return i;
    return -1;
}

int flat_grid(int nseg) {
    const int cus = ifcbk_num_cus();
    return nseg < cus ? nseg : cus;
}

template <int CIN, int KOUT, int R, int S, int WM, int WN, int MT, int KPS, int NSTAGE>
int launch_flat(const ConvArgs& a, const FlatArgs& f, int ms, hipStream_t st) {
    if (ms != 16 * MT * WM) return -1;
    const dim3 grid((unsigned)flat_grid(f.nseg)), block(512);
    if (a.bs_raw) hipLaunchKernelGGL((conv_flat<CIN, KOUT, R, S, WM, WN, MT, KPS, NSTAGE, FLAT_WPMAX, 3>), grid, block, 0, st, a, f);
    else hipLaunchKernelGGL((conv_flat<CIN, KOUT, R, S, WM, WN, MT, KPS, NSTAGE, FLAT_WPMAX, 0>), grid, block, 0, st, a, f);
    return 0;
}

}  // namespace

// Rows of the BatchNorm partial sums (= blocks of the persistent grid) if the flat kernel serves a stride-1 gather of `cin`
// channels over an [N,H,W] map into `kout` channels, filter R x S, gather padding (ph, pw) -- 0: it does not.
int ifcbk_conv_flat_rows(int dtype, int N, int H, int W, int cin, int kout, int R, int S, int ph, int pw, int P, int Q) {
    const int mode = flat_mode();
    if (mode <= 0 || dtype != IFCBK_BF16) return 0;
    const int i = flat_find(cin, kout, R, S);
    if (i < 0) return 0;
    if (ph < 0 || ph > R - 1 || pw < 0 || pw > S - 1) return 0;
    if (P != H + 2 * ph - R + 1 || Q != W + 2 * pw - S + 1 || P < 1 || Q < 1) return 0;
    const int Wp = W + pw, Hp = H + ph;
    if (Wp > FLAT_WPMAX) return 0;
    const int64_t G = (int64_t)N * Hp * Wp;
    if (G + (int64_t)(ifcbk_num_cus() + 1) * kFlat[i].ms + 4096 >= (1ll << 31)) return 0;
    const int nseg = cdiv(G, kFlat[i].ms);
    if (mode < 2 && nseg < 2 * ifcbk_num_cus()) return 0;      // small grids: the implicit GEMM's 128-pixel tiles fill more CUs
    return flat_grid(nseg);
}

// (the caller has checked ifcbk_conv_flat_rows; accumulate / residual / segmented / table forms stay with the implicit GEMM)
int ifcbk_conv_flat_launch(ifcbk_ctx* ctx, void* args, int N, hipStream_t st) {
    ConvArgs& a = *reinterpret_cast<ConvArgs*>(args);
    FlatArgs f;
    f.ph = -a.base_h; f.pw = -a.base_w; f.N = N;
    f.Hp = a.H + f.ph; f.Wp = a.W + f.pw; f.HpWp = f.Hp * f.Wp;
    f.G = N * f.HpWp;
    f.fHW = make_fastdiv((uint32_t)f.HpWp); f.fW = make_fastdiv((uint32_t)f.Wp);
    f.ybytes = (unsigned)((int64_t)a.M * a.ldy * 2);
    f.rawbytes = a.bs_raw ? (unsigned)((int64_t)a.M * a.bs_ld * 2) : 0u;
    const int i = flat_find(a.C, a.K, a.R, a.S);
    if (i < 0) IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv_flat: no instantiation for C=%d K=%d %dx%d", a.C, a.K, a.R, a.S);
    if ((int64_t)a.M * a.ldy * 2 >= (1ll << 31) || (a.bs_raw && (int64_t)a.M * a.bs_ld * 2 >= (1ll << 31)))
        IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv_flat: a tensor exceeds the 2 GiB buffer-descriptor window");
    f.nseg = cdiv(f.G, kFlat[i].ms);
    a.tilesN = 1;
#ifdef IFCBK_EXPERIMENT_FLAT
    if (const char* e = getenv("IFCBK_DEBUG_DROP")) a.dbg = atoi(e);       // timing-only builds (wrong results): see FLAT_DBG
#endif
    int rc;
    switch (i) {
        case 0: rc = launch_flat<64, 48, 5, 5, 4, 1, 4, 4, 2>(a, f, kFlat[i].ms, st); break;
        case 1: rc = launch_flat<64, 96, 3, 3, 2, 2, 6, 4, 2>(a, f, kFlat[i].ms, st); break;
        case 2: rc = launch_flat<96, 64, 3, 3, 4, 1, 3, 4, 2>(a, f, kFlat[i].ms, st); break;
        default: rc = launch_flat<96, 96, 3, 3, 2, 2, 6, 4, 2>(a, f, kFlat[i].ms, st); break;
    }
    if (rc) IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv_flat: table / template mismatch");
    IFCBK_LAUNCH_CHECK(ctx, "conv_flat");
    return 0;
}
