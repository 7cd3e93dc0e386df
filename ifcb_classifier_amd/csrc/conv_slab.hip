// Pixel-slab convolution (forward / stride-1 input gradient), bf16, round 5: conv_pp2's wide-tile ping-pong block (conv_big.hip)
// for the multi-tap layers, with the PIXEL operand taken out of the per-K-tile LDS-DMA stream.
//
// Why.  As an implicit GEMM a K-tile (64 k = one tap x 64 channels) brings 320 pixel rows + 192 filter rows of 128 B into LDS: 64
// LDS-DMA pieces of 1 KiB per 1,920 cycles of MFMA -- and the CU's global->LDS path takes ~50 cycles per piece whatever it carries
// (a K-tile costs 1.3-1.4 us where its MFMAs need 0.8).  The seven taps of a 1x7 / 7x1 filter read the SAME
// pixels, shifted by one position each.  So, per 64-channel chunk, the tile's pixels land in LDS ONCE, as a SLAB of consecutive
// flat slots, and every tap reads its fragments from that slab at a shifted row; only the filter still streams per K-tile:
// (49 slab + 7 x 24 filter) = 217 pieces per 7 K-tiles instead of 448 -- 31 instead of 64 per K-tile.
//
// Flat slots (as conv_flat.hip, generalised to either axis as the MINOR one).  With (a, b) = (h, w) for filters that extend along
// w ("row-major": 1x7, 3x3 ...) or (w, h) for filters that extend along h only ("column-major": 7x1 -- the taps of a vertical
// filter are then adjacent slots too), input (n, a, b) lives at slot
//     (n * MajP + a + pa) * MinP + (b + pb),     MajP = A + pa, MinP = B + pb     (pa, pb: the gather's padding)
// -- one shared band of pb (pa * MinP) invalid slots between lines (images) serves as the padding of both neighbours: a slot
// whose decoded (a, b) falls into a band, or whose image is >= N, is requested through an out-of-range offset and reads zeros.
// Output pixel (n, oa, ob) sits at slot g = (n * MajP + oa) * MinP + ob and tap (ta, tb) of ANY output reads slot g + ta * MinP + tb:
// a constant shift.  A tile is 32 * MT consecutive output pixels in (n, oa, ob) order -- VALID pixels only, so a lane's 16 fragment
// rows are not consecutive slots across a line end: every lane carries the slab row of its pixel of each of its MT pixel tiles
// (rowbase[], MT registers) and adds the tap's shift; the XOR swizzle of the 128-byte rows depends on the shifted row, which costs
// ~5 VALU per fragment pair -- against the ~8 address VALU per LDS-DMA piece that the gather no longer issues.
// Column-major tiles walk the image column by column: the epilogue maps a tile row back to its NHWC row (ConvArgs::tr).
//
// Reduction order: (chunk, tap, channel-in-chunk) -- conv_pp2 / conv_igemm run (tap, channel).  Same products, another fp32
// summation order: results agree to fp32 rounding (~1e-7 of the output), not bit for bit; the dispatch therefore depends on the
// layer's shape only (never on which batch size makes another kernel faster), so that a RUN batch equals its parts bit for bit.
//
// Schedule: conv_pp2's, literally (two groups of four waves half a phase apart, two phases per K-tile, raw s_barrier, counted
// vmcnt, inline-asm fragment reads) with these differences:
//   * K-tile kt = chunk cc * TAPS + tap t; its filter tile is 64 k at element offset t * C + cc * 64 of every filter row;
//   * no pixel pieces per K-tile; the slab of chunk cc + 1 (buffer (cc + 1) & 1) is requested during chunk cc, one piece per wave
//     and load part, from the ODD phase of tap 0 on (its buffer was last read in the odd phase of the previous chunk's last tap:
//     conv_pp2's WAR rule -- a slot is re-filled two load parts after its last read) and early in the chunk, so that the waits that
//     cover the filter tiles (in-order vmcnt) find them long landed; the odd phase of a chunk's last tap waits for everything.
#include "conv_common.h"
#include <stdlib.h>

namespace {

#define SLAB_DSREAD(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))
#define SLAB_DSREAD_OFF(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))

struct SlabArgs {
    int MajP, MinP, ImgP;        // lines per image / slots per line / slots per image of the padded flat layout
    int pa, pb;                  // padding of the gather along the major / minor axis
    int IA, IB;                  // input extent along the major / minor axis
    int OA, OB;                  // output extent along the major / minor axis
    int sa, sb, simg;            // input element stride of one step along the major / minor axis (W * ldx and ldx, or swapped), of one image
    int TB;                      // filter taps along the minor axis (filter order is (r, s): tap t = r * S + s)
    int tr;                      // 1: column-major
    int N;
    fastdiv_t fImg, fLine, fOPQ, fOB, fTB;
};

// SLABP: LDS-DMA pieces (8 rows of 128 B) of one slab; wave w requests pieces w, w + 8, ...
template <int TN, int MT, int PM0, int TAPS, int SLABP, int MODE>
__global__ __launch_bounds__(512) void conv_slab(ConvArgs a, SlabArgs s) {
    typedef bf16_t T;
    constexpr int ES = 2, CE = 8, BK = 64;
    constexpr int PM1 = MT - PM0;
    static_assert(PM0 > 0 && PM1 > 0 && TAPS >= 3, "phases");
    constexpr int PMX = PM0 > PM1 ? PM0 : PM1;
    constexpr int HM = 16 * MT, BM = 2 * HM, BN = 64 * TN;
    constexpr int ROWB = BK * ES;                          // 128 bytes per slab / filter row
    constexpr int SPW = (SLABP + 7) / 8;                   // pieces per wave (the last ones of some waves do not exist)
    constexpr int SLAB_BYTES = SLABP * 1024;
    constexpr int BBUF = BN * ROWB;
    constexpr int RING_BYTES = 2 * SLAB_BYTES + 2 * BBUF;
    constexpr int LDC = BN + CE;
    constexpr int CT_BYTES = BM * LDC * ES + 8 * BN * 2 * 4;
    constexpr int SMEM_BYTES = RING_BYTES > CT_BYTES ? RING_BYTES : CT_BYTES;
    static_assert(SMEM_BYTES <= 160 * 1024, "LDS");
    static_assert(SPW <= 2 * (TAPS - 1), "one slab piece per wave and load part, from tap 0's odd phase to the last tap's even phase");
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
    T* sC = reinterpret_cast<T*>(smem);
    float* sRed = reinterpret_cast<float*>(smem + BM * LDC * ES);
    unsigned char* sFilt = smem + 2 * SLAB_BYTES;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int grp = wave >> 2, wc = wave & 3;
    const int bid = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int mtile = bid / a.tilesN, ntile = bid - mtile * a.tilesN;
    const int m0 = mtile * BM, n0 = ntile * BN;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.wbytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int lrow8 = lane >> 3;
    const int csrc = (lane & 7) ^ lrow8;                   // logical 16-byte chunk this lane fetches (source-side swizzle)
    const int frow = lane & 15, fchunk = lane >> 4;

    // output pixel (tile order index m) -> its flat slot
    auto slot_of = [&](int m) -> int {
        const unsigned n = fdiv((unsigned)m, s.fOPQ);
        const unsigned rem = (unsigned)m - n * s.fOPQ.d;
        const unsigned oa = fdiv(rem, s.fOB);
        const unsigned ob = rem - oa * s.fOB.d;
        return (int)((n * (unsigned)s.MajP + oa) * (unsigned)s.MinP + ob);
    };
    const int G0 = __builtin_amdgcn_readfirstlane(slot_of(m0));
    // this lane's slab row of its pixel in each of its MT pixel tiles (rows past the tensor: row 0 -- computed, never stored)
    int rowbase[MT];
#pragma unroll
    for (int ml = 0; ml < MT; ++ml) {
        const int m = m0 + grp * HM + ml * 16 + frow;
        rowbase[ml] = m < a.M ? slot_of(m) - G0 : 0;
    }
    // this lane's source of each of its wave's slab pieces: piece j = rows 8 * (wave + 8 j) .. + 7, row = slot G0 + row index
    unsigned slabva[SPW];
#pragma unroll
    for (int j = 0; j < SPW; ++j) {
        const int R = 8 * (wave + 8 * j) + lrow8;
        const unsigned F = (unsigned)(G0 + R);
        const unsigned n = fdiv(F, s.fImg);
        const unsigned rem = F - n * (unsigned)s.ImgP;
        const unsigned line = fdiv(rem, s.fLine);
        const unsigned col = rem - line * (unsigned)s.MinP;
        const int ia = (int)line - s.pa, ib = (int)col - s.pb;
        const bool v = (int)n < s.N && ia >= 0 && ib >= 0 && ia < s.IA && ib < s.IB;
        slabva[j] = v ? (unsigned)((int)n * s.simg + ia * s.sa + ib * s.sb) * (unsigned)ES + (unsigned)(csrc * 16) : OOB;
    }
    const int ncc = (a.C + BK - 1) / BK;
    const int cclim = (a.C - csrc * CE + BK - 1) / BK;      // chunks cc < cclim hold this lane's 8 channels
    // A last chunk of at most 32 channels (160 = 64 + 64 + 32) takes its taps in PAIRS: K-tile u of that chunk multiplies tap 2u in its
    // first 32 k and tap 2u + 1 in its second -- (TAPS + 1) / 2 K-tiles instead of TAPS half-empty ones
    constexpr int TAPSH = (TAPS + 1) / 2;
    const int tw = a.C - (ncc - 1) * BK;                    // channels of the last chunk (1..64)
    const bool tailpair = tw <= 32;
    const int nfull = tailpair ? ncc - 1 : ncc;             // chunks of TAPS K-tiles
    const int nk = nfull * TAPS + (tailpair ? TAPSH : 0);
    const bool pvalid = (csrc & 3) * CE < tw;               // paired K-tiles: this lane's 8 channels exist in the tail chunk
    const unsigned padd = (unsigned)((csrc >> 2) * (a.C * ES - 64));      // ... and chunks 4-7 fetch the NEXT tap's chunks 0-3
    unsigned woff[TN];
#pragma unroll
    for (int p = 0; p < TN; ++p) {
        const int n = n0 + p * 64 + wave * 8 + lrow8;
        woff[p] = n < a.K ? (unsigned)(n * a.Kg + csrc * CE) * (unsigned)ES : OOB;
    }
    // slab piece j of chunk CC into buffer CC & 1 (the caller has checked that this wave has a piece j: wave + 8 j < SLABP)
#define SLAB_ISSUE_PIECE(J, CC)                                                                                           \
    {                                                                                                                     \
        unsigned char* dst = smem + ((CC) & 1) * SLAB_BYTES + (wave + 8 * (J)) * 1024;                                    \
        lds_dma16(rsA, (lptr_t)dst, (CC) < cclim ? slabva[J] : OOB, (CC) * 128);                                          \
    }
    // filter tile of K-tile KT (chunk CC, first tap TT, paired?) into buffer KT & 1
#define SLAB_ISSUE_B(KT, CC, TT, PAIR)                                                                                    \
    {                                                                                                                     \
        const bool lv = (KT) < nk && ((PAIR) ? (pvalid && (csrc < 4 || (TT) + 1 < TAPS)) : (CC) < cclim);                 \
        _Pragma("unroll") for (int p = 0; p < TN; ++p) {                                                                  \
            unsigned char* dst = sFilt + ((KT) & 1) * BBUF + (p * 64 + wave * 8) * ROWB;                                  \
            const unsigned vo = (PAIR) ? woff[p] + padd : woff[p];                                                        \
            lds_dma16(rsB, (lptr_t)dst, (lv && woff[p] != OOB) ? vo : OOB, ((TT) * a.C + (CC) * BK) * ES);                \
        }                                                                                                                 \
    }
    // prologue: the slab of chunk 0 and the filter tile of K-tile 0
    const int npw = (SLABP - wave + 7) / 8;                // slab pieces of this wave (wave-uniform)
#pragma unroll
    for (int j = 0; j < SPW; ++j)
        if (j < npw) SLAB_ISSUE_PIECE(j, 0)
    SLAB_ISSUE_B(0, 0, 0, nfull == 0)

    unsigned faB[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int ph = ((kk * 4 + fchunk) ^ (frow & 7)) * 16;
        faB[kk] = (unsigned)(size_t)(lptr_t)(sFilt + (wc * 16 * TN + frow) * ROWB + ph);
    }
    const unsigned slab0 = (unsigned)(size_t)(lptr_t)smem;

    f32x4_t acc[MT][TN];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();        // group 1 runs one barrier behind group 0

    bf16x8_t fb[TN][2];
    bf16x8_t fa[PMX][2];
#pragma unroll
    for (int i = 0; i < TN; ++i) fb[i][0] = fb[i][1] = bf16x8_t{};
#pragma unroll
    for (int i = 0; i < PMX; ++i) fa[i][0] = fa[i][1] = bf16x8_t{};

    // fragment pair of pixel tile ML at slab row rowbase + SHA: chunk (kk * 4 + fchunk) ^ (row & 7) of the row; PAIRC (compile-time):
    // the second 32 k are chunks 0-3 of row rowbase + SHB (the next tap of a paired K-tile) instead of chunks 4-7 of the same row
#define SLAB_READ_A(PAIRC, DST, ML, SHA, SHB, SBASE)                                                                      \
    {                                                                                                                     \
        const int rowA = rowbase[ML] + (SHA);                                                                             \
        const unsigned adA = (SBASE) + (unsigned)(rowA * ROWB) + (unsigned)(((fchunk ^ rowA) & 7) << 4);                  \
        SLAB_DSREAD(fa[DST][0], adA);                                                                                     \
        if (PAIRC) {                                                                                                      \
            const int rowB = rowbase[ML] + (SHB);                                                                         \
            const unsigned adB = (SBASE) + (unsigned)(rowB * ROWB) + (unsigned)(((fchunk ^ rowB) & 7) << 4);              \
            SLAB_DSREAD(fa[DST][1], adB);                                                                                 \
        } else {                                                                                                          \
            SLAB_DSREAD(fa[DST][1], adA ^ 64u);                                                                           \
        }                                                                                                                 \
    }
    // One K-tile (both phases).  U: its index inside its chunk, LASTU: the chunk's last, MORE: a next chunk exists (its slab is
    // requested during this one), CCN: that chunk; NKT / NCC / NTT / NPAIR: the NEXT K-tile, whose filter tile is requested here
#define SLAB_KTILE(PAIRC, PARQ, SHA, SHB, SBASE, U, LASTU, MORE, CCN, NKT, NCC, NTT, NPAIR)                               \
    {                                                                                                                     \
        /* ================================================ even phase */                                                 \
        {                                                                                                                 \
            const unsigned bB0 = faB[0] + (PARQ) * BBUF, bB1 = faB[1] + (PARQ) * BBUF;                                    \
            _Pragma("unroll") for (int nt = 0; nt < TN; ++nt) {                                                           \
                if (nt == 0) { SLAB_DSREAD_OFF(fb[0][0], bB0, 0); SLAB_DSREAD_OFF(fb[0][1], bB1, 0); }                    \
                if (nt == 1) { SLAB_DSREAD_OFF(fb[1][0], bB0, 16 * ROWB); SLAB_DSREAD_OFF(fb[1][1], bB1, 16 * ROWB); }    \
                if (nt == 2) { SLAB_DSREAD_OFF(fb[2][0], bB0, 32 * ROWB); SLAB_DSREAD_OFF(fb[2][1], bB1, 32 * ROWB); }    \
                if (nt == 3) { SLAB_DSREAD_OFF(fb[3][0], bB0, 48 * ROWB); SLAB_DSREAD_OFF(fb[3][1], bB1, 48 * ROWB); }    \
            }                                                                                                             \
            _Pragma("unroll") for (int ml = 0; ml < PM0; ++ml) SLAB_READ_A(PAIRC, ml, ml, SHA, SHB, SBASE)                \
            SLAB_ISSUE_B(NKT, NCC, NTT, NPAIR)                                                                            \
            /* slab piece 2U - 1 of the next chunk (even phases of K-tiles 1 .. of a chunk) */                            \
            if ((MORE) && (U) >= 1 && 2 * (U) - 1 < npw) {                                                                \
                _Pragma("unroll") for (int j = 0; j < SPW; ++j)                                                           \
                    if (j == 2 * (U) - 1) SLAB_ISSUE_PIECE(j, CCN)                                                        \
            }                                                                                                             \
            /* (nothing this K-tile reads is still in flight: its filter tile was waited for in the previous odd phase, its  \
               slab a chunk ago) */                                                                                       \
            __builtin_amdgcn_s_barrier();                                                                                 \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                            \
            _Pragma("unroll") for (int nt = 0; nt < TN; ++nt) asm volatile("" : "+v"(fb[nt][0]), "+v"(fb[nt][1]));        \
            _Pragma("unroll") for (int ml = 0; ml < PM0; ++ml) asm volatile("" : "+v"(fa[ml][0]), "+v"(fa[ml][1]));       \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
            __builtin_amdgcn_s_setprio(1);                                                                                \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                              \
                _Pragma("unroll") for (int ml = 0; ml < PM0; ++ml)                                                        \
                    _Pragma("unroll") for (int nt = 0; nt < TN; ++nt)                                                     \
                        acc[ml][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt][kk], fa[ml][kk], acc[ml][nt], 0, 0, 0); \
            __builtin_amdgcn_s_setprio(0);                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
            __builtin_amdgcn_s_barrier();                                                                                 \
            asm volatile("" ::: "memory");                                                                                \
        }                                                                                                                 \
        /* ================================================ odd phase */                                                  \
        {                                                                                                                 \
            _Pragma("unroll") for (int ml = 0; ml < PM1; ++ml) SLAB_READ_A(PAIRC, ml, PM0 + ml, SHA, SHB, SBASE)          \
            /* slab piece 2U of the next chunk (odd phases of K-tiles 0 .. of a chunk, not its last), then: the filter tile \
               of the next K-tile has landed once at most the slab pieces requested behind it are outstanding */          \
            const bool so = (MORE) && 2 * (U) < npw && !(LASTU);                                                          \
            const bool se = (MORE) && (U) >= 1 && 2 * (U) - 1 < npw;                                                      \
            if (so) {                                                                                                     \
                _Pragma("unroll") for (int j = 0; j < SPW; ++j)                                                           \
                    if (j == 2 * (U)) SLAB_ISSUE_PIECE(j, CCN)                                                            \
            }                                                                                                             \
            if (LASTU) wait_vmcnt<0>();                     /* ... and the next chunk's slab, whole */                    \
            else if (so && se) wait_vmcnt<2>();                                                                           \
            else if (so || se) wait_vmcnt<1>();                                                                           \
            else wait_vmcnt<0>();                                                                                         \
            __builtin_amdgcn_s_barrier();                                                                                 \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                            \
            _Pragma("unroll") for (int ml = 0; ml < PM1; ++ml) asm volatile("" : "+v"(fa[ml][0]), "+v"(fa[ml][1]));       \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
            __builtin_amdgcn_s_setprio(1);                                                                                \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                              \
                _Pragma("unroll") for (int ml = 0; ml < PM1; ++ml)                                                        \
                    _Pragma("unroll") for (int nt = 0; nt < TN; ++nt)                                                     \
                        acc[PM0 + ml][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt][kk], fa[ml][kk], acc[PM0 + ml][nt], 0, 0, 0); \
            __builtin_amdgcn_s_setprio(0);                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
            __builtin_amdgcn_s_barrier();                                                                                 \
            asm volatile("" ::: "memory");                                                                                \
        }                                                                                                                 \
    }

    int kt = 0;
    // ---- chunks of TAPS K-tiles: tap t = (ta, tb) along (major, minor), filter order (r, s)
    for (int cc = 0; cc < nfull; ++cc) {
        const unsigned sbase = slab0 + (unsigned)((cc & 1) * SLAB_BYTES);
        const bool more = cc + 1 < ncc;
        int ta = 0, tb = 0;
#pragma unroll 1
        for (int tt = 0; tt < TAPS; ++tt, ++kt) {
            const unsigned par = (unsigned)(kt & 1);
            const int shift = ta * s.MinP + tb;
            const bool lastt = tt == TAPS - 1;
            const int ncc_ = lastt ? cc + 1 : cc, ntt_ = lastt ? 0 : tt + 1;
            const bool npair_ = lastt && cc + 1 == nfull;          // (only reached with a paired tail: otherwise kt + 1 == nk and the tile is dropped)
            SLAB_KTILE(false, par, shift, shift, sbase, tt, lastt, more, cc + 1, kt + 1, ncc_, ntt_, npair_)
            if (++tb == s.TB) { tb = 0; ++ta; }
        }
    }
    // ---- a last chunk of <= 32 channels: two taps per K-tile
    if (tailpair) {
        const int cc = ncc - 1;
        const unsigned sbase = slab0 + (unsigned)((cc & 1) * SLAB_BYTES);
        int ta = 0, tb = 0;
#pragma unroll 1
        for (int u = 0; u < TAPSH; ++u, ++kt) {
            const unsigned par = (unsigned)(kt & 1);
            const int shA = ta * s.MinP + tb;
            if (++tb == s.TB) { tb = 0; ++ta; }
            const int shB = 2 * u + 1 < TAPS ? ta * s.MinP + tb : shA;     // (an absent second tap: its filter half is zero, any finite pixels do)
            if (++tb == s.TB) { tb = 0; ++ta; }
            SLAB_KTILE(true, par, shA, shB, sbase, u, u == TAPSH - 1, false, cc, kt + 1, cc, 2 * u + 2, true)
        }
    }
#undef SLAB_KTILE
#undef SLAB_ISSUE_PIECE
#undef SLAB_ISSUE_B
#undef SLAB_READ_A
    if (grp == 0) __builtin_amdgcn_s_barrier();
    wait_vmcnt<0>();
    __syncthreads();
    {
        const int g4 = lane >> 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) {
                const int m = grp * HM + mt * 16 + frow;
                const int n = wc * (16 * TN) + nt * 16 + 4 * g4;
                Mma<T>::pack4(sC + m * LDC + n, acc[mt][nt]);
            }
    }
    __syncthreads();
    conv_epilogue_store<T, BM, BN, 512, MODE>(a, sC, sRed, t, lane, wave, m0, n0, mtile);
}

// IFCBK_CONV_SLAB: 0 = never, 1 = the layers it was measured on (default), 2 = wherever the kernel applies (tests)
int slab_mode() {
    const char* e = getenv("IFCBK_CONV_SLAB");
    return e ? atoi(e) : 1;
}

struct SlabGeom {
    int tr, TA, TB, IA, IB, OA, OB, pa, pb, MajP, MinP;
};

// geometry of the flat layout for a stride-1 gather (H, W input, P x Q output, R x S taps, gather padding ph, pw)
bool slab_geom(int H, int W, int P, int Q, int R, int S, int ph, int pw, SlabGeom* g) {
    if (ph < 0 || pw < 0 || ph > R - 1 || pw > S - 1) return false;
    if (P != H + 2 * ph - R + 1 || Q != W + 2 * pw - S + 1 || P < 1 || Q < 1) return false;
    g->tr = (S == 1 && R > 1) ? 1 : 0;
    if (g->tr) { g->TA = S; g->TB = R; g->IA = W; g->IB = H; g->OA = Q; g->OB = P; g->pa = pw; g->pb = ph; }
    else { g->TA = R; g->TB = S; g->IA = H; g->IB = W; g->OA = P; g->OB = Q; g->pa = ph; g->pb = pw; }
    g->MajP = g->IA + g->pa;
    g->MinP = g->IB + g->pb;
    return true;
}

// rows a tile's slab can need: BM valid pixels spread over lines (and across image ends), plus the halo of the farthest tap
int slab_rows(const SlabGeom& g, int bm) {
    const int lines = (bm + g.OB - 1) / g.OB + 1;                         // lines a run of bm pixels can touch
    const int imgs = (bm + g.OA * g.OB - 1) / (g.OA * g.OB) + 1;           // images it can touch
    return bm + lines * (g.MinP - g.OB) + (imgs - 1) * (g.MajP - g.OA) * g.MinP + (g.TA - 1) * g.MinP + (g.TB - 1) + 1;
}

struct SlabPlan { int mt, tn, taps, slabp; };

// instantiations: (MT, TN, TAPS, SLABP)
const SlabPlan kSlab[] = {
    {10, 3, 7, 50},     // 17x17 1x7 / 7x1, 192 output channels (and 160): 320-pixel tiles, <= 400 slab rows
    {10, 2, 7, 50},     // ... 128 output channels
};

int slab_find(int mt, int tn, int taps, int rows) {
    for (int i = 0; i < (int)(sizeof(kSlab) / sizeof(kSlab[0])); ++i)
        if (kSlab[i].mt == mt && kSlab[i].tn == tn && kSlab[i].taps == taps && rows <= 8 * kSlab[i].slabp) return i;
    return -1;
}

template <int TN, int MT, int TAPS, int SLABP>
void launch_slab(const ConvArgs& a, const SlabArgs& s, hipStream_t st) {
    dim3 grid((unsigned)(cdiv(a.M, 32 * MT) * a.tilesN)), block(512);
    constexpr int PM0 = MT == 10 ? 4 : MT / 2;
    if (a.bs_raw) hipLaunchKernelGGL((conv_slab<TN, MT, PM0, TAPS, SLABP, 3>), grid, block, 0, st, a, s);
    else hipLaunchKernelGGL((conv_slab<TN, MT, PM0, TAPS, SLABP, 0>), grid, block, 0, st, a, s);
}

}  // namespace

// Does the slab kernel serve this stride-1 gather?  (N images of H x W x C gathered with an R x S filter and padding (ph, pw) into
// P x Q x K.)  Returns the pixel tile in units of 32 pixels (the BatchNorm partial rows are cdiv(N*P*Q, 32 * mt)), 0: not served.
int ifcbk_conv_slab_plan(int dtype, int N, int H, int W, int C, int K, int R, int S, int ph, int pw, int P, int Q) {
    const int mode = slab_mode();
    if (mode <= 0 || dtype != IFCBK_BF16) return 0;
    SlabGeom g;
    if (!slab_geom(H, W, P, Q, R, S, ph, pw, &g)) return 0;
    const int taps = R * S;
    if (C % 8 || K % 8) return 0;
    const int64_t M = (int64_t)N * P * Q;
    if ((int64_t)(N + 1) * g.MajP * g.MinP + 4096 >= (1ll << 31)) return 0;
    const int mt = 10;
    const int tn = K <= 128 ? 2 : 3;
    if (K > 64 * tn) return 0;                              // one column tile: the slab is not shared between column tiles
    if (slab_find(mt, tn, taps, slab_rows(g, 32 * mt)) < 0) return 0;
    if (mode < 2) {
        // measured niche: the 17x17 stage's 7-tap layers (and anything of that shape class).  The layer's SHAPE decides, never the
        // batch (see the header): a batch of any size, its parts, and the training forward all sum in the same order
        if (K < 128 || C < 64) return 0;
    }
    (void)M;
    return mt;
}

int ifcbk_conv_slab_launch(ifcbk_ctx* ctx, void* args, int N, hipStream_t st) {
    ConvArgs& a = *reinterpret_cast<ConvArgs*>(args);
    SlabGeom g;
    if (!slab_geom(a.H, a.W, a.P, a.Q, a.R, a.S, -a.base_h, -a.base_w, &g)) IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv_slab: geometry");
    const int mt = 10, tn = a.K <= 128 ? 2 : 3, taps = a.R * a.S;
    const int i = slab_find(mt, tn, taps, slab_rows(g, 32 * mt));
    if (i < 0 || a.K > 64 * tn) IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv_slab: no instantiation for K=%d taps=%d", a.K, taps);
    SlabArgs s;
    s.MajP = g.MajP; s.MinP = g.MinP; s.ImgP = g.MajP * g.MinP;
    s.pa = g.pa; s.pb = g.pb; s.IA = g.IA; s.IB = g.IB; s.OA = g.OA; s.OB = g.OB;
    s.sa = g.tr ? a.ldx : a.W * a.ldx;
    s.sb = g.tr ? a.W * a.ldx : a.ldx;
    s.simg = a.H * a.W * a.ldx;
    s.TB = g.TB; s.tr = g.tr; s.N = N;
    s.fImg = make_fastdiv((uint32_t)s.ImgP); s.fLine = make_fastdiv((uint32_t)s.MinP);
    s.fOPQ = make_fastdiv((uint32_t)(g.OA * g.OB)); s.fOB = make_fastdiv((uint32_t)g.OB); s.fTB = make_fastdiv((uint32_t)g.TB);
    a.tilesN = 1;
    a.tr = g.tr;
    a.fP = make_fastdiv((uint32_t)a.P);
    switch (i) {
        case 0: launch_slab<3, 10, 7, 50>(a, s, st); break;
        default: launch_slab<2, 10, 7, 50>(a, s, st); break;
    }
    IFCBK_LAUNCH_CHECK(ctx, "conv_slab");
    return 0;
}
