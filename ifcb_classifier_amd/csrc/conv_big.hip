// Wide-tile implicit-GEMM convolution (forward / stride-1 input gradient), bf16, for the big-M layers of a training step:
// 256- or 320-pixel x 128..256-channel tiles, ONE 512-thread block per CU, eight waves as two GROUPS that run half a phase
// apart ("ping-pong"): while the four waves of one group sit in their MFMA cluster, the four of the other group (their SIMD
// partners: waves w and w+4 share a SIMD) read fragments from LDS and issue the next LDS-DMA pieces -- the matrix pipe and the
// load path of every SIMD stay busy at the same time, which a lock-step block (every wave loads, then every wave multiplies)
// cannot do with one block per CU.
//
//   C[m][n] = sum_k A[m][k] * B[n][k]     m = output pixel, n = output channel, k = (r,s,c); see conv_igemm.hip
//
// Geometry.  Wave w: group g = w >> 2 owns the pixel half [g*16*MT, (g+1)*16*MT) of the tile, column quarter wc = w & 3 owns
// 16*TN channels: MT x TN accumulator tiles of 16x16 per wave (MT = 8 or 10: 256 / 320 pixels, TN = 2..4: 128..256 channels).
// A 320-pixel tile exists for the 17x17 maps: 256 images x 289 pixels = 73,984 rows are 289 tiles of 256 (a second, nearly
// empty round on 256 CUs) but 232 tiles of 320 -- one round.
//
// A K-tile (64 k) is consumed in TWO phases of PM0 and PM1 = MT - PM0 pixel tiles (conv_pp2 below has the schedule); the pixel
// operand lives in two slots per K-tile parity, the filter operand in two K-tile buffers, both filled by LDS-DMA pieces of 1 KiB
// (8 rows of 128 B) that the waves issue in their load half-phase and wait for with a COUNTED vmcnt.
//
// Ordering (physical barriers are numbered; group 1 executes one extra barrier up front, group 0 one at the end):
//   group 0:  load(g) | #2g | mfma(g) | #2g+1 | load(g+1) ...        group 1:  #2g | load(g) | #2g+1 | mfma(g) | #2g+2 ...
//   RAW: a piece read in load(g+1) was waited for (vmcnt) at the end of load(g) by the wave that issued it, i.e. before #2g
//        (group 0) / #2g+1 (group 1); the earliest reader (group 0, after #2g+1) has passed both.
//   WAR: a slot read in load(g) is re-filled by pieces issued in load(h), h >= g + 2: the latest reader (group 1) has its
//        fragments in registers (lgkmcnt(0)) right after #2g+1, the earliest writer (group 0 in load(g+2)) starts after #2g+3.
// Fragment reads are inline-asm ds_read_b128 (the compiler orders every LDS access it can see behind ALL pending LDS-DMA,
// which would drain the ring every phase); their lgkmcnt(0) sits behind the barrier, in front of the MFMA cluster.
//
// (Round 2 also had a finer-grained form, `conv_big`: MT/2 phases of 4*TN MFMAs per K-tile with a 2*NPH-slot ring.  It was the
// kernel the measurements quoted below were first made on, ran at the same speed as the two-phase form and was removed in
// round 3; DESIGN.md 5.4 keeps its numbers.)
#include "conv_common.h"
#include <stdlib.h>
#include <string.h>

namespace {

#define BIG_DSREAD(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))

// ---------------------------------------------------------------- the kernel: two long phases per K-tile
// Measured on the finer-grained predecessor (loads dropped through zero-record descriptors): the operand traffic is
// NOT what bounds it (6e 7x1: 52.8 us -> 48.0 us with both operands dropped) -- the half-phase between two barriers is as long
// as the LOAD side needs (fragment reads, piece issue, address VALU, waits: ~400 cycles) while the MFMA side has only 4*TN
// MFMAs (192-256 cycles) to set against it.  Here a K-tile has TWO phases, of PM0 and PM1 = MT - PM0 pixel tiles: 2*PM*TN
// MFMAs (384-576 cycles) per half-phase against about the same load work as before -- half the barriers per MFMA.
//   phase 2T   (even): reads the filter fragments of K-tile T and the pixel tiles [0, PM0); issues the EVEN pixel slot and the
//                      filter pieces of K-tile T+1 (PM0/2 + TN pieces per wave); then vmcnt(PM0/2 + TN): the odd slot of T is in
//   phase 2T+1 (odd):  reads pixel tiles [PM0, MT); issues the ODD pixel slot of K-tile T+1 (PM1/2 pieces); then vmcnt(PM1/2)
// Slots are double-buffered by K-tile parity; every slot is re-filled two phases after its last read (the WAR rule above).
// timing experiment (built with `make EXTRA=-DIFCBK_EXPERIMENT_FRAG_AFFINE`, run with IFCBK_DEBUG_DROP=f; wrong results): what BatchNorm-apply + ReLU on the pixel fragment AFTER its LDS read
// would cost -- the producer's activation never written, the consumer normalises what it multiplies.
// Optimistic: scale / shift are lane constants here; a real version adds two LDS reads per fragment for the per-channel pair
// and a border mask for padded taps.
__device__ __forceinline__ bf16x8_t dbg_affine_relu(bf16x8_t v, float sc, float sh) {
    u32x4_t u = __builtin_bit_cast(u32x4_t, v);
    u32x4_t r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float lo = __uint_as_float(u[j] << 16), hi = __uint_as_float(u[j] & 0xffff0000u);
        lo = fmaxf(fmaf(lo, sc, sh), 0.f);
        hi = fmaxf(fmaf(hi, sc, sh), 0.f);
        r[j] = pack2bf(lo, hi);
    }
    return __builtin_bit_cast(bf16x8_t, r);
}

template <int TN, int MT, int PM0, int MODE>
__global__ __launch_bounds__(512) void conv_pp2(ConvArgs a) {
    typedef bf16_t T;
    constexpr int ES = 2, CE = 8, BK = 64;
    constexpr int PM1 = MT - PM0;
    static_assert(PM0 % 2 == 0 && PM1 % 2 == 0 && PM0 > 0 && PM1 > 0, "whole pieces per wave");
    constexpr int PMX = PM0 > PM1 ? PM0 : PM1;
    constexpr int NA0 = PM0 / 2, NA1 = PM1 / 2;           // pixel pieces per wave in the even / odd phase
    constexpr int HM = 16 * MT, BM = 2 * HM, BN = 64 * TN;
    constexpr int ROWB = BK * ES;                          // 128 bytes per tile row
    constexpr int E_BYTES = 2 * PM0 * 16 * ROWB, O_BYTES = 2 * PM1 * 16 * ROWB;      // even / odd pixel slot
    constexpr int APAR = E_BYTES + O_BYTES;                // pixel bytes per K-tile parity (= BM rows)
    constexpr int BBUF = BN * ROWB;
    constexpr int A_BYTES = 2 * APAR, RING_BYTES = A_BYTES + 2 * BBUF;
    constexpr int LDC = BN + CE;
    constexpr int CT_BYTES = BM * LDC * ES + 8 * BN * 2 * 4;
    constexpr int SMEM_BYTES = RING_BYTES > CT_BYTES ? RING_BYTES : CT_BYTES;
    static_assert(SMEM_BYTES <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
    T* sC = reinterpret_cast<T*>(smem);
    float* sRed = reinterpret_cast<float*>(smem + BM * LDC * ES);

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
    const int csrc = (lane & 7) ^ lrow8;
    // pixel rows of this lane: piece i of the even slot = slot rows 8*(wave + 8*i) .., likewise the odd slot
    constexpr int NPX = NA0 + NA1;
    int off0[NPX], bh[NPX], bw[NPX];
    unsigned va[NPX];
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
        const bool odd = i >= NA0;
        const int pm = odd ? PM1 : PM0;
        const int srow = 8 * (wave + 8 * (odd ? i - NA0 : i)) + lrow8;       // row inside the slot: [half][pm*16]
        const int half = srow / (16 * pm), rr = srow - half * 16 * pm;
        const int m = m0 + half * HM + (odd ? 16 * PM0 : 0) + rr;
        const bool rv = m < a.M;
        const int mm = rv ? m : 0;
        const int n = (int)fdiv((uint32_t)mm, a.fPQ);
        const int rem = mm - n * a.PQ;
        const int p = (int)fdiv((uint32_t)rem, a.fQ);
        const int q = rem - p * a.Q;
        bh[i] = rv ? p * a.ostr_h + a.base_h : -(1 << 24);
        bw[i] = q * a.ostr_w + a.base_w;
        off0[i] = ((n * a.H + bh[i]) * a.W + bw[i]) * a.ldx;
        va[i] = bh[i] >= 0 ? (unsigned)(off0[i] + csrc * CE) * (unsigned)ES : OOB;
    }
    int kc = csrc * CE, kr = 0, ks = 0;
    while (kc >= a.C) {
        kc -= a.C;
        if (++ks == a.S) { ks = 0; ++kr; }
    }
    int tapoff = (kr * a.W + ks) * a.ldx + kc;
    int ktA = 0;                                       // K-tile of the next pixel pieces
    unsigned woff[TN];
#pragma unroll
    for (int p = 0; p < TN; ++p) {
        const int n = n0 + p * 64 + wave * 8 + lrow8;
        woff[p] = n < a.K ? (unsigned)(n * a.Kg + csrc * CE) * (unsigned)ES : OOB;
    }
    const int nk = (a.Kg + BK - 1) / BK;
    const int rowstep = a.W * a.ldx, colwrap = a.S * a.ldx;
    const bool plain = a.R == 1 && a.S == 1 && a.base_h == 0 && a.base_w == 0;
    const bool ktail_ok = (nk - 1) * BK + csrc * CE < a.Kg;

    // pixel pieces [I0, I0+CNT) of K-tile ktA (even slot: I0 = 0, odd: I0 = NA0) into their slot rows
#define PP_ISSUE_A(I0, CNT, SLOT_OFF)                                                                                     \
    {                                                                                                                     \
        unsigned char* slot = smem + (ktA & 1) * APAR + (SLOT_OFF);                                                       \
        const bool kvalid = kr < a.R;                                                                                     \
        const bool cut = ktA >= nk || (ktA == nk - 1 && !ktail_ok);                                                       \
        _Pragma("unroll") for (int i = 0; i < (CNT); ++i) {                                                               \
            unsigned char* dst = slot + (wave + 8 * i) * 8 * ROWB;                                                        \
            if (plain) {                                                                                                  \
                lds_dma16(rsA, (lptr_t)dst, cut ? OOB : va[(I0) + i], ktA * 128);                                         \
            } else {                                                                                                      \
                const int hr = bh[(I0) + i] + kr, wr = bw[(I0) + i] + ks;                                                 \
                const bool v = kvalid && (unsigned)hr < (unsigned)a.H && (unsigned)wr < (unsigned)a.W;                    \
                const unsigned voff = v ? (unsigned)(off0[(I0) + i] + tapoff) * (unsigned)ES : OOB;                       \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)dst, 16, voff, 0, 0, 0);                            \
            }                                                                                                             \
        }                                                                                                                 \
    }
#define PP_ADVANCE_K()                                                                                                    \
    {                                                                                                                     \
        ++ktA;                                                                                                            \
        if (!plain) {                                                                                                     \
            kc += BK;                                                                                                     \
            tapoff += BK;                                                                                                 \
            while (kc >= a.C) {                                                                                           \
                kc -= a.C;                                                                                                \
                tapoff += a.ldx - a.C;                                                                                    \
                if (++ks == a.S) { ks = 0; ++kr; tapoff += rowstep - colwrap; }                                           \
            }                                                                                                             \
        }                                                                                                                 \
    }
#define PP_ISSUE_B(KT)                                                                                                    \
    {                                                                                                                     \
        _Pragma("unroll") for (int p = 0; p < TN; ++p) {                                                                  \
            unsigned char* dst = smem + A_BYTES + ((KT) & 1) * BBUF + (p * 64 + wave * 8) * ROWB;                          \
            lds_dma16(rsB, (lptr_t)dst, (KT) < nk ? woff[p] : OOB, (KT) * 128);                                           \
        }                                                                                                                 \
    }
    // virtual phases -2 (even: K-tile 0's even slot + filter tile) and -1 (odd slot), then the first real wait + barrier
    PP_ISSUE_A(0, NA0, 0)
    PP_ISSUE_B(0)
    PP_ISSUE_A(NA0, NA1, E_BYTES)
    PP_ADVANCE_K()

    const int frow = lane & 15, fchunk = lane >> 4;
    unsigned faE[2], faO[2], faB[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int ph = ((kk * 4 + fchunk) ^ (frow & 7)) * 16;
        faE[kk] = (unsigned)(size_t)(lptr_t)(smem + (grp * 16 * PM0 + frow) * ROWB + ph);
        faO[kk] = (unsigned)(size_t)(lptr_t)(smem + E_BYTES + (grp * 16 * PM1 + frow) * ROWB + ph);
        faB[kk] = (unsigned)(size_t)(lptr_t)(smem + A_BYTES + (wc * 16 * TN + frow) * ROWB + ph);
    }

    f32x4_t acc[MT][TN];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    wait_vmcnt<NA1>();                                 // even slot + filter tile of K-tile 0 have landed
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();        // group 1 runs one barrier behind group 0

    bf16x8_t fb[TN][2];
    bf16x8_t fa[PMX][2];
    const bool dbg_nomfma = a.dbg & 1, dbg_noread = a.dbg & 2, dbg_nodma = a.dbg & 4;
#ifdef IFCBK_EXPERIMENT_FRAG_AFFINE
    const float dbg_sc = 1.0f + 1e-3f * (float)(lane & 7), dbg_sh = 1e-3f * (float)(lane >> 3);
#endif
#pragma unroll
    for (int i = 0; i < TN; ++i) fb[i][0] = fb[i][1] = bf16x8_t{};
#pragma unroll
    for (int i = 0; i < PMX; ++i) fa[i][0] = fa[i][1] = bf16x8_t{};
    for (int kt = 0; kt < nk; ++kt) {
        const unsigned par = (unsigned)(kt & 1);
        // ================================================ even phase
        {
            const unsigned bB0 = faB[0] + par * BBUF, bB1 = faB[1] + par * BBUF;
            const unsigned bA0 = faE[0] + par * APAR, bA1 = faE[1] + par * APAR;
            if (!dbg_noread) {
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) {
                if (nt == 0) { BIG_DSREAD(fb[0][0], bB0, 0); BIG_DSREAD(fb[0][1], bB1, 0); }
                if (nt == 1) { BIG_DSREAD(fb[1][0], bB0, 16 * ROWB); BIG_DSREAD(fb[1][1], bB1, 16 * ROWB); }
                if (nt == 2) { BIG_DSREAD(fb[2][0], bB0, 32 * ROWB); BIG_DSREAD(fb[2][1], bB1, 32 * ROWB); }
                if (nt == 3) { BIG_DSREAD(fb[3][0], bB0, 48 * ROWB); BIG_DSREAD(fb[3][1], bB1, 48 * ROWB); }
            }
#pragma unroll
            for (int ml = 0; ml < PM0; ++ml) {
                if (ml == 0) { BIG_DSREAD(fa[0][0], bA0, 0); BIG_DSREAD(fa[0][1], bA1, 0); }
                if (ml == 1) { BIG_DSREAD(fa[1][0], bA0, 16 * ROWB); BIG_DSREAD(fa[1][1], bA1, 16 * ROWB); }
                if (ml == 2) { BIG_DSREAD(fa[2][0], bA0, 32 * ROWB); BIG_DSREAD(fa[2][1], bA1, 32 * ROWB); }
                if (ml == 3) { BIG_DSREAD(fa[3][0], bA0, 48 * ROWB); BIG_DSREAD(fa[3][1], bA1, 48 * ROWB); }
                if (ml == 4) { BIG_DSREAD(fa[4][0], bA0, 64 * ROWB); BIG_DSREAD(fa[4][1], bA1, 64 * ROWB); }
                if (ml == 5) { BIG_DSREAD(fa[5][0], bA0, 80 * ROWB); BIG_DSREAD(fa[5][1], bA1, 80 * ROWB); }
            }
            }
            if (!dbg_nodma) {
            PP_ISSUE_A(0, NA0, 0)                       // K-tile kt+1 (ktA), even slot
            PP_ISSUE_B(kt + 1)
            }
            wait_vmcnt<NA0 + TN>();                     // everything older than this phase's pieces: the odd slot of K-tile kt
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) asm volatile("" : "+v"(fb[nt][0]), "+v"(fb[nt][1]));
#pragma unroll
            for (int ml = 0; ml < PM0; ++ml) asm volatile("" : "+v"(fa[ml][0]), "+v"(fa[ml][1]));
#ifdef IFCBK_EXPERIMENT_FRAG_AFFINE
            if (a.dbg & 16) {
#pragma unroll
                for (int ml = 0; ml < PM0; ++ml) {
                    fa[ml][0] = dbg_affine_relu(fa[ml][0], dbg_sc, dbg_sh);
                    fa[ml][1] = dbg_affine_relu(fa[ml][1], dbg_sc, dbg_sh);
                }
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            if (!dbg_nomfma)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int ml = 0; ml < PM0; ++ml)
#pragma unroll
                    for (int nt = 0; nt < TN; ++nt)
                        acc[ml][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt][kk], fa[ml][kk], acc[ml][nt], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        // ================================================ odd phase
        {
            const unsigned bA0 = faO[0] + par * APAR, bA1 = faO[1] + par * APAR;
            if (!dbg_noread) {
#pragma unroll
            for (int ml = 0; ml < PM1; ++ml) {
                if (ml == 0) { BIG_DSREAD(fa[0][0], bA0, 0); BIG_DSREAD(fa[0][1], bA1, 0); }
                if (ml == 1) { BIG_DSREAD(fa[1][0], bA0, 16 * ROWB); BIG_DSREAD(fa[1][1], bA1, 16 * ROWB); }
                if (ml == 2) { BIG_DSREAD(fa[2][0], bA0, 32 * ROWB); BIG_DSREAD(fa[2][1], bA1, 32 * ROWB); }
                if (ml == 3) { BIG_DSREAD(fa[3][0], bA0, 48 * ROWB); BIG_DSREAD(fa[3][1], bA1, 48 * ROWB); }
                if (ml == 4) { BIG_DSREAD(fa[4][0], bA0, 64 * ROWB); BIG_DSREAD(fa[4][1], bA1, 64 * ROWB); }
                if (ml == 5) { BIG_DSREAD(fa[5][0], bA0, 80 * ROWB); BIG_DSREAD(fa[5][1], bA1, 80 * ROWB); }
            }
            }
            if (!dbg_nodma) {
            PP_ISSUE_A(NA0, NA1, E_BYTES)               // K-tile kt+1 (ktA), odd slot
            }
            PP_ADVANCE_K()
            wait_vmcnt<NA1>();                          // the even slot and the filter tile of K-tile kt+1 have landed
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int ml = 0; ml < PM1; ++ml) asm volatile("" : "+v"(fa[ml][0]), "+v"(fa[ml][1]));
#ifdef IFCBK_EXPERIMENT_FRAG_AFFINE
            if (a.dbg & 16) {
#pragma unroll
                for (int ml = 0; ml < PM1; ++ml) {
                    fa[ml][0] = dbg_affine_relu(fa[ml][0], dbg_sc, dbg_sh);
                    fa[ml][1] = dbg_affine_relu(fa[ml][1], dbg_sc, dbg_sh);
                }
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            if (!dbg_nomfma)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int ml = 0; ml < PM1; ++ml)
#pragma unroll
                    for (int nt = 0; nt < TN; ++nt)
                        acc[PM0 + ml][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt][kk], fa[ml][kk], acc[PM0 + ml][nt], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
#undef PP_ISSUE_A
#undef PP_ISSUE_B
#undef PP_ADVANCE_K
    if (grp == 0) __builtin_amdgcn_s_barrier();
    wait_vmcnt<0>();
    __syncthreads();
    if (a.dbg & 8) {                                   // timing-only: no epilogue (one store keeps the accumulators alive)
        float sacc = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) sacc += acc[mt][nt][0] + acc[mt][nt][3];
        if (sacc == 12345.678f) reinterpret_cast<float*>(a.y)[t] = sacc;
        return;
    }
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

// IFCBK_CONV_BIG: 0 = never, 1 = where the plan below expects a gain (default), 2 = wherever the kernel applies (tests force
// it onto small shapes).  Read on every call -- a getenv per conv launch is noise next to the launch itself.
int big_mode() {
    const char* e = getenv("IFCBK_CONV_BIG");
    return e ? atoi(e) : 1;
}
int big_force(const char* name) {
    const char* e = getenv(name);
    return e ? atoi(e) : 0;
}

template <int TN, int MT>
void launch_big(const ConvArgs& a, hipStream_t st) {
    dim3 grid((unsigned)(cdiv(a.M, 32 * MT) * a.tilesN)), block(512);
    constexpr int PM0 = MT == 10 ? 4 : MT / 2;
    if (a.seg_n) {                                       // eval-mode sibling GEMM: per-segment destinations (conv_epilogue_store MODE 4)
        hipLaunchKernelGGL((conv_pp2<TN, MT, PM0, 4>), grid, block, 0, st, a);
        return;
    }
    if (a.bs_tab) hipLaunchKernelGGL((conv_pp2<TN, MT, PM0, 5>), grid, block, 0, st, a);
    else if (a.bs_raw) hipLaunchKernelGGL((conv_pp2<TN, MT, PM0, 3>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((conv_pp2<TN, MT, PM0, 0>), grid, block, 0, st, a);
}

}  // namespace

// Tile choice for the wide-tile kernel; returns false when conv_igemm / conv_ws should serve the GEMM (M pixels, K output
// channels, Kg reduction length).  One block per CU: the cost of a candidate is rounds x (phases per K-tile) x (MFMAs per
// phase + a fixed part for the phase's loads and barriers); the grid must fill most of the chip.
bool ifcbk_conv_big_plan(int dtype, int M, int K, int Kg, int* mt_out, int* tn_out) {
    const int mode = big_mode();
    if (mode <= 0 || dtype != IFCBK_BF16) return false;
    const int fmt = big_force("IFCBK_CONV_BIG_MT"), ftn = big_force("IFCBK_CONV_BIG_TN");
    const int nk = cdiv(Kg, 64);
    // measured per layer at batch 256 (scripts/conv_big_check.py, interleaved rounds against conv_igemm): +20-25 % on the 17x17
    // layers with 192 output channels (320 x 192 tiles: one round of 232 blocks instead of 578 on 512 slots), +7-25 % on the
    // fused sibling 1x1 GEMMs (256 x 256); SLOWER with 128-channel tiles (0.7-0.9x) and on grids of many rounds (Conv2d_4a,
    // 20 rounds: 0.83x -- one block per CU exposes every tile's prologue and epilogue, two co-resident blocks hide them)
    if (mode < 2 && (nk < 4 || K < 136 || (int64_t)M < 192 * 256)) return false;
    const int cus = ifcbk_num_cus();
    double best = 0;
    int bmt = 0, btn = 0;
    for (int mt = 8; mt <= 10; mt += 2)
        for (int tn = (mode < 2 ? 3 : 2); tn <= 4; ++tn) {
            if ((fmt && mt != fmt) || (ftn && tn != ftn)) continue;
            if (mt == 10 && tn == 4) continue;                      // 160 accumulator registers: not built
            const int64_t tiles = (int64_t)cdiv(M, 32 * mt) * cdiv(K, 64 * tn);
            const double rounds = (double)cdiv(tiles, cus);
            const double cost = rounds * (mt / 2) * (4.0 * tn + 6.0);
            if (!bmt || cost < best) { best = cost; bmt = mt; btn = tn; }
        }
    if (!bmt) return false;
    const int64_t tiles = (int64_t)cdiv(M, 32 * bmt) * cdiv(K, 64 * btn);
    // grids of many rounds: only with the 256-channel tile (the sibling 1x1 GEMMs of large RUN batches: one pass over the block input
    // instead of two or three 128-channel column tiles -- at batch 2048 Mixed_5b-5d / 6b-6e 0.76-1.00 -> 0.69-0.90 ms each; the
    // 192-channel tiles lose there: Conv2d_4a 0.95x, Mixed_7b / 7c 0.87-0.89x)
    if (mode < 2 && (tiles < (3 * cus) / 4 || (tiles > 6 * cus && btn != 4))) return false;
    if (mt_out) *mt_out = bmt;
    if (tn_out) *tn_out = btn;
    return true;
}

int ifcbk_conv_big_launch(ifcbk_ctx* ctx, void* args, int mt, int tn, hipStream_t st) {
    ConvArgs& a = *reinterpret_cast<ConvArgs*>(args);
    a.tilesN = cdiv(a.K, 64 * tn);
    // timing-only diagnostics (wrong results): a descriptor with zero records drops every load through it while the
    // instruction stream, the waits and the barriers stay -- what does one operand's traffic cost?
#ifdef IFCBK_EXPERIMENT_DROP          // make EXTRA=-DIFCBK_EXPERIMENT_DROP: never in the shipped library (a stray variable must not corrupt results)
    if (const char* e = getenv("IFCBK_DEBUG_DROP")) {
        if (strchr(e, 'a')) a.xbytes = 0;
        if (strchr(e, 'b')) a.wbytes = 0;
        if (strchr(e, 'm')) a.dbg |= 1;           // no MFMAs
        if (strchr(e, 'r')) a.dbg |= 2;           // no fragment reads
        if (strchr(e, 'd')) a.dbg |= 4;           // no LDS-DMA pieces
        if (strchr(e, 'e')) a.dbg |= 8;           // no epilogue
        if (strchr(e, 'f')) a.dbg |= 16;          // affine + ReLU on every pixel fragment after its LDS read
    }
#endif
    if (mt == 8 && tn == 2) launch_big<2, 8>(a, st);
    else if (mt == 8 && tn == 3) launch_big<3, 8>(a, st);
    else if (mt == 8 && tn == 4) launch_big<4, 8>(a, st);
    else if (mt == 10 && tn == 2) launch_big<2, 10>(a, st);
    else if (mt == 10 && tn == 3) launch_big<3, 10>(a, st);
    else IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv_big: no instantiation for mt=%d tn=%d", mt, tn);
    IFCBK_LAUNCH_CHECK(ctx, "conv_big");
    return 0;
}
