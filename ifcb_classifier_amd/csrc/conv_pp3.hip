// Persistent wide-tile implicit-GEMM convolution (bf16, gfx950), round 4: conv_pp2's ping-pong main loop (conv_big.hip) for grids
// of SEVERAL tiles per CU, with the per-tile fixed cost taken out.
//
// conv_pp2 lives for one tile: launch + address set-up + the first LDS-DMA round trip (5 us), then the main loop, then an epilogue
// (C tile through LDS, 7.5-9 us) during which nothing multiplies -- 20 of 51 us on a 17x17 layer, and every tile of a
// many-round grid (Conv2d_4a: 20 rounds; the fused sibling 1x1 GEMMs: 3-4; every layer of a batch-1024 RUN forward) pays it again.
// Here a block walks tiles  t = first, first + grid, ...  and
//   * the pipeline never drains at a tile boundary: in the LAST K-tile of tile i the pieces a conv_pp2 block would request for the
//     (non-existent) K-tile nk are the pieces of K-tile 0 of tile i+1 -- the ring parity simply runs on; the per-lane gather
//     addresses are switched to the next tile in that K-tile's load part;
//   * the epilogue is register-direct and DEFERRED: at the boundary a wave packs its accumulators to bf16 (48 registers for the
//     256 x 192 tile; BatchNorm statistics of the rounded values by DPP row reductions, or the folded eval affine + ReLU), zeroes
//     them and goes on multiplying; the packed tile leaves as one burst of 8-byte buffer stores (one per 16 x 16 tile) issued
//     behind the LDS-DMA pieces of the next tile's first phase, so that the counted vmcnt waits of the following phases do not
//     wait for them.  Only the last tile of a block has an exposed (register-direct) epilogue.
// Statistics: one partial row per (tile, pixel half): ifcbk_conv2d_fwd_mblocks = 2 per M tile (as conv_ws).
//
// Tile 256 pixels x 192 channels (MT = 8, TN = 3): 96 accumulator + 48 packed + 56 fragment registers.  Serves the plain gathers
// (forward of any stride, first-writer stride-1 input gradients) whose epilogue is a raw store (+ statistics) or the eval affine
// (+ReLU); everything else (accumulate, residual, BN-backward sums, segments, stride-2 classes) stays on conv_pp2 / conv_igemm.
#include "conv_common.h"
#include <stdlib.h>

namespace {

#define P3_DSREAD(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

// sum over the 16 lanes of a DPP row (all lanes end up with it): quad xor 1, quad xor 2, half-row mirror, row mirror
__device__ __forceinline__ float row16_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, false));   // row_half_mirror
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, false));   // row_mirror
    return x;
}

// EPI 0: raw store (+ BatchNorm statistics when a.part); EPI 1: y = act(acc * scale[n] + shift[n])
template <int TN, int MT, int PM0, int EPI, bool PLAIN>
__global__ __launch_bounds__(512) void conv_pp3(ConvArgs a, int ntiles, unsigned ybytes) {
    constexpr int ES = 2, CE = 8, BK = 64;
    constexpr int PM1 = MT - PM0;
    static_assert(PM0 % 2 == 0 && PM1 % 2 == 0 && PM0 > 0 && PM1 > 0, "whole pieces per wave");
    constexpr int PMX = PM0 > PM1 ? PM0 : PM1;
    constexpr int NA0 = PM0 / 2, NA1 = PM1 / 2, NPX = NA0 + NA1;
    constexpr int HM = 16 * MT, BM = 2 * HM, BN = 64 * TN;
    constexpr int ROWB = BK * ES;
    constexpr int E_BYTES = 2 * PM0 * 16 * ROWB, O_BYTES = 2 * PM1 * 16 * ROWB;
    constexpr int APAR = E_BYTES + O_BYTES;
    constexpr int BBUF = BN * ROWB;
    constexpr int A_BYTES = 2 * APAR, RING_BYTES = A_BYTES + 2 * BBUF;
    constexpr int KMAX = 2048;                           // channels whose scale / shift are staged in LDS (EPI 1)
    constexpr int SS_BYTES = EPI == 1 ? 2 * KMAX * 4 : 0;
    constexpr int NST = MT * TN;                         // deferred stores per wave and tile
    static_assert(RING_BYTES + SS_BYTES <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) unsigned char smem[RING_BYTES + SS_BYTES];
    float* sS = reinterpret_cast<float*>(smem + RING_BYTES);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int grp = wave >> 2, wc = wave & 3;
    const int nk = (a.Kg + BK - 1) / BK;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, ybytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int lrow8 = lane >> 3;
    const int csrc = (lane & 7) ^ lrow8;
    const int frow = lane & 15, fchunk = lane >> 4, g4 = lane >> 4;

    if (EPI == 1) {
        for (int i = t; i < a.K; i += 512) {
            sS[i] = a.ep_scale[i];
            sS[KMAX + i] = a.ep_shift[i];
        }
        __syncthreads();
    }

    // ---- gather state of the tile whose pieces are being requested
    int tile = (int)xcd_remap(blockIdx.x, gridDim.x);
    int m0 = 0, n0 = 0, mtile = 0;
    int off0[NPX], bh[NPX], bw[NPX];
    unsigned va[NPX];
    unsigned woff[TN];
    int kc = 0, kr = 0, ks = 0, tapoff = 0, ktA = 0;
    bool dead = false;                                   // no tile left: request nothing (dropped loads keep the counts)
    const int rowstep = a.W * a.ldx, colwrap = a.S * a.ldx;
    const bool ktail_ok = (nk - 1) * BK + csrc * CE < a.Kg;

#define P3_SETUP()                                                                                                        \
    {                                                                                                                     \
        mtile = tile / a.tilesN;                                                                                          \
        const int ntile = tile - mtile * a.tilesN;                                                                        \
        m0 = mtile * BM;                                                                                                  \
        n0 = ntile * BN;                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < NPX; ++i) {                                                                 \
            const bool odd = i >= NA0;                                                                                    \
            const int pm = odd ? PM1 : PM0;                                                                               \
            const int srow = 8 * (wave + 8 * (odd ? i - NA0 : i)) + lrow8;                                                \
            const int half = srow / (16 * pm), rr = srow - half * 16 * pm;                                                \
            const int m = m0 + half * HM + (odd ? 16 * PM0 : 0) + rr;                                                     \
            const bool rv = m < a.M;                                                                                      \
            const int mm = rv ? m : 0;                                                                                    \
            const int n = (int)fdiv((uint32_t)mm, a.fPQ);                                                                 \
            const int rem = mm - n * a.PQ;                                                                                \
            const int p = (int)fdiv((uint32_t)rem, a.fQ);                                                                 \
            const int q = rem - p * a.Q;                                                                                  \
            const int bhh = rv ? p * a.ostr_h + a.base_h : -(1 << 24);                                                    \
            const int bww = q * a.ostr_w + a.base_w;                                                                      \
            const int o0 = ((n * a.H + bhh) * a.W + bww) * a.ldx;                                                         \
            if (!PLAIN) { bh[i] = bhh; bw[i] = bww; off0[i] = o0; }                                                       \
            va[i] = bhh >= 0 ? (unsigned)(o0 + csrc * CE) * (unsigned)ES : OOB;                                           \
        }                                                                                                                 \
        _Pragma("unroll") for (int p = 0; p < TN; ++p) {                                                                  \
            const int n = n0 + p * 64 + wave * 8 + lrow8;                                                                 \
            woff[p] = n < a.K ? (unsigned)(n * a.Kg + csrc * CE) * (unsigned)ES : OOB;                                    \
        }                                                                                                                 \
        kc = csrc * CE; kr = 0; ks = 0;                                                                                   \
        while (kc >= a.C) {                                                                                               \
            kc -= a.C;                                                                                                    \
            if (++ks == a.S) { ks = 0; ++kr; }                                                                            \
        }                                                                                                                 \
        tapoff = (kr * a.W + ks) * a.ldx + kc;                                                                            \
        ktA = 0;                                                                                                          \
    }
    // pixel pieces [I0, I0+CNT) of K-tile ktA of the gather tile into the slot of parity PARQ
#define P3_ISSUE_A(I0, CNT, SLOT_OFF, PARQ)                                                                               \
    {                                                                                                                     \
        unsigned char* slot = smem + (PARQ) * APAR + (SLOT_OFF);                                                          \
        const bool kvalid = kr < a.R;                                                                                     \
        const bool cut = dead || ktA >= nk || (ktA == nk - 1 && !ktail_ok);                                               \
        _Pragma("unroll") for (int i = 0; i < (CNT); ++i) {                                                               \
            unsigned char* dst = slot + (wave + 8 * i) * 8 * ROWB;                                                        \
            if (PLAIN) {                                                                                                  \
                lds_dma16(rsA, (lptr_t)dst, cut ? OOB : va[(I0) + i], ktA * 128);                                         \
            } else {                                                                                                      \
                const int hr = bh[(I0) + i] + kr, wr = bw[(I0) + i] + ks;                                                 \
                const bool v = !dead && kvalid && (unsigned)hr < (unsigned)a.H && (unsigned)wr < (unsigned)a.W;           \
                const unsigned voff = v ? (unsigned)(off0[(I0) + i] + tapoff) * (unsigned)ES : OOB;                       \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)dst, 16, voff, 0, 0, 0);                            \
            }                                                                                                             \
        }                                                                                                                 \
    }
#define P3_ADVANCE_K()                                                                                                    \
    {                                                                                                                     \
        ++ktA;                                                                                                            \
        if (!PLAIN) {                                                                                                     \
            kc += BK;                                                                                                     \
            tapoff += BK;                                                                                                 \
            while (kc >= a.C) {                                                                                           \
                kc -= a.C;                                                                                                \
                tapoff += a.ldx - a.C;                                                                                    \
                if (++ks == a.S) { ks = 0; ++kr; tapoff += rowstep - colwrap; }                                           \
            }                                                                                                             \
        }                                                                                                                 \
    }
#define P3_ISSUE_B(PARQ)                                                                                                  \
    {                                                                                                                     \
        _Pragma("unroll") for (int p = 0; p < TN; ++p) {                                                                  \
            unsigned char* dst = smem + A_BYTES + (PARQ) * BBUF + (p * 64 + wave * 8) * ROWB;                              \
            lds_dma16(rsB, (lptr_t)dst, (!dead && ktA < nk) ? woff[p] : OOB, ktA * 128);                                  \
        }                                                                                                                 \
    }

    // ---- deferred output of the tile that has finished
    u32x2_t packed[MT][TN];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) packed[i][j] = u32x2_t{0u, 0u};
    int bq = 0;                                          // 2: this phase issues the burst, 1: the phase after it, 0: no burst in flight
    unsigned yb = OOB;                                   // this lane's byte offset of (tile row frow of its half, first channel)
    unsigned nvalid = 0;                                 // bit nt: the lane's four channels of column tile nt exist
    const int rowstep16 = 16 * a.ldy * ES;
    int m0_fin = 0, n0_fin = 0, mtile_fin = 0;

    // the packed tile leaves in ONE straight-line burst of NST 8-byte stores (static register indices: a store slot chosen by a
    // run-time index compiled to a branch tree that cost more than the stores), issued BEHIND the LDS-DMA pieces of the phase in
    // which the tile is packed: the two waits that follow only cover pieces requested before the burst (their allowed-outstanding
    // counts grow by NST), the third one -- a whole K-tile later -- is the first that has the burst in front of it
#define P3_BURST()                                                                                                        \
    {                                                                                                                     \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                                 \
            _Pragma("unroll") for (int nt = 0; nt < TN; ++nt)                                                             \
                __builtin_amdgcn_raw_buffer_store_b64(packed[mt][nt], rsY, (nvalid >> nt & 1u) ? yb : OOB,                \
                                                      mt * rowstep16 + nt * 32, 0);                                       \
    }
    static_assert(NST <= 40, "vmcnt range");

    f32x4_t acc[MT][TN];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // the finished tile (m0_fin, n0_fin): accumulators -> packed bf16 (+ affine / + statistics), accumulators zeroed
#define P3_PACK()                                                                                                         \
    {                                                                                                                     \
        const int nb = n0_fin + wc * (16 * TN) + 4 * g4;                                                                  \
        float s1[TN][4], s2[TN][4];                                                                                       \
        _Pragma("unroll") for (int nt = 0; nt < TN; ++nt) {                                                               \
            f32x4_t sc = f32x4_t{1.f, 1.f, 1.f, 1.f}, sh = f32x4_t{0.f, 0.f, 0.f, 0.f};                                   \
            if (EPI == 1) {                                                                                               \
                /* (inline asm: an LDS load the compiler can see would be ordered behind ALL pending LDS-DMA) */          \
                const int nn = nb + nt * 16 < a.K ? nb + nt * 16 : 0;                                                     \
                const unsigned sa = (unsigned)(size_t)(lptr_t)(sS + nn);                                                  \
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)"              \
                             : "=&v"(sc), "=&v"(sh) : "v"(sa), "n"(KMAX * 4) : "memory");                                 \
            }                                                                                                             \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) s1[nt][j] = s2[nt][j] = 0.f;                                    \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                           \
                f32x4_t v = acc[mt][nt];                                                                                  \
                if (EPI == 1) {                                                                                           \
                    /* the affine acts on the conv output AS STORED (rounded to bf16), like conv_epilogue_store: the same  \
                       bits whichever kernel the batch size selects */                                                     \
                    const unsigned r01 = pack2bf(v[0], v[1]), r23 = pack2bf(v[2], v[3]);                                  \
                    v[0] = __uint_as_float(r01 << 16); v[1] = __uint_as_float(r01 & 0xffff0000u);                         \
                    v[2] = __uint_as_float(r23 << 16); v[3] = __uint_as_float(r23 & 0xffff0000u);                         \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                       \
                        v[j] = v[j] * sc[j] + sh[j];                                                                      \
                        if (a.ep_relu) v[j] = fmaxf(v[j], 0.f);                                                           \
                    }                                                                                                     \
                }                                                                                                         \
                u32x2_t u;                                                                                                \
                u.x = pack2bf(v[0], v[1]);                                                                                \
                u.y = pack2bf(v[2], v[3]);                                                                                \
                packed[mt][nt] = u;                                                                                       \
                if (EPI == 0) {                                                                                           \
                    const float r0 = __uint_as_float(u.x << 16), r1 = __uint_as_float(u.x & 0xffff0000u);                 \
                    const float r2 = __uint_as_float(u.y << 16), r3 = __uint_as_float(u.y & 0xffff0000u);                 \
                    s1[nt][0] += r0; s2[nt][0] += r0 * r0;                                                                \
                    s1[nt][1] += r1; s2[nt][1] += r1 * r1;                                                                \
                    s1[nt][2] += r2; s2[nt][2] += r2 * r2;                                                                \
                    s1[nt][3] += r3; s2[nt][3] += r3 * r3;                                                                \
                }                                                                                                         \
                acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};                                                                \
            }                                                                                                             \
        }                                                                                                                 \
        if (EPI == 0 && a.part) {                                                                                         \
            /* sums over the 16 pixel lanes of the row; lane frow == 0 of every channel group writes its 4 x TN channels */\
            float* prow = a.part + (size_t)(2 * mtile_fin + grp) * 2 * a.K;                                               \
            _Pragma("unroll") for (int nt = 0; nt < TN; ++nt) {                                                           \
                f32x4_t q1, q2;                                                                                           \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                           \
                    q1[j] = row16_sum(s1[nt][j]);                                                                         \
                    q2[j] = row16_sum(s2[nt][j]);                                                                         \
                }                                                                                                         \
                const int nn = nb + nt * 16;                                                                              \
                if (frow == 0 && nn < a.K) {                                                                              \
                    *reinterpret_cast<f32x4_t*>(prow + nn) = q1;                                                          \
                    *reinterpret_cast<f32x4_t*>(prow + a.K + nn) = q2;                                                    \
                }                                                                                                         \
            }                                                                                                             \
        }                                                                                                                 \
        yb = (unsigned)((m0_fin + grp * HM + frow) * a.ldy + nb) * (unsigned)ES;                                          \
        nvalid = 0;                                                                                                       \
        _Pragma("unroll") for (int nt = 0; nt < TN; ++nt) nvalid |= (nb + nt * 16 < a.K ? 1u : 0u) << nt;                 \
    }

    // ---- prologue: K-tile 0 of the first tile (even slot + filter tile, odd slot), then dropped stores that make the first
    // phases' counted waits see the same queue as every later phase
    P3_SETUP()
    P3_ISSUE_A(0, NA0, 0, 0)
    P3_ISSUE_B(0)
    P3_ISSUE_A(NA0, NA1, E_BYTES, 0)
    P3_ADVANCE_K()

    unsigned faE[2], faO[2], faB[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int ph = ((kk * 4 + fchunk) ^ (frow & 7)) * 16;
        faE[kk] = (unsigned)(size_t)(lptr_t)(smem + (grp * 16 * PM0 + frow) * ROWB + ph);
        faO[kk] = (unsigned)(size_t)(lptr_t)(smem + E_BYTES + (grp * 16 * PM1 + frow) * ROWB + ph);
        faB[kk] = (unsigned)(size_t)(lptr_t)(smem + A_BYTES + (wc * 16 * TN + frow) * ROWB + ph);
    }

    wait_vmcnt<NA1>();                                 // even slot + filter tile of K-tile 0 have landed
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();        // group 1 runs one barrier behind group 0

    bf16x8_t fb[TN][2];
    bf16x8_t fa[PMX][2];
#pragma unroll
    for (int i = 0; i < TN; ++i) fb[i][0] = fb[i][1] = bf16x8_t{};
#pragma unroll
    for (int i = 0; i < PMX; ++i) fa[i][0] = fa[i][1] = bf16x8_t{};

    unsigned par = 0;
    bool fin = false;                                  // a finished tile waits to be packed
    for (;;) {
        for (int kt = 0; kt < nk; ++kt) {
            const bool lastk = kt == nk - 1;
            // ================================================ even phase
            {
                if (kt == 0 && fin) {
                    P3_PACK()
                    fin = false;
                    bq = 2;
                }
                const unsigned bB0 = faB[0] + par * BBUF, bB1 = faB[1] + par * BBUF;
                const unsigned bA0 = faE[0] + par * APAR, bA1 = faE[1] + par * APAR;
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) {
                    if (nt == 0) { P3_DSREAD(fb[0][0], bB0, 0); P3_DSREAD(fb[0][1], bB1, 0); }
                    if (nt == 1) { P3_DSREAD(fb[1][0], bB0, 16 * ROWB); P3_DSREAD(fb[1][1], bB1, 16 * ROWB); }
                    if (nt == 2) { P3_DSREAD(fb[2][0], bB0, 32 * ROWB); P3_DSREAD(fb[2][1], bB1, 32 * ROWB); }
                    if (nt == 3) { P3_DSREAD(fb[3][0], bB0, 48 * ROWB); P3_DSREAD(fb[3][1], bB1, 48 * ROWB); }
                }
#pragma unroll
                for (int ml = 0; ml < PM0; ++ml) {
                    if (ml == 0) { P3_DSREAD(fa[0][0], bA0, 0); P3_DSREAD(fa[0][1], bA1, 0); }
                    if (ml == 1) { P3_DSREAD(fa[1][0], bA0, 16 * ROWB); P3_DSREAD(fa[1][1], bA1, 16 * ROWB); }
                    if (ml == 2) { P3_DSREAD(fa[2][0], bA0, 32 * ROWB); P3_DSREAD(fa[2][1], bA1, 32 * ROWB); }
                    if (ml == 3) { P3_DSREAD(fa[3][0], bA0, 48 * ROWB); P3_DSREAD(fa[3][1], bA1, 48 * ROWB); }
                    if (ml == 4) { P3_DSREAD(fa[4][0], bA0, 64 * ROWB); P3_DSREAD(fa[4][1], bA1, 64 * ROWB); }
                    if (ml == 5) { P3_DSREAD(fa[5][0], bA0, 80 * ROWB); P3_DSREAD(fa[5][1], bA1, 80 * ROWB); }
                }
                if (lastk) {
                    // this tile needs no further pieces: the gather moves on to the block's next tile (its K-tile 0 takes the
                    // ring slots K-tile nk of this tile would have taken)
                    m0_fin = m0; n0_fin = n0; mtile_fin = mtile;
                    const int next = tile + (int)gridDim.x;
                    if (next < ntiles) {
                        tile = next;
                        P3_SETUP()
                    } else {
                        dead = true;
                    }
                }
                P3_ISSUE_A(0, NA0, 0, par ^ 1u)
                P3_ISSUE_B(par ^ 1u)
                if (bq == 2) {
                    P3_BURST()
                    wait_vmcnt<NA0 + TN + NST>();       // the odd slot of this K-tile (requested before everything above)
                } else {
                    wait_vmcnt<NA0 + TN>();
                }
                __builtin_amdgcn_s_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) asm volatile("" : "+v"(fb[nt][0]), "+v"(fb[nt][1]));
#pragma unroll
                for (int ml = 0; ml < PM0; ++ml) asm volatile("" : "+v"(fa[ml][0]), "+v"(fa[ml][1]));
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
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
#pragma unroll
                for (int ml = 0; ml < PM1; ++ml) {
                    if (ml == 0) { P3_DSREAD(fa[0][0], bA0, 0); P3_DSREAD(fa[0][1], bA1, 0); }
                    if (ml == 1) { P3_DSREAD(fa[1][0], bA0, 16 * ROWB); P3_DSREAD(fa[1][1], bA1, 16 * ROWB); }
                    if (ml == 2) { P3_DSREAD(fa[2][0], bA0, 32 * ROWB); P3_DSREAD(fa[2][1], bA1, 32 * ROWB); }
                    if (ml == 3) { P3_DSREAD(fa[3][0], bA0, 48 * ROWB); P3_DSREAD(fa[3][1], bA1, 48 * ROWB); }
                    if (ml == 4) { P3_DSREAD(fa[4][0], bA0, 64 * ROWB); P3_DSREAD(fa[4][1], bA1, 64 * ROWB); }
                    if (ml == 5) { P3_DSREAD(fa[5][0], bA0, 80 * ROWB); P3_DSREAD(fa[5][1], bA1, 80 * ROWB); }
                }
                P3_ISSUE_A(NA0, NA1, E_BYTES, par ^ 1u)
                P3_ADVANCE_K()
                if (bq == 2) {
                    wait_vmcnt<NA1 + NST>();            // the even slot and the filter tile of the next K-tile: requested before the burst
                    bq = 0;
                } else {
                    wait_vmcnt<NA1>();
                }
                __builtin_amdgcn_s_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int ml = 0; ml < PM1; ++ml) asm volatile("" : "+v"(fa[ml][0]), "+v"(fa[ml][1]));
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
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
            par ^= 1u;
        }
        fin = true;
        if (dead) break;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    // ---- the block's last tile: its stores are the only exposed ones
    P3_PACK()
    P3_BURST()
    wait_vmcnt<0>();
#undef P3_SETUP
#undef P3_ISSUE_A
#undef P3_ISSUE_B
#undef P3_ADVANCE_K
#undef P3_BURST
#undef P3_PACK
}

// IFCBK_CONV_PP3: 0 = never, 1 = the measured niche (DEFAULT), 2 = wherever the kernel applies (tests).
// Measured (scripts/conv_pp3_fixedcost.py, scripts/conv_pp3_check.py): on the SAME 256 x 192 tile a block that walks 8
// tiles takes exactly as long as 8 conv_pp2 blocks (171.8 vs 172.3 us at 8 K-tiles per tile, 432 vs 433 at 32): the 11 us a tile
// costs beyond its K-tiles are not launch + prologue + LDS-staged epilogue but the time its 96 KB of output need to leave the CU
// (7-10 B/clk per CU with every CU storing at once = the chip's write bandwidth), and loads issued behind the stores wait for them
// (vmcnt retires in order).  So training never takes it.  The niche is the EVAL forward of a large RUN batch, whose affine epilogue
// has no statistics to reduce: where conv_pp2 would run this very tile (MT 8, TN 3: the 8x8 layers at batch >= 768) it is 1.00-1.10x,
// on the 320 x 192 layers from ~4.5 tiles per CU on 1.01-1.03x; the RUN forward at batch 1024: 50.1 -> 51.5 k img/s.
int pp3_mode() {
    const char* e = getenv("IFCBK_CONV_PP3");
    return e ? atoi(e) : 1;
}

template <int EPI, bool PLAIN>
void launch_pp3(const ConvArgs& a, int ntiles, int grid, unsigned ybytes, hipStream_t st) {
    hipLaunchKernelGGL((conv_pp3<3, 8, 4, EPI, PLAIN>), dim3((unsigned)grid), dim3(512), 0, st, a, ntiles, ybytes);
}

}  // namespace

// Does the persistent kernel serve this GEMM (M pixels, K output channels, reduction Kg)?  `epi`: 0 raw (+ statistics), 1 affine.
// One 256 x 192 tile template; worth it when a CU gets at least two tiles (a single tile per CU leaves only the exposed
// register-direct epilogue: conv_pp2's LDS-staged one is the better of the two).
bool ifcbk_conv_pp3_plan(int dtype, int M, int K, int Kg, int epi) {
    const int mode = pp3_mode();
    if (mode <= 0 || dtype != IFCBK_BF16 || (epi != 0 && epi != 1)) return false;
    if (K % 8 || (epi == 1 && K > 2048)) return false;
    const int nk = cdiv(Kg, 64);
    if (nk < 3) return false;                               // (the burst of a tile must be out of the way before the next tile is packed)
    if (mode >= 2) return true;
    if (epi != 1) return false;                             // training forward / input gradients: never (see pp3_mode)
    int bmt = 0, btn = 0;
    if (!ifcbk_conv_big_plan(dtype, M, K, Kg, &bmt, &btn) || btn != 3) return false;      // only where conv_pp2 would run 192-channel tiles
    const int cus = ifcbk_num_cus();
    const int64_t tiles = (int64_t)cdiv(M, 256) * cdiv(K, 192);
    if ((double)cdiv(K, 192) * 192 / K > 1.2) return false;
    return bmt == 8 ? 2 * tiles >= 3 * cus : tiles >= 4 * cus;      // (measured at batch 768 / 1024: DESIGN 5.9)
}

int ifcbk_conv_pp3_launch(ifcbk_ctx* ctx, void* args, hipStream_t st) {
    ConvArgs& a = *reinterpret_cast<ConvArgs*>(args);
    a.tilesN = cdiv(a.K, 192);
    const int64_t ntiles = (int64_t)cdiv(a.M, 256) * a.tilesN;
    if (ntiles >= (1ll << 30)) IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv_pp3: grid too large");
    const int64_t yb = (int64_t)a.M * a.ldy * 2;
    if (yb >= (1ll << 31)) IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv_pp3: output exceeds the 2 GiB buffer-descriptor window");
    int cus = ifcbk_num_cus();
    if (const char* e = getenv("IFCBK_CONV_PP3_GRID")) { const int f = atoi(e); if (f > 0) cus = f; }      // test hook: few blocks walk many tiles
    const int grid = (int)(ntiles < cus ? ntiles : cus);
    const bool plain = a.R == 1 && a.S == 1 && a.base_h == 0 && a.base_w == 0 && a.ostr_h == 1 && a.ostr_w == 1;
    if (a.ep_scale) { if (plain) launch_pp3<1, true>(a, (int)ntiles, grid, (unsigned)yb, st); else launch_pp3<1, false>(a, (int)ntiles, grid, (unsigned)yb, st); }
    else { if (plain) launch_pp3<0, true>(a, (int)ntiles, grid, (unsigned)yb, st); else launch_pp3<0, false>(a, (int)ntiles, grid, (unsigned)yb, st); }
    IFCBK_LAUNCH_CHECK(ctx, "conv_pp3");
    return 0;
}

void ifcbk_conv_pp3_name(int Kg, bool affine, bool plain, char* name, size_t cap) {
    (void)Kg;
    snprintf(name, cap, "conv_pp3<3, 8, 4, %d, %s>", affine ? 1 : 0, plain ? "true" : "false");
}
