// Implicit-GEMM convolution forward / input-gradient on MFMA (gfx950), bf16 storage, fp32 accumulate.
//
//   C[m][n] = sum_k A[m][k] * B[n][k]      m = output pixel (n,p,q), n = output channel,
//                                          k = (r,s,c) with c fastest  (NHWC activations, KRSC filters)
// A is gathered on the fly (never materialised): each 16-byte chunk = 8 consecutive channels of one
// input pixel.  One kernel serves forward and dgrad: dgrad is the same gather over dy with the flipped,
// transposed filter and "input dilation" (positions not divisible by the stride contribute zero).
//
// Tile: 128 pixels x (32*NT) channels x 64 k per step, 4 waves (2x2), v_mfma_f32_16x16x32_bf16; both operand
// tiles are staged by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip) into a 2-stage ring: the DMA of
// tile k+1 is in flight while tile k is multiplied.
// The filter fragment is the MFMA A operand and the pixel fragment the B operand, so each lane ends with
// 4 consecutive CHANNELS of one pixel (one 8-byte LDS store per tile); the block tile is then written out
// through LDS as whole 16-byte chunks (coalesced rows) and the BatchNorm batch statistics (sum, sum of
// squares of the ROUNDED outputs) are reduced in the same pass -- no extra read of the conv output.
#include "conv_common.h"
#include <stdlib.h>

namespace {

// NT: 16-column tiles per wave in N (block N = 32*NT); WM: waves along M (block M = 64*WM, threads = 128*WM);
// NSTAGE: LDS ring depth (NSTAGE-1 tiles of LDS-DMA in flight across the per-step barrier, counted vmcnt).
// MODE 0: forward / stride-1 dgrad.  MODE 1: dgrad with input dilation (any stride-2 filter; 3/4 of the taps multiply
// zeros).  MODE 2: one parity class of a stride-2 dgrad as a dense stride-1 conv with the class's sub-filter.
template <class T, int NT, int WM, int NSTAGE, int MODE>
__global__ __launch_bounds__(128 * WM) void conv_igemm(ConvArgs a) {
    constexpr bool STRIDED = MODE == 1;
    constexpr int ES = (int)sizeof(T);
    constexpr int CE = 16 / ES;                        // elements per 16-byte chunk
    constexpr int BK = 128 / ES;                       // k elements per tile step (128-byte rows)
    typedef typename Mma<T>::frag_t frag_t;
    constexpr int NW = 2 * WM;
    constexpr int NTHREADS = 64 * NW;
    constexpr int BM = 64 * WM;
    constexpr int BN = 32 * NT;
    constexpr int JB = (NT * 4 + NW - 1) / NW;         // B-tile LDS-DMA instructions per wave per tile
    constexpr int G = 4 + JB;                          // LDS-DMA instructions per wave per tile
    constexpr int D = NSTAGE - 1;                      // prefetch distance (tiles)
    constexpr int LDC = BN + CE;                       // C-tile row stride (elements)
    constexpr int STAGE = (BM + BN) * BK;              // elements per pipeline stage
    constexpr int STAGE_BYTES = NSTAGE * STAGE * ES;
    constexpr int CT_BYTES = BM * LDC * ES + NW * BN * 2 * 4;
    constexpr int MAIN_BYTES = STAGE_BYTES > CT_BYTES ? STAGE_BYTES : CT_BYTES;
    constexpr int DUMMY_BYTES = (JB * NW > NT * 4) ? 1024 : 0;
    __shared__ __attribute__((aligned(16))) unsigned char smem[MAIN_BYTES + DUMMY_BYTES];
    T* sStage = reinterpret_cast<T*>(smem);                       // [NSTAGE][A: BM*BK | B: BN*BK]
    T* sC = reinterpret_cast<T*>(smem);                           // [BM][LDC] (epilogue)
    float* sRed = reinterpret_cast<float*>(smem + BM * LDC * ES); // [NW][2][BN] (epilogue)
    T* sDummy = reinterpret_cast<T*>(smem + MAIN_BYTES);          // sink of the padding LDS-DMA (uniform vmcnt)

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int bid = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int mtile = bid / a.tilesN, ntile = bid - mtile * a.tilesN;
    const int m0 = mtile * BM, n0 = ntile * BN;

    // ---- LDS-DMA roles: one wave-instruction fills 8 tile rows (1 KiB); lane -> (row l>>3, phys chunk l&7).
    // Loads are buffer_load ... lds through two SRDs: the hardware range check turns the out-of-range offset given to
    // padded / strided-out / tail chunks into zeros, so the gather needs no zero page, no 64-bit address math and no
    // branches -- per row and k-step: two adds, two unsigned compares, one add, one select.
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.wbytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int lrow8 = lane >> 3;
    const int csrc = (lane & 7) ^ lrow8;               // logical 16-byte chunk this lane fetches, every row group
    int off0[4], bh[4], bw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int m = m0 + (wave * 4 + j) * 8 + lrow8;
        bool rv = m < a.M;
        int mm = rv ? m : 0;
        int n = (int)fdiv((uint32_t)mm, a.fPQ);
        int rem = mm - n * a.PQ;
        int p = (int)fdiv((uint32_t)rem, a.fQ);
        int q = rem - p * a.Q;
        bh[j] = rv ? p * a.ostr_h + a.base_h : -(1 << 24);       // rows past M never pass the range check
        bw[j] = q * a.ostr_w + a.base_w;
        off0[j] = STRIDED ? n * a.H * a.W * a.ldx : ((n * a.H + bh[j]) * a.W + bw[j]) * a.ldx;
    }
    int kc, kr, ks, tapoff;
    {
        // csrc*CE < BK: a short subtract loop instead of two integer divisions
        kc = csrc * CE;
        kr = 0;
        ks = 0;
        while (kc >= a.C) {
            kc -= a.C;
            if (++ks == a.S) { ks = 0; ++kr; }
        }
        tapoff = (kr * a.W + ks) * a.ldx + kc;
    }
    int wk = MODE == 2 ? ((a.w_rbase + 2 * kr) * a.wSfull + a.w_sbase + 2 * ks) * a.C + kc : 0;   // k index inside the FULL filter row
    unsigned woff[JB];
    bool gvalid[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        int grp = j * NW + wave;
        gvalid[j] = grp < NT * 4;
        int n = n0 + grp * 8 + lrow8;
        woff[j] = (gvalid[j] && n < a.K) ? (MODE == 2 ? (unsigned)(n * a.wKg) : (unsigned)(n * a.Kg + csrc * CE)) * (unsigned)ES : OOB;
    }
    const int nk = (a.Kg + BK - 1) / BK;
    const int hmask = (1 << a.ish) - 1, wmask = (1 << a.isw) - 1;
    const int rowstep = a.W * a.ldx, colwrap = a.S * a.ldx;
    const int wrowskip = MODE == 2 ? (2 * a.wSfull - 2 * a.S) * a.C : 0;     // from the class's last tap column to the first of the next tap row

    // 1x1 filter without padding (most layers of the inception / resnet graphs): the gather is a plain row read --
    // per-lane offsets are constants and the k advance is a scalar (soffset): no per-step VALU address work at all
    const bool plain = (MODE == 0 || MODE == 3 || MODE == 4 || MODE == 5) && a.R == 1 && a.S == 1 && a.base_h == 0 && a.base_w == 0;
    unsigned va[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) va[j] = bh[j] >= 0 ? (unsigned)(off0[j] + csrc * CE) * (unsigned)ES : OOB;
    const bool ktail_ok = (nk - 1) * BK + csrc * CE < a.Kg;      // this lane's chunk of the LAST k step is inside the row

#define ISSUE_TILE(kt, stage)                                                                              \
    {                                                                                                      \
        T* dstA = sStage + (stage) * STAGE;                                                                \
        T* dstB = dstA + BM * BK;                                                                          \
        if (plain) {                                                                                       \
            const bool cut = (kt) == nk - 1 && !ktail_ok;                                                  \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                  \
                lds_dma16(rsA, (lptr_t)(dstA + (wave * 4 + j) * 8 * BK), cut ? OOB : va[j], (kt) * 128);   \
        } else {                                                                                           \
            const bool kvalid = kr < a.R;                                                                  \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                \
                int hr = bh[j] + kr, wr = bw[j] + ks;                                                      \
                unsigned voff;                                                                             \
                if (STRIDED) {                                                                             \
                    bool v = kvalid && hr >= 0 && wr >= 0 && ((hr & hmask) == 0) && ((wr & wmask) == 0);   \
                    int hi = hr >> a.ish, wi = wr >> a.isw;                                                \
                    v = v && hi < a.H && wi < a.W;                                                         \
                    voff = v ? (unsigned)(off0[j] + (hi * a.W + wi) * a.ldx + kc) * (unsigned)ES : OOB;    \
                } else {                                                                                   \
                    bool v = kvalid && (unsigned)hr < (unsigned)a.H && (unsigned)wr < (unsigned)a.W;       \
                    voff = v ? (unsigned)(off0[j] + tapoff) * (unsigned)ES : OOB;                          \
                }                                                                                          \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(dstA + (wave * 4 + j) * 8 * BK), 16, voff, 0, 0, 0); \
            }                                                                                              \
        }                                                                                                  \
        if (MODE == 2) {                                                                                   \
            const bool kv2 = ((kt) * BK + csrc * CE) < a.Kg;                                               \
            _Pragma("unroll") for (int j = 0; j < JB; ++j) {                                               \
                unsigned voff = (kv2 && woff[j] != OOB) ? woff[j] + (unsigned)wk * (unsigned)ES : OOB;     \
                T* dst = gvalid[j] ? dstB + (j * NW + wave) * 8 * BK : sDummy;                             \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)dst, 16, voff, 0, 0, 0);             \
            }                                                                                              \
        } else {                                                                                           \
            /* filter rows are k-contiguous: per-lane row offset + scalar k advance.  Chunks past Kg in the last step   \
               read the head of the next row (or zeros past the buffer): harmless, the pixel operand is zero there */ \
            _Pragma("unroll") for (int j = 0; j < JB; ++j) {                                               \
                T* dst = gvalid[j] ? dstB + (j * NW + wave) * 8 * BK : sDummy;                             \
                lds_dma16(rsB, (lptr_t)dst, woff[j], (kt) * 128);                                          \
            }                                                                                              \
        }                                                                                                  \
        if (!plain) {                                                                                      \
            kc += BK;                                                                                      \
            tapoff += BK;                                                                                  \
            if (MODE == 2) wk += BK;                                                                       \
            while (kc >= a.C) {                                                                            \
                kc -= a.C;                                                                                 \
                tapoff += a.ldx - a.C;                                                                     \
                if (MODE == 2) wk += a.C;                  /* next tap column: +2 taps, -C channels */     \
                if (++ks == a.S) { ks = 0; ++kr; tapoff += rowstep - colwrap; if (MODE == 2) wk += wrowskip; } \
            }                                                                                              \
        }                                                                                                  \
    }

    f32x4_t acc[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // prologue: D tiles in flight
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < nk) ISSUE_TILE(d, d)

    const int frow = lane & 15, fchunk = lane >> 4;
    // fragment addresses: the swizzle term depends only on (frow & 7, fchunk, kk); m/n tiles are immediates
    const T* fA[2];
    const T* fB[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int ph = ((kk * 4 + fchunk) ^ (frow & 7)) * CE;
        fA[kk] = sStage + (wm * 64 + frow) * BK + ph;
        fB[kk] = sStage + BM * BK + (wn * (NT * 16) + frow) * BK + ph;
    }
    int stage = 0;                // stage holding tile kt
    int istage = D % NSTAGE;      // stage the next issue goes to
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once all but the tiles issued after it are done (counted, never a full drain mid-loop)
        const int newer = nk - 1 - kt < D - 1 ? nk - 1 - kt : D - 1;
        if (D >= 3 && newer >= 2) wait_vmcnt<2 * G>();
        else if (D >= 2 && newer >= 1) wait_vmcnt<G>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();   // every wave's share of tile kt is in LDS; stage (kt-1) is no longer read
        if (kt + D < nk) ISSUE_TILE(kt + D, istage)
        const int soff = stage * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            frag_t fa[4], fb[NT];
            const T* pa = fA[kk] + soff;
            const T* pb = fB[kk] + soff;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) fa[mt] = *reinterpret_cast<const frag_t*>(pa + mt * 16 * BK);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) fb[nt] = *reinterpret_cast<const frag_t*>(pb + nt * 16 * BK);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    Mma<T>::run(fb[nt], fa[mt], acc[nt][mt]);
        }
        stage = stage + 1 == NSTAGE ? 0 : stage + 1;
        istage = istage + 1 == NSTAGE ? 0 : istage + 1;
    }
#undef ISSUE_TILE
    __syncthreads();            // all stages drained and consumed: the epilogue reuses the ring as the C tile

    // ---- epilogue: acc -> bf16 C tile in LDS (lane: 4 consecutive channels of one pixel)
    {
        const int g = lane >> 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                int m = wm * 64 + mt * 16 + frow;
                int n = wn * (NT * 16) + nt * 16 + 4 * g;
                Mma<T>::pack4(sC + m * LDC + n, acc[nt][mt]);
            }
    }
    __syncthreads();
    conv_epilogue_store<T, BM, BN, NTHREADS, MODE>(a, sC, sRed, t, lane, wave, m0, n0, mtile);
}

// ---------------------------------------------------------------- persistent, warp-specialised variant
// One 512-thread block per CU walks over tiles (tile = block + r * grid).  Waves 4-7 ("producers") only run the LDS-DMA
// gather of both operand tiles, two k-steps ahead in a 3-stage ring, and simply keep going into the NEXT tile's first
// k-steps while waves 0-3 ("consumers", the 2x2 MFMA waves of conv_igemm) write the finished tile out -- no prologue
// latency after the first tile, no address arithmetic and no DMA issue stalls on the MFMA waves.  One s_barrier per
// k-step is the whole protocol: producers wait (counted vmcnt) until their pieces of step s have landed, everybody
// meets, producers then refill the stage the consumers finished one step earlier.  Both roles execute exactly
// tiles * nk barriers.  The consumers' epilogue goes from registers straight to memory (no LDS: the ring belongs to the
// producers, and gfx950 has no partial-workgroup barrier for a C tile): a lane holds 4 consecutive channels of a pixel
// per 16x16 tile (8-byte stores, 192-byte runs per pixel that L2 merges), the BatchNorm sums are reduced over the 16
// pixel lanes with cross-lane adds and every consumer wave writes its own partial row (2 rows per 128-pixel tile).
// Serves: forward / stride-1 first-writer input gradients in bf16 without residual (MODE 0 of conv_igemm).
template <int NT>
__global__ __launch_bounds__(512) void conv_ws(ConvArgs a, int total_tiles) {
    typedef bf16_t T;
    constexpr int ES = 2, CE = 8, BK = 64, BM = 128, BN = 32 * NT, NSTAGE = 3;
    constexpr int STAGE = (BM + BN) * BK;              // elements per ring stage
    constexpr int G = 4 + NT;                          // LDS-DMA instructions per producer wave per k-step
    __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * STAGE * ES];
    T* sStage = reinterpret_cast<T*>(smem);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int b0 = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int ngrid = (int)gridDim.x;
    const int ntl = (total_tiles - b0 + ngrid - 1) / ngrid;          // tiles of this block (grid <= total_tiles: >= 1)
    const int nk = (a.Kg + BK - 1) / BK;
    const int total_steps = ntl * nk;

    if (wave >= 4) {
        // ================================================================ producers
        const int p = wave - 4;
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.wbytes, 0x00020000);
        constexpr unsigned OOB = 0x80000000u;
        const int lrow8 = lane >> 3;
        const int csrc = (lane & 7) ^ lrow8;
        const bool plain = a.R == 1 && a.S == 1 && a.base_h == 0 && a.base_w == 0;
        const bool ktail_ok = (nk - 1) * BK + csrc * CE < a.Kg;
        const int rowstep = a.W * a.ldx, colwrap = a.S * a.ldx;
        // per-tile gather state
        int off0[4], bh[4], bw[4];
        unsigned va[4], woff[NT];
        int kc = 0, kr = 0, ks = 0, tapoff = 0, kt_issue = 0, r_issue = 0;
#define WS_SETUP_TILE(tile)                                                                                   \
        {                                                                                                     \
            const int mtile_ = (tile) / a.tilesN, ntile_ = (tile) - mtile_ * a.tilesN;                        \
            const int m0_ = mtile_ * BM, n0_ = ntile_ * BN;                                                   \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                   \
                int m = m0_ + (p * 4 + j) * 8 + lrow8;                                                        \
                bool rv = m < a.M;                                                                            \
                int mm = rv ? m : 0;                                                                          \
                int n = (int)fdiv((uint32_t)mm, a.fPQ);                                                       \
                int rem = mm - n * a.PQ;                                                                      \
                int pp = (int)fdiv((uint32_t)rem, a.fQ);                                                      \
                int q = rem - pp * a.Q;                                                                       \
                bh[j] = rv ? pp * a.ostr_h + a.base_h : -(1 << 24);                                           \
                bw[j] = q * a.ostr_w + a.base_w;                                                              \
                off0[j] = ((n * a.H + bh[j]) * a.W + bw[j]) * a.ldx;                                          \
                va[j] = bh[j] >= 0 ? (unsigned)(off0[j] + csrc * CE) * (unsigned)ES : OOB;                    \
            }                                                                                                 \
            _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                  \
                int n = n0_ + (j * 4 + p) * 8 + lrow8;                                                        \
                woff[j] = n < a.K ? (unsigned)(n * a.Kg + csrc * CE) * (unsigned)ES : OOB;                    \
            }                                                                                                 \
            kc = csrc * CE; kr = 0; ks = 0;                                                                   \
            while (kc >= a.C) { kc -= a.C; if (++ks == a.S) { ks = 0; ++kr; } }                               \
            tapoff = (kr * a.W + ks) * a.ldx + kc;                                                            \
        }
        // issue the next k-step in program order (tile r_issue, step kt_issue) into ring stage `stage`: exactly G instructions
#define WS_ISSUE(stage)                                                                                       \
        {                                                                                                     \
            if (kt_issue == 0) WS_SETUP_TILE(b0 + r_issue * ngrid)                                            \
            T* dstA = sStage + (stage) * STAGE;                                                               \
            T* dstB = dstA + BM * BK;                                                                         \
            if (plain) {                                                                                      \
                const bool cut = kt_issue == nk - 1 && !ktail_ok;                                             \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                 \
                    lds_dma16(rsA, (lptr_t)(dstA + (p * 4 + j) * 8 * BK), cut ? OOB : va[j], kt_issue * 128); \
            } else {                                                                                          \
                const bool kvalid = kr < a.R;                                                                 \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
                    int hr = bh[j] + kr, wr = bw[j] + ks;                                                     \
                    bool v = kvalid && (unsigned)hr < (unsigned)a.H && (unsigned)wr < (unsigned)a.W;          \
                    unsigned voff = v ? (unsigned)(off0[j] + tapoff) * (unsigned)ES : OOB;                    \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(dstA + (p * 4 + j) * 8 * BK), 16, voff, 0, 0, 0); \
                }                                                                                             \
                kc += BK;                                                                                     \
                tapoff += BK;                                                                                 \
                while (kc >= a.C) {                                                                           \
                    kc -= a.C;                                                                                \
                    tapoff += a.ldx - a.C;                                                                    \
                    if (++ks == a.S) { ks = 0; ++kr; tapoff += rowstep - colwrap; }                           \
                }                                                                                             \
            }                                                                                                 \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                    \
                lds_dma16(rsB, (lptr_t)(dstB + (j * 4 + p) * 8 * BK), woff[j], kt_issue * 128);               \
            if (++kt_issue == nk) { kt_issue = 0; ++r_issue; }                                                \
        }
        if (total_steps > 0) WS_ISSUE(0)
        if (total_steps > 1) WS_ISSUE(1)
        int istage = 2;
        for (int s = 0; s < total_steps; ++s) {
            if (s + 1 < total_steps) wait_vmcnt<G>();       // step s has landed; step s+1 may still be in flight
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                   // consumers are done with step s-1: its stage is free
            if (s + 2 < total_steps) WS_ISSUE(istage)
            istage = istage == 2 ? 0 : istage + 1;
        }
#undef WS_ISSUE
#undef WS_SETUP_TILE
        return;
    }

    // ==================================================================== consumers
    typedef bf16x8_t frag_t;
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fchunk = lane >> 4;
    const T* fA[2];
    const T* fB[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int ph = ((kk * 4 + fchunk) ^ (frow & 7)) * CE;
        fA[kk] = sStage + (wm * 64 + frow) * BK + ph;
        fB[kk] = sStage + BM * BK + (wn * (NT * 16) + frow) * BK + ph;
    }
    int stage = 0;
    for (int r = 0; r < ntl; ++r) {
        const int tile = b0 + r * ngrid;
        const int mtile = tile / a.tilesN, ntile = tile - mtile * a.tilesN;
        const int m0 = mtile * BM, n0 = ntile * BN;
        f32x4_t acc[NT][4];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; ++kt) {
            __builtin_amdgcn_s_barrier();                   // every producer's share of this step is in LDS
            const int soff = stage * STAGE;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                frag_t fa[4], fb[NT];
                const T* pa = fA[kk] + soff;
                const T* pb = fB[kk] + soff;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) fa[mt] = *reinterpret_cast<const frag_t*>(pa + mt * 16 * BK);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) fb[nt] = *reinterpret_cast<const frag_t*>(pb + nt * 16 * BK);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        Mma<T>::run(fb[nt], fa[mt], acc[nt][mt]);
            }
            stage = stage == NSTAGE - 1 ? 0 : stage + 1;
        }
        // ---- epilogue, registers -> memory
        const int g = lane >> 4;
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const int ncol = n0 + wn * (NT * 16) + 4 * g;                 // + nt*16: this lane's 4 channels of tile nt
        bool mv[4];
        T* row[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = m0 + wm * 64 + mt * 16 + frow;
            mv[mt] = m < a.M;
            row[mt] = (T*)a.y + (size_t)(mv[mt] ? m : 0) * a.ldy + ncol;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const bool nv = ncol + nt * 16 < a.K;                     // K is a multiple of 8: the 4 channels are valid together
            float sc[4], sh[4];
            if (a.ep_scale) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    sc[j] = nv ? a.ep_scale[ncol + nt * 16 + j] : 0.f;
                    sh[j] = nv ? a.ep_shift[ncol + nt * 16 + j] : 0.f;
                }
            }
            f32x2_t s1[2] = {f32x2_t{0.f, 0.f}, f32x2_t{0.f, 0.f}}, s2[2] = {f32x2_t{0.f, 0.f}, f32x2_t{0.f, 0.f}};
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                f32x4_t v = acc[nt][mt];
                if (a.ep_scale) {
                    // the affine acts on the conv output AS STORED in training (rounded to bf16), like conv_igemm's epilogue
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = bf2f(f2bf(v[j])) * sc[j] + sh[j];
                        if (a.ep_relu) v[j] = fmaxf(v[j], 0.f);
                    }
                }
                uint2 u;
                u.x = pack2bf(v[0], v[1]);
                u.y = pack2bf(v[2], v[3]);
                if (mv[mt] && nv) *reinterpret_cast<uint2*>(row[mt] + nt * 16) = u;
                // statistics of the ROUNDED values (what bn_apply reads back); rows past M / columns past K are zeros
                const f32x2_t lo = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u)};
                const f32x2_t hi = {__uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
                s1[0] += lo; s1[1] += hi;
                s2[0] += lo * lo; s2[1] += hi * hi;
            }
            if (a.part) {
#pragma unroll
                for (int off = 1; off < 16; off <<= 1)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            s1[h][j] += __shfl_xor(s1[h][j], off);
                            s2[h][j] += __shfl_xor(s2[h][j], off);
                        }
                if (frow == 0 && nv) {
                    float* prow = a.part + ((size_t)(mtile * 2 + wm) * 2) * a.K + ncol + nt * 16;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        prow[j] = s1[j >> 1][j & 1];
                        prow[a.K + j] = s2[j >> 1][j & 1];
                    }
                }
            }
        }
    }
}

// N-tile choice.  Per-block time ~ (BN + 48) (measured round 1: the 48 stands for the pixel tile's load + the fixed parts),
// so without other constraints the widest tile that wastes no columns wins.  IFCBK_CONV_MQ=1 also counts ROUNDS: the chip
// holds 512 blocks (2 per CU); a 578-block grid (every 17x17 layer with K <= 192 at batch 256) runs a second, nearly empty
// round, and narrower tiles (more, shorter blocks) can finish sooner.
int pick_nt(int K, int maxnt, int M, int bm) {
    static int force = -1, mq = -1;
    if (force < 0) { const char* e = getenv("IFCBK_CONV_NT"); force = e ? atoi(e) : 0; }
    if (mq < 0) { const char* e = getenv("IFCBK_CONV_MQ"); mq = e ? atoi(e) : 0; }
    if (force > 0 && force <= maxnt) return force;
    int best = 1;
    long bestc = -1;
    const long tilesM = M > 0 && mq ? cdiv(M, bm) : 0;
    for (int nt = 1; nt <= maxnt; ++nt) {
        int bn = 32 * nt;
        long c = (long)cdiv(K, bn) * (bn + 48);
        if (tilesM) c = cdiv(tilesM * cdiv(K, bn), 512) * (bn + 48);
        if (bestc < 0 || c < bestc || (c == bestc && nt > best)) { bestc = c; best = nt; }
    }
    return best;
}

// (the forward gather takes any stride -- alexnet's first conv is 11x11 / stride 4; the input gradient's dilated gather and its
// parity-class split are written for strides 1 and 2)
int check_desc(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, bool dgrad = false) {
    if (!d) IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv: null desc");
    if (d->dtype != IFCBK_BF16 && d->dtype != IFCBK_F32) IFCBK_FAIL(ctx, IFCBK_EUNSUPPORTED, "conv: dtype must be bf16 or f32");
    const int ce = dtype_chunk(d->dtype), es = dtype_esize(d->dtype);
    if (d->C % ce || d->K % ce || d->ldx % ce || d->ldy % ce || d->C <= 0 || d->K <= 0)
        IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv: C=%d K=%d ldx=%d ldy=%d must be positive multiples of %d", d->C, d->K, d->ldx, d->ldy, ce);
    if (d->stride_h < 1 || d->stride_w < 1 || (dgrad && (d->stride_h > 2 || d->stride_w > 2)))
        IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv: stride must be >= 1 (1 or 2 for the input gradient)");
    int P = (d->H + 2 * d->pad_h - d->R) / d->stride_h + 1, Q = (d->W + 2 * d->pad_w - d->S) / d->stride_w + 1;
    if (P != d->P || Q != d->Q) IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv: P,Q=%d,%d inconsistent (expect %d,%d)", d->P, d->Q, P, Q);
    if ((int64_t)d->N * d->P * d->Q * d->ldy * es >= (1ll << 31) || (int64_t)d->N * d->H * d->W * d->ldx * es >= (1ll << 31))
        IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv: a tensor exceeds the 2 GiB buffer-descriptor window");
    return 0;
}

// block-M choice: 128-pixel tiles (4 waves, 2-stage ring, 2 blocks/CU) beat 256-pixel tiles (8 waves, 3-stage ring,
// 1 block/CU) on every inception layer measured in round 1; IFCBK_CONV_WM=4 forces the large tile for experiments
int pick_wm(int M, int K) {
    static int force = -1;
    if (force < 0) { const char* e = getenv("IFCBK_CONV_WM"); force = e ? atoi(e) : 0; }
    (void)M; (void)K;
    return force == 4 ? 4 : 2;
}

template <class T, int NT, int WM, int NSTAGE>
void launch(const ConvArgs& a, hipStream_t st) {
    int tilesM = cdiv(a.M, 64 * WM);
    dim3 grid((unsigned)(tilesM * a.tilesN)), block(128 * WM);
    if (a.seg_n) hipLaunchKernelGGL((conv_igemm<T, NT, WM, NSTAGE, 4>), grid, block, 0, st, a);
    else if (a.bs_tab) hipLaunchKernelGGL((conv_igemm<T, NT, WM, NSTAGE, 5>), grid, block, 0, st, a);
    else if (a.bs_raw) hipLaunchKernelGGL((conv_igemm<T, NT, WM, NSTAGE, 3>), grid, block, 0, st, a);
    else if (a.wKg) hipLaunchKernelGGL((conv_igemm<T, NT, WM, NSTAGE, 2>), grid, block, 0, st, a);
    else if (a.ish | a.isw) hipLaunchKernelGGL((conv_igemm<T, NT, WM, NSTAGE, 1>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((conv_igemm<T, NT, WM, NSTAGE, 0>), grid, block, 0, st, a);
}

// persistent warp-specialised kernel.  Measured per layer (batch 256, one lane): +25-40 % where a CU gets at most two
// tiles (every 8x8 layer: 705 -> 987 TF/s on Mixed_7b/7c 3x3, 500 -> 630 on the 1x3 / 3x1), neutral on the 578-tile
// 17x17 layers, 10-25 % SLOWER where a CU walks through many short tiles (35x35 layers, the fused 17x17 1x1s): there
// the consumers' epilogue is exposed, while conv_igemm's partner block covers it.  Hence: grids of at most
// IFCBK_CONV_WS_TILES tiles (default 2 per CU) with at least IFCBK_CONV_WS k-steps (default 4; 0 = never).
int ws_min_steps() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("IFCBK_CONV_WS"); v = e ? atoi(e) : 4; }
    return v;
}
int num_cus();
int ws_max_tiles() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("IFCBK_CONV_WS_TILES"); v = e ? atoi(e) : 2 * num_cus(); }
    return v;
}
// M, K, Kg of the GEMM view (pixels, output channels, reduction length)
bool ws_shape_ok(int dtype, int M, int K, int Kg) {
    const int ms = ws_min_steps();
    if (ms <= 0 || dtype != IFCBK_BF16 || pick_wm(M, K) != 2 || cdiv(Kg, 64) < ms) return false;
    const int nt = pick_nt(K, 6, M, 128);
    return (int64_t)cdiv(M, 128) * cdiv(K, 32 * nt) <= ws_max_tiles();
}
int num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <int NT>
void launch_ws(const ConvArgs& a, hipStream_t st) {
    const int total = cdiv(a.M, 128) * a.tilesN;
    const int grid = total < num_cus() ? total : num_cus();
    hipLaunchKernelGGL((conv_ws<NT>), dim3((unsigned)grid), dim3(512), 0, st, a, total);
}

// the wide-tile kernel serves plain gathers (forward of any stride, stride-1 input gradients), with every epilogue variant
bool big_ok(const ConvArgs& a) { return !a.wKg && !(a.ish | a.isw); }
bool pp3_ok(const ConvArgs& a) {
    return big_ok(a) && !a.seg_n && !a.bs_raw && !a.bs_tab && !a.accumulate && !a.ep_res && (a.ep_scale == nullptr || a.part == nullptr);
}

int run(ifcbk_ctx* ctx, ConvArgs& a, int dtype, hipStream_t st) {
    const bool f32 = dtype == IFCBK_F32;
    if (!a.wKg && !(a.ish | a.isw) && a.ostr_h == 1 && a.ostr_w == 1 && !a.seg_n && !a.bs_tab && !a.accumulate && !a.ep_res && a.PQ > 0) {
        // stride-1 3x3 / 5x5 layers over 48..96 channels: the flat-image kernel (conv_flat.hip)
        const int N = a.M / a.PQ;
        if (ifcbk_conv_flat_rows(dtype, N, a.H, a.W, a.C, a.K, a.R, a.S, -a.base_h, -a.base_w, a.P, a.Q) > 0)
            return ifcbk_conv_flat_launch(ctx, &a, N, st);
    }
    // stride-1 multi-tap layers of the 17x17 class: the pixel-slab kernel (conv_slab.hip); every epilogue but the table / segment forms.
    // Its reduction order differs from the other kernels' (same products, fp32 sums in another order), so the plan looks at the layer's
    // shape only: whatever the batch, such a layer always takes this kernel, and an eval batch equals its parts bit for bit
    if (big_ok(a) && a.ostr_h == 1 && a.ostr_w == 1 && !a.seg_n && !a.bs_tab && a.PQ > 0) {
        const int N = a.M / a.PQ;
        if (ifcbk_conv_slab_plan(dtype, N, a.H, a.W, a.C, a.K, a.R, a.S, -a.base_h, -a.base_w, a.P, a.Q) > 0)
            return ifcbk_conv_slab_launch(ctx, &a, N, st);
    }
    // grids of several tiles per CU whose epilogue is a raw store (+ statistics) or the eval affine: the persistent kernel
    if (pp3_ok(a) && ifcbk_conv_pp3_plan(dtype, a.M, a.K, a.Kg, a.ep_scale ? 1 : 0)) return ifcbk_conv_pp3_launch(ctx, &a, st);
    {
        int bmt = 0, btn = 0;
        if (big_ok(a) && ifcbk_conv_big_plan(dtype, a.M, a.K, a.Kg, &bmt, &btn)) return ifcbk_conv_big_launch(ctx, &a, bmt, btn, st);
    }
    int wm = f32 ? 2 : pick_wm(a.M, a.K);
    int nt = pick_nt(a.K, f32 ? 4 : (wm == 4 ? 5 : 6), a.M, 64 * wm);
    a.tilesN = cdiv(a.K, 32 * nt);
    if ((int64_t)cdiv(a.M, 64 * wm) * a.tilesN >= (1ll << 31)) IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv: grid too large");
    if (ws_shape_ok(dtype, a.M, a.K, a.Kg) && !a.seg_n && !a.bs_raw && !a.wKg && !(a.ish | a.isw) && !a.accumulate && !a.ep_res) {
        switch (nt) {
            case 1: launch_ws<1>(a, st); break;
            case 2: launch_ws<2>(a, st); break;
            case 3: launch_ws<3>(a, st); break;
            case 4: launch_ws<4>(a, st); break;
            case 5: launch_ws<5>(a, st); break;
            default: launch_ws<6>(a, st); break;
        }
        IFCBK_LAUNCH_CHECK(ctx, "conv_ws");
        return 0;
    }
    if (f32) {
        switch (nt) {
            case 1: launch<float, 1, 2, 2>(a, st); break;
            case 2: launch<float, 2, 2, 2>(a, st); break;
            case 3: launch<float, 3, 2, 2>(a, st); break;
            default: launch<float, 4, 2, 2>(a, st); break;
        }
    } else if (wm == 4) {
        switch (nt) {
            case 1: launch<bf16_t, 1, 4, 3>(a, st); break;
            case 2: launch<bf16_t, 2, 4, 3>(a, st); break;
            case 3: launch<bf16_t, 3, 4, 3>(a, st); break;
            case 4: launch<bf16_t, 4, 4, 3>(a, st); break;
            default: launch<bf16_t, 5, 4, 3>(a, st); break;
        }
    } else {
        switch (nt) {
            case 1: launch<bf16_t, 1, 2, 2>(a, st); break;
            case 2: launch<bf16_t, 2, 2, 2>(a, st); break;
            case 3: launch<bf16_t, 3, 2, 2>(a, st); break;
            case 4: launch<bf16_t, 4, 2, 2>(a, st); break;
            case 5: launch<bf16_t, 5, 2, 2>(a, st); break;
            default: launch<bf16_t, 6, 2, 2>(a, st); break;
        }
    }
    IFCBK_LAUNCH_CHECK(ctx, "conv_igemm");
    return 0;
}

}  // namespace

int ifcbk_num_cus() { return num_cus(); }

int ifcbk_conv_fwd_nt(int K, int M) { return pick_nt(K, pick_wm(M, K) == 4 ? 5 : 6, M, 64 * pick_wm(M, K)); }

int ifcbk_conv_fwd_wm(int M, int K) { return pick_wm(M, K); }

bool ifcbk_conv_ws_shape(int dtype, int M, int K, int Kg) { return ws_shape_ok(dtype, M, K, Kg); }

static bool fwd_rows(const ifcbk_conv_desc* d) {
    return ifcbk_conv_rows_ok(d->dtype, d->C, d->K, d->R, d->S, d->stride_h, d->stride_w, d->pad_h, d->pad_w, d->Q);
}

extern "C" int ifcbk_conv2d_fwd_mblocks(const ifcbk_conv_desc* d) {
    if (fwd_rows(d)) return ifcbk_conv_rows_blocks(d->N, d->P);
    int M = d->N * d->P * d->Q;
    if (d->stride_h == 1 && d->stride_w == 1)
        if (int fs = ifcbk_conv_flat_rows(d->dtype, d->N, d->H, d->W, d->C, d->K, d->R, d->S, d->pad_h, d->pad_w, d->P, d->Q)) return fs;
    if (d->stride_h == 1 && d->stride_w == 1)
        if (int smt = ifcbk_conv_slab_plan(d->dtype, d->N, d->H, d->W, d->C, d->K, d->R, d->S, d->pad_h, d->pad_w, d->P, d->Q)) return cdiv(M, 32 * smt);
    if (ifcbk_conv_pp3_plan(d->dtype, M, d->K, d->R * d->S * d->C, 0)) return 2 * cdiv(M, 256);      // conv_pp3: one partial row per pixel half
    {
        int bmt = 0, btn = 0;
        if (ifcbk_conv_big_plan(d->dtype, M, d->K, d->R * d->S * d->C, &bmt, &btn)) return cdiv(M, 32 * bmt);
    }
    if (ws_shape_ok(d->dtype, M, d->K, d->R * d->S * d->C)) return 2 * cdiv(M, 128);      // conv_ws: one partial row per consumer wave row
    return cdiv(M, 64 * pick_wm(M, d->K));
}

struct FwdSegs { int n, end[4], ld[4], aff[4]; void* y[4]; };
static int conv_fwd_impl(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const void* x, const void* w, void* y, float* bn_part,
                         const float* scale, const float* shift, const void* residual, int ldr, int relu, void* stream,
                         const FwdSegs* seg = nullptr);

extern "C" int ifcbk_conv2d_fwd_affine_segments(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const void* x, const void* w, int nseg,
                                                void* const* ys, const int32_t* ldys, const int32_t* ksegs, const int32_t* affine,
                                                const float* scale, const float* shift, void* stream) {
    if (!d || nseg < 1 || nseg > 4 || !ys || !ldys || !ksegs || !affine || !scale || !shift)
        IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv2d_fwd_affine_segments: bad args");
    FwdSegs sg;
    sg.n = nseg;
    int end = 0;
    const int ce = dtype_chunk(d->dtype);
    for (int q = 0; q < 4; ++q) {
        const bool live = q < nseg;
        if (live) {
            if (!ys[q] || ksegs[q] <= 0 || ksegs[q] % ce || ldys[q] % ce)
                IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv2d_fwd_affine_segments: segment %d: null / size not a multiple of %d", q, ce);
            end += ksegs[q];
        }
        sg.end[q] = end; sg.ld[q] = live ? ldys[q] : 0; sg.aff[q] = live ? affine[q] : 0; sg.y[q] = live ? ys[q] : nullptr;
    }
    if (end != d->K) IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv2d_fwd_affine_segments: segment sizes sum to %d, K=%d", end, d->K);
    return conv_fwd_impl(ctx, d, x, w, ys[0], nullptr, scale, shift, nullptr, 0, 1, stream, &sg);
}

extern "C" int ifcbk_conv2d_fwd(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const void* x, const void* w, void* y,
                                float* bn_part, void* stream) {
    return conv_fwd_impl(ctx, d, x, w, y, bn_part, nullptr, nullptr, nullptr, 0, 0, stream);
}

extern "C" int ifcbk_conv2d_fwd_affine(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const void* x, const void* w, void* y,
                                       const float* scale, const float* shift, const void* residual, int ldr, int relu,
                                       void* stream) {
    if (!scale || !shift) IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv2d_fwd_affine: scale/shift required");
    return conv_fwd_impl(ctx, d, x, w, y, nullptr, scale, shift, residual, ldr, relu, stream);
}

// eval: conv + folded BatchNorm affine (+ReLU) + the 3x3 / stride-2 / unpadded max pool that is the activation's only consumer, in one
// pass (the row-streaming kernel keeps the pooled maxima in registers): 1 where the descriptor is served
extern "C" int ifcbk_conv2d_fwd_affine_maxpool_ok(const ifcbk_conv_desc* d) {
    return d && ifcbk_conv_rows_pool_ok(d->dtype, d->C, d->K, d->R, d->S, d->stride_h, d->stride_w, d->pad_h, d->pad_w, d->P, d->Q) ? 1 : 0;
}

extern "C" int ifcbk_conv2d_fwd_affine_maxpool(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const void* x, const void* w, void* y_pooled,
                                               int ldp, const float* scale, const float* shift, int relu, void* stream) {
    if (!ifcbk_conv2d_fwd_affine_maxpool_ok(d)) IFCBK_FAIL(ctx, IFCBK_EUNSUPPORTED, "conv2d_fwd_affine_maxpool: not served by the row-streaming kernel");
    if (!x || !w || !y_pooled || !scale || !shift || ldp < d->K || ldp % 8) IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv2d_fwd_affine_maxpool: bad operand");
    {
        // (the un-pooled activation is never stored: only its shape must be consistent, not its size inside the descriptor window)
        ifcbk_conv_desc one = *d;
        one.N = 1;
        if (int e = check_desc(ctx, &one)) return e;
    }
    // image groups inside the 2 GiB descriptor window of the input
    const int64_t per = (int64_t)d->H * d->W * d->ldx * 2;
    const int64_t G = per > 0 ? ((1ll << 31) - 1) / per : d->N;
    if (G < 1) IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv2d_fwd_affine_maxpool: one image exceeds the descriptor window");
    const int Pp = (d->P - 3) / 2 + 1, Qp = (d->Q - 3) / 2 + 1;
    for (int64_t g0 = 0; g0 < d->N; g0 += G) {
        const int n = (int)(d->N - g0 < G ? d->N - g0 : G);
        const int e = ifcbk_conv_rows_pool_launch(ctx, n, d->H, d->W, d->ldx, d->P, d->Q, d->pad_h, d->pad_w, (const char*)x + g0 * per, w,
                                                  (char*)y_pooled + g0 * Pp * Qp * ldp * 2, ldp, scale, shift, relu, (hipStream_t)stream);
        if (e) return e;
    }
    return 0;
}

static int conv_fwd_impl(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const void* x, const void* w, void* y, float* bn_part,
                         const float* scale, const float* shift, const void* residual, int ldr, int relu, void* stream,
                         const FwdSegs* seg) {
    if (d && (d->dtype == IFCBK_BF16 || d->dtype == IFCBK_F32) && d->N > 1) {
        // The kernels address a tensor through a 32-bit buffer descriptor (2 GiB).  A forward WITHOUT batch statistics (eval /
        // affine epilogues: every image on its own) whose tensors exceed the window runs as several launches over image groups
        // -- same bits as one launch; a training forward (bn_part) cannot be cut: its statistics are per launch (check_desc fails)
        const int64_t es = dtype_esize(d->dtype);
        int64_t per = (int64_t)d->H * d->W * d->ldx * es;
        const int64_t ypix = (int64_t)d->P * d->Q;
        if (seg) { for (int q = 0; q < seg->n; ++q) per = per > ypix * seg->ld[q] * es ? per : ypix * seg->ld[q] * es; }
        else per = per > ypix * d->ldy * es ? per : ypix * d->ldy * es;
        if (residual) per = per > ypix * ldr * es ? per : ypix * ldr * es;
        const int64_t G = per > 0 ? ((1ll << 31) - 1) / per : d->N;
        if (d->N > G && G >= 1 && !bn_part) {
            for (int64_t g0 = 0; g0 < d->N; g0 += G) {
                ifcbk_conv_desc dd = *d;
                dd.N = (int)(d->N - g0 < G ? d->N - g0 : G);
                FwdSegs sg;
                if (seg) {
                    sg = *seg;
                    for (int q = 0; q < seg->n; ++q) sg.y[q] = (char*)seg->y[q] + g0 * ypix * seg->ld[q] * es;
                }
                const int e = conv_fwd_impl(ctx, &dd, (const char*)x + g0 * d->H * d->W * d->ldx * es, w,
                                            seg ? sg.y[0] : (y ? (char*)y + g0 * ypix * d->ldy * es : nullptr), nullptr, scale, shift,
                                            residual ? (const char*)residual + g0 * ypix * ldr * es : nullptr, ldr, relu, stream,
                                            seg ? &sg : nullptr);
                if (e) return e;
            }
            return 0;
        }
    }
    if (int e = check_desc(ctx, d)) return e;
    if (fwd_rows(d) && !residual && !seg)
        return ifcbk_conv_rows_launch(ctx, d->C, d->K, d->N, d->H, d->W, d->ldx, d->P, d->Q, d->ldy, d->pad_h, d->pad_w, x, w, y,
                                      bn_part, scale, shift, relu, (hipStream_t)stream);
    ConvArgs a;
    a.dbg = 0; a.tr = 0; a.fP = make_fastdiv(1);
    a.ep_scale = scale; a.ep_shift = shift; a.ep_res = residual; a.ep_ldr = ldr; a.ep_relu = relu;
    a.x = x; a.w = w; a.y = y; a.part = bn_part;
    a.bs_raw = nullptr; a.bs_mean = a.bs_invstd = a.bs_scale = a.bs_shift = nullptr; a.bs_ld = 0; a.bs_tab = nullptr;
    a.seg_n = seg ? seg->n : 0;
    for (int q = 0; q < 4; ++q) {
        a.seg_end[q] = seg ? seg->end[q] : 0; a.seg_ld[q] = seg ? seg->ld[q] : 0; a.seg_aff[q] = seg ? seg->aff[q] : 0;
        a.seg_y[q] = seg ? seg->y[q] : nullptr;
    }
    const int es = dtype_esize(d->dtype);
    a.xbytes = (unsigned)((int64_t)d->N * d->H * d->W * d->ldx * es); a.wbytes = (unsigned)((int64_t)d->K * d->R * d->S * d->C * es);
    a.H = d->H; a.W = d->W; a.C = d->C; a.ldx = d->ldx;
    a.K = d->K; a.R = d->R; a.S = d->S;
    a.P = d->P; a.Q = d->Q; a.ldy = d->ldy;
    a.ostr_h = d->stride_h; a.ostr_w = d->stride_w; a.base_h = -d->pad_h; a.base_w = -d->pad_w;
    a.ish = 0; a.isw = 0;
    a.wKg = 0; a.wSfull = 0; a.w_rbase = 0; a.w_sbase = 0; a.oH = 0; a.oW = 0; a.o_a = 0; a.o_b = 0;
    a.M = d->N * d->P * d->Q; a.Kg = d->R * d->S * d->C; a.accumulate = 0; a.PQ = d->P * d->Q;
    a.fPQ = make_fastdiv(a.PQ); a.fQ = make_fastdiv(a.Q);
    return run(ctx, a, d->dtype, (hipStream_t)stream);
}

struct BnStatArgs {
    const void* raw;
    int ld;
    const float *mean, *invstd, *scale, *shift;
    float* part;
    const ifcbk_bs_chunk* tab;
};

// the fused variant exists for the plain (stride-1, first-writer, implicit-GEMM) input gradient only
static bool dgrad_bnstat_ok(const ifcbk_conv_desc* d) {
    if (d->stride_h != 1 || d->stride_w != 1) return false;
    return !ifcbk_conv_rows_ok(d->dtype, d->K, d->C, d->R, d->S, d->stride_h, d->stride_w, 2 - d->pad_h, 2 - d->pad_w, d->W);
}

static int dgrad_impl(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const void* dy, const void* wT, void* dx, int accumulate,
                      const BnStatArgs* bs, void* stream);

extern "C" int ifcbk_conv2d_dgrad(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const void* dy, const void* wT, void* dx,
                                  int accumulate, void* stream) {
    return dgrad_impl(ctx, d, dy, wT, dx, accumulate, nullptr, stream);
}

extern "C" int ifcbk_conv2d_dgrad_bnstat_mblocks(const ifcbk_conv_desc* d) {
    if (!d || !dgrad_bnstat_ok(d)) return 0;
    int M = d->N * d->H * d->W;
    if (int fs = ifcbk_conv_flat_rows(d->dtype, d->N, d->P, d->Q, d->K, d->C, d->R, d->S, d->R - 1 - d->pad_h, d->S - 1 - d->pad_w, d->H, d->W)) return fs;
    if (int smt = ifcbk_conv_slab_plan(d->dtype, d->N, d->P, d->Q, d->K, d->C, d->R, d->S, d->R - 1 - d->pad_h, d->S - 1 - d->pad_w, d->H, d->W)) return cdiv(M, 32 * smt);
    {
        int bmt = 0, btn = 0;
        if (ifcbk_conv_big_plan(d->dtype, M, d->C, d->R * d->S * d->K, &bmt, &btn)) return cdiv(M, 32 * bmt);
    }
    return cdiv(M, 64 * pick_wm(M, d->C));
}

extern "C" int ifcbk_conv2d_dgrad_bnstat(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const void* dy, const void* wT, void* dx,
                                         const void* prev_raw, int prev_ld, const float* prev_mean, const float* prev_invstd,
                                         const float* prev_scale, const float* prev_shift, float* part, void* stream) {
    if (!d || !dgrad_bnstat_ok(d)) IFCBK_FAIL(ctx, IFCBK_EUNSUPPORTED, "conv2d_dgrad_bnstat: stride-1 implicit-GEMM input gradients only");
    if (!prev_raw || !prev_mean || !prev_invstd || !prev_scale || !prev_shift || !part)
        IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv2d_dgrad_bnstat: null operand");
    BnStatArgs bs = {prev_raw, prev_ld, prev_mean, prev_invstd, prev_scale, prev_shift, part, nullptr};
    return dgrad_impl(ctx, d, dy, wT, dx, 0, &bs, stream);
}

extern "C" int ifcbk_conv2d_dgrad_bnstat_table(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const void* dy, const void* wT, void* dx,
                                               const ifcbk_bs_chunk* table, float* part, void* stream) {
    if (!d || !dgrad_bnstat_ok(d)) IFCBK_FAIL(ctx, IFCBK_EUNSUPPORTED, "conv2d_dgrad_bnstat_table: stride-1 implicit-GEMM input gradients only");
    if (!table || !part) IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv2d_dgrad_bnstat_table: null operand");
    if (d->C % 8) IFCBK_FAIL(ctx, IFCBK_EINVAL, "conv2d_dgrad_bnstat_table: C must be a multiple of 8");
    BnStatArgs bs = {table, 0, nullptr, nullptr, nullptr, nullptr, part, table};       // raw != null selects MODE 3
    return dgrad_impl(ctx, d, dy, wT, dx, 0, &bs, stream);
}

static int dgrad_impl(ifcbk_ctx* ctx, const ifcbk_conv_desc* d, const void* dy, const void* wT, void* dx, int accumulate,
                      const BnStatArgs* bs, void* stream) {
    if (int e = check_desc(ctx, d, true)) return e;
    // the input gradient of a 3x3/stride-1 conv is a 3x3/stride-1 conv of dy with the flipped filter (wT), padding 2 - pad
    if (!bs && !accumulate && ifcbk_conv_rows_ok(d->dtype, d->K, d->C, d->R, d->S, d->stride_h, d->stride_w, 2 - d->pad_h, 2 - d->pad_w, d->W))
        return ifcbk_conv_rows_launch(ctx, d->K, d->C, d->N, d->P, d->Q, d->ldy, d->H, d->W, d->ldx, 2 - d->pad_h, 2 - d->pad_w, dy,
                                      wT, dx, nullptr, nullptr, nullptr, 0, (hipStream_t)stream);
    // gather over dy [N,P,Q,K] producing dx [N,H,W,C]: roles of (H,W,C) and (P,Q,K) swap
    ConvArgs a;
    a.dbg = 0; a.tr = 0; a.fP = make_fastdiv(1);
    a.ep_scale = nullptr; a.ep_shift = nullptr; a.ep_res = nullptr; a.ep_ldr = 0; a.ep_relu = 0;
    a.x = dy; a.w = wT; a.y = dx; a.part = bs ? bs->part : nullptr;
    a.seg_n = 0;
    a.bs_raw = bs ? bs->raw : nullptr; a.bs_ld = bs ? bs->ld : 0; a.bs_tab = bs ? bs->tab : nullptr;
    a.bs_mean = bs ? bs->mean : nullptr; a.bs_invstd = bs ? bs->invstd : nullptr;
    a.bs_scale = bs ? bs->scale : nullptr; a.bs_shift = bs ? bs->shift : nullptr;
    const int es = dtype_esize(d->dtype);
    a.xbytes = (unsigned)((int64_t)d->N * d->P * d->Q * d->ldy * es); a.wbytes = (unsigned)((int64_t)d->K * d->R * d->S * d->C * es);
    a.H = d->P; a.W = d->Q; a.C = d->K; a.ldx = d->ldy;
    a.K = d->C; a.R = d->R; a.S = d->S;
    a.P = d->H; a.Q = d->W; a.ldy = d->ldx;
    a.ostr_h = 1; a.ostr_w = 1;
    a.wKg = 0; a.wSfull = 0; a.w_rbase = 0; a.w_sbase = 0; a.oH = 0; a.oW = 0; a.o_a = 0; a.o_b = 0;
    a.accumulate = accumulate;
    if (d->stride_h == 2 && d->stride_w == 2 && d->R >= 2 && d->S >= 2 && d->H >= 2 && d->W >= 2) {
        // stride 2: dx pixels of parity class (pa, pb) only see the filter taps r = (pa+pad_h) mod 2 (+2, +4 ...), likewise s:
        // four dense stride-1 convolutions over dy with sub-filters of the flipped filter (9 taps -> 4+2+2+1), each writing
        // its own quarter of dx -- instead of one dilated gather in which 3 of 4 tap products are zeros.
        for (int pa = 0; pa < 2; ++pa)
            for (int pb = 0; pb < 2; ++pb) {
                const int r0 = (pa + d->pad_h) & 1, s0 = (pb + d->pad_w) & 1;
                const int nU = (d->R - r0 + 1) / 2, nV = (d->S - s0 + 1) / 2;
                const int Hc = (d->H - pa + 1) / 2, Wc = (d->W - pb + 1) / 2;
                ConvArgs c = a;
                c.R = nU; c.S = nV;
                c.P = Hc; c.Q = Wc;
                c.base_h = (pa + d->pad_h - r0) / 2 - (nU - 1);
                c.base_w = (pb + d->pad_w - s0) / 2 - (nV - 1);
                c.ish = 0; c.isw = 0;
                c.wKg = d->R * d->S * d->K; c.wSfull = d->S;
                c.w_rbase = d->R - 1 - r0 - 2 * (nU - 1); c.w_sbase = d->S - 1 - s0 - 2 * (nV - 1);
                c.oH = d->H; c.oW = d->W; c.o_a = pa; c.o_b = pb;
                c.M = d->N * Hc * Wc; c.Kg = nU * nV * d->K; c.PQ = Hc * Wc;
                c.fPQ = make_fastdiv(c.PQ); c.fQ = make_fastdiv(c.Q);
                if (int e = run(ctx, c, d->dtype, (hipStream_t)stream)) return e;
            }
        return 0;
    }
    a.base_h = -(d->R - 1 - d->pad_h); a.base_w = -(d->S - 1 - d->pad_w);
    a.ish = d->stride_h == 2 ? 1 : 0; a.isw = d->stride_w == 2 ? 1 : 0;
    a.M = d->N * d->H * d->W; a.Kg = d->R * d->S * d->K; a.PQ = d->H * d->W;
    a.fPQ = make_fastdiv(a.PQ); a.fQ = make_fastdiv(a.Q);
    return run(ctx, a, d->dtype, (hipStream_t)stream);
}
