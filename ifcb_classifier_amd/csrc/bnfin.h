// In-kernel finalize of the per-block partial rows of a BatchNorm reduction ("the last block to arrive sums").
//
// Every kernel that produces BatchNorm partial sums -- the conv epilogues (batch statistics of the forward pass, the backward
// sums of the producing layer in the fused input gradient) and bn_bwd's reduction pass -- used to be followed by a one-block-per-
// 16-channels finalize kernel: 192 launches of 9-14 us per inception_v3 step that sit on the dependency chain between a conv and
// its bn_apply / between the reduction and bn_bwd_dx (1.15 ms of the 21.3 ms step, measured by skipping them).  Here the producer
// finishes the job itself, in two levels so that no block ever sums more than ~sqrt(rows) rows:
//   level 1  the partial rows are cut into groups of GR consecutive rows; a block that has written its row(s) adds them to its
//            group's counter, and the block that completes a group sums the group (rows in index order, double) into gsum[group];
//   level 2  that block then bumps the launch counter; the block that completes the last group sums gsum over the groups (index
//            order, double) and writes what the finalize kernel wrote (forward: mean, invstd, scale, shift, running statistics;
//            backward: dbeta, dgamma and their copy for bn_bwd_dx).
// The result does not depend on which block arrives last: group membership and both summation orders are fixed.
//
// Visibility across the eight XCDs (each has its own L2, coherent only at kernel boundaries for ordinary accesses) without an
// agent-scope release fence -- `__threadfence()` writes the XCD's whole L2 back (buffer_wbl2) and cost 2.3 ms per step when every
// conv block executed one: the partial rows and the group sums are written with agent-scope relaxed atomic stores (sc1: write-
// through), each thread waits for its own stores (s_waitcnt vmcnt(0)) before the block's barrier, thread 0 then bumps the counter
// with an agent-scope relaxed atomic, and the summing block reads rows with agent-scope relaxed atomic loads (sc1: not served
// from a stale line of its own L2) issued after the counter value came back.  Measured cost of the arrive: < 0.1 ms per step.
#pragma once
#include "common.h"
#include <stdlib.h>

struct BnFin {
    unsigned* cnt;          // [0] launch counter, [1 + g] group counters; all zero between launches.  nullptr: no in-kernel finalize
    double* gsum;           // [ngroups][2][C]
    const float* part;      // the partial rows [rows][2][ldp] this launch writes
    // forward: the channels are cut into up to four segments, one BatchNorm each (fused sibling convs); pointers are indexed from
    // the segment's first channel; gamma == nullptr: the segment is skipped (nothing of it is written)
    int nseg, seg_end[4];
    const float* gamma[4];
    const float* beta[4];
    float* rmean[4];        // nullable
    float* rvar[4];
    float* o0;              // backward: dgamma
    float* o1;              //           dbeta
    float* o2;              // forward: mean [C]                 backward: sums[2C] = (dbeta, dgamma) of this batch
    float* o3;              //          invstd
    float* o4;              //          scale
    float* o5;              //          shift
    double invM, unbias;
    float eps, momentum;
    int kind;               // 0 forward statistics, 1 backward sums
    int rows, GR, ngroups, C, ldp, accumulate;
};

// rows per group: even (kernels that write two rows per tile keep a tile inside one group), about sqrt(rows), at least 16
static inline int bnfin_group_rows(int rows) {
    int g = 16;
    while (g * g < rows) g += 2;
    return g;
}
static inline int bnfin_groups(int rows) { const int g = bnfin_group_rows(rows); return (rows + g - 1) / g; }
constexpr int BNFIN_MAX_GROUPS = 1023;          // the counters of a lane: 4 KiB
static inline size_t bnfin_gsum_bytes(int rows, int C) { return (size_t)bnfin_groups(rows) * 2 * C * sizeof(double); }

// device-side descriptor from the public one (host): counters = the launching lane's block, group sums = the head of the lane's
// workspace.  cnt stays null (-> the caller launches the finalize kernel) when the switch is off or the scratch does not fit.
static inline bool bnfin_switch(int bit) {
    static int v = -1;
    if (v < 0) { const char* e = getenv("IFCBK_BN_FIN"); v = e ? atoi(e) : 1; }      // bit 0: conv producers (default), bit 1: bn_bwd's reduction pass
    return (v >> bit) & 1;
}
static inline BnFin bnfin_make(ifcbk_ctx* ctx, const ifcbk_bnfin* h, const float* part, int rows, int C, int ldp) {
    BnFin f = {};
    if (!h || !bnfin_switch(0) || !ctx->fin || rows <= 0 || bnfin_groups(rows) > BNFIN_MAX_GROUPS) return f;
    if (bnfin_gsum_bytes(rows, C) > ctx->ws_bytes) return f;
    f.cnt = ctx->fin;
    f.gsum = (double*)ctx->ws;
    f.part = part;
    f.kind = h->kind;
    f.nseg = h->nseg;
    for (int q = 0; q < 4; ++q) {
        f.seg_end[q] = h->seg_end[q];
        f.gamma[q] = h->gamma[q]; f.beta[q] = h->beta[q]; f.rmean[q] = h->running_mean[q]; f.rvar[q] = h->running_var[q];
    }
    if (h->kind == 0) {
        f.o2 = h->mean; f.o3 = h->invstd; f.o4 = h->scale; f.o5 = h->shift;
        const double M = (double)h->M;
        f.invM = 1.0 / M;
        f.unbias = h->M > 1 ? M / (M - 1.0) : 1.0;
        f.eps = h->eps; f.momentum = h->momentum;
    } else {
        f.o0 = h->dgamma; f.o1 = h->dbeta; f.o2 = h->sums;
        f.accumulate = h->accumulate;
    }
    f.rows = rows; f.GR = bnfin_group_rows(rows); f.ngroups = bnfin_groups(rows); f.C = C; f.ldp = ldp;
    return f;
}

// the partial rows must be written through this
__device__ __forceinline__ void bnfin_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Called by ALL threads of a block (NT of them, thread id t) after the block's partial rows [row0, row0 + nrows) have been written
// with bnfin_store, `writers` = how many blocks write each row (a row's channel ranges come from that many blocks).  Contains
// block barriers: every thread of the block must reach it, with block-uniform arguments.
template <int NT>
__device__ __forceinline__ void bnfin_arrive(const BnFin& f, const int row0, const int nrows, const int writers, const int t) {
    if (!f.cnt) return;
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int g = row0 / f.GR;
    const int r_lo = g * f.GR;
    const int r_hi = min(r_lo + f.GR, f.rows);
    if (t == 0) {
        const unsigned old = __hip_atomic_fetch_add(f.cnt + 1 + g, (unsigned)nrows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (old + (unsigned)nrows == (unsigned)((r_hi - r_lo) * writers)) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    // ---- level 1: this block completed group g
    const int C = f.C;
    for (int i = t; i < 2 * C; i += NT) {
        const int which = i >= C ? 1 : 0;
        const int c = i - which * C;
        const float* p = f.part + ((size_t)r_lo * 2 + which) * f.ldp + c;
        double s = 0.0;
        int r = r_lo;
        for (; r + 8 <= r_hi; r += 8) {          // eight loads in flight, the adds in row order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __hip_atomic_load(p + (size_t)(r - r_lo + u) * 2 * f.ldp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)v[u];
        }
        for (; r < r_hi; ++r) s += (double)__hip_atomic_load(p + (size_t)(r - r_lo) * 2 * f.ldp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(f.gsum + ((size_t)g * 2 + which) * C + c, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        __hip_atomic_store(f.cnt + 1 + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // ready for the next launch
        const unsigned old = __hip_atomic_fetch_add(f.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (old + 1u == (unsigned)f.ngroups) ? 1 : 0;
        if (s_last) __hip_atomic_store(f.cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    // ---- level 2: this block completed the last group
    for (int c = t; c < C; c += NT) {
        double sa = 0.0, sb = 0.0;
        int q = 0;
        for (; q + 4 <= f.ngroups; q += 4) {
            double va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                va[u] = __hip_atomic_load(f.gsum + ((size_t)(q + u) * 2 + 0) * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                vb[u] = __hip_atomic_load(f.gsum + ((size_t)(q + u) * 2 + 1) * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { sa += va[u]; sb += vb[u]; }
        }
        for (; q < f.ngroups; ++q) {
            sa += __hip_atomic_load(f.gsum + ((size_t)q * 2 + 0) * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sb += __hip_atomic_load(f.gsum + ((size_t)q * 2 + 1) * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (f.kind == 0) {
            // what bn_finalize_kernel writes (bn.hip)
            int si = 0;
#pragma unroll
            for (int z = 0; z < 3; ++z)
                if (z + 1 < f.nseg && c >= f.seg_end[z]) si = z + 1;
            if (!f.gamma[si]) continue;
            const int cs = c - (si ? f.seg_end[si - 1] : 0);
            const double mean = sa * f.invM;
            double var = sb * f.invM - mean * mean;
            if (var < 0.0) var = 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
            const float sc = f.gamma[si][cs] * invstd;
            f.o2[c] = (float)mean;
            f.o3[c] = invstd;
            f.o4[c] = sc;
            f.o5[c] = f.beta[si][cs] - (float)mean * sc;
            if (f.rmean[si]) {
                f.rmean[si][cs] = (1.f - f.momentum) * f.rmean[si][cs] + f.momentum * (float)mean;
                f.rvar[si][cs] = (1.f - f.momentum) * f.rvar[si][cs] + f.momentum * (float)(var * f.unbias);
            }
        } else {
            // what bn_bwd_finalize_kernel writes: first sum = dbeta, second = dgamma
            f.o2[c] = (float)sa;
            f.o2[C + c] = (float)sb;
            f.o1[c] = f.accumulate ? f.o1[c] + (float)sa : (float)sa;
            f.o0[c] = f.accumulate ? f.o0[c] + (float)sb : (float)sb;
        }
    }
}
