// ctx lifetime, workspace, error text and the program runner of libifcbk.
#include "common.h"
#include <execinfo.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

// IFCBK_SEGV_BACKTRACE=1: print the host call stack (frames of this library and of the HIP runtime) when the process takes a
// SIGSEGV / SIGABRT, then die with the default action.  Diagnostic only (used to locate the runtime call that faults in a
// multi-stream hipGraph capture); async-signal-safe calls only: backtrace_symbols_fd writes straight to stderr.
static void segv_backtrace(int sig) {
    static const char head[] = "\n[ifcbk] fatal signal -- host backtrace:\n";
    (void)!write(2, head, sizeof(head) - 1);
    void* frames[48];
    const int n = backtrace(frames, 48);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
static void maybe_install_backtrace() {
    static int done = 0;
    if (done) return;
    done = 1;
    const char* e = getenv("IFCBK_SEGV_BACKTRACE");
    if (!e || !atoi(e)) return;
    void* warm[4];
    (void)backtrace(warm, 4);          // loads libgcc now, not inside the handler
    // on an alternate stack: a stack overflow (runaway recursion) must still be able to report itself
    static char altstack[1 << 16];
    stack_t ss;
    ss.ss_sp = altstack; ss.ss_size = sizeof(altstack); ss.ss_flags = 0;
    sigaltstack(&ss, nullptr);
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = segv_backtrace;
    sa.sa_flags = SA_ONSTACK | SA_NODEFER;
    sigaction(SIGSEGV, &sa, nullptr);
    sigaction(SIGABRT, &sa, nullptr);
    sigaction(SIGBUS, &sa, nullptr);
}

extern "C" const char* ifcbk_version(void) { return "ifcbk 0.1 (gfx950, bf16 MFMA)"; }

static char g_create_err[256] = "no ifcbk_ctx_create call has failed";      // ifcbk_last_error(NULL)

extern "C" int ifcbk_ctx_create(int device, ifcbk_ctx** out) {
    if (!out) return IFCBK_EINVAL;
    *out = nullptr;
    maybe_install_backtrace();
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || device < 0 || device >= n) {
        snprintf(g_create_err, sizeof(g_create_err), "ifcbk_ctx_create(device=%d): hipGetDeviceCount -> %d device(s), %s", device, n,
                 hipGetErrorString(e));
        return IFCBK_EHIP;
    }
    ifcbk_ctx* c = (ifcbk_ctx*)calloc(1, sizeof(ifcbk_ctx));
    if (!c) return IFCBK_ENOMEM;
    c->device = device;
    c->ws_lanes = 4;
    const char* what = "hipSetDevice";
    e = hipSetDevice(device);
    if (e == hipSuccess) { what = "hipMalloc"; e = hipMalloc(&c->zeros, 4096); }
    if (e == hipSuccess) { what = "hipMemset"; e = hipMemset(c->zeros, 0, 4096); }
    if (e != hipSuccess) {
        snprintf(g_create_err, sizeof(g_create_err), "ifcbk_ctx_create(device=%d): %s: %s", device, what, hipGetErrorString(e));
        free(c);
        return IFCBK_EHIP;
    }
    *out = c;
    return IFCBK_OK;
}

static void graph_free(ifcbk_ctx* c, ifcbk_graph* g);

extern "C" int ifcbk_ctx_destroy(ifcbk_ctx* c) {
    if (!c) return IFCBK_OK;
    // Lifetime rule: a graph's kernel nodes point into this ctx's arenas and were ordered through its capture streams and
    // events, so no graph outlives its ctx -- whatever the caller did not destroy goes first, before anything it refers to.
    while (c->graphs) graph_free(c, c->graphs);
    if (c->ws_base) (void)hipFree(c->ws_base);
    if (c->zeros) (void)hipFree(c->zeros);
    for (int l = 1; l < IFCBK_MAX_LANES; ++l) {
        if (c->lane_st[l]) (void)hipStreamDestroy(c->lane_st[l]);
        if (c->cap_st[l]) (void)hipStreamDestroy(c->cap_st[l]);
    }
    for (int i = 0; i < c->n_xev; ++i) (void)hipEventDestroy(c->xev[i]);
    for (int i = 0; i < c->n_cev; ++i) (void)hipEventDestroy(c->cev[i]);
    free(c->cev);
    for (int i = 0; i < c->n_ev; ++i) (void)hipEventDestroy(c->ev[i]);
    free(c->ev);
    for (int s = 0; s < 256; ++s) {
        for (int i = 0; i < c->slot_n[s]; ++i) (void)hipEventDestroy(c->slot_ev[s][i]);
        free(c->slot_ev[s]);
        free(c->slot_rec[s]);
    }
    free(c);
    return IFCBK_OK;
}

extern "C" int ifcbk_ctx_reserve(ifcbk_ctx* c, size_t bytes) {
    if (!c) return IFCBK_EINVAL;
    if (bytes <= c->ws_bytes) return IFCBK_OK;
    IFCBK_HIP(c, hipSetDevice(c->device));
    IFCBK_HIP(c, hipDeviceSynchronize());
    if (c->ws_base) IFCBK_HIP(c, hipFree(c->ws_base));
    c->ws = c->ws_base = nullptr;
    c->ws_bytes = 0;
    bytes = (bytes + 255) & ~(size_t)255;
    IFCBK_HIP(c, hipMalloc(&c->ws_base, bytes * (size_t)c->ws_lanes));     // one arena per program lane in use
    c->ws = c->ws_base;
    c->ws_bytes = bytes;
    ++c->ws_epoch;
    return IFCBK_OK;
}

extern "C" size_t ifcbk_ctx_workspace_bytes(ifcbk_ctx* c) { return c ? c->ws_bytes : 0; }

extern "C" int ifcbk_ctx_set_lanes(ifcbk_ctx* c, int lanes) {
    if (!c || lanes < 1 || lanes > IFCBK_MAX_LANES) return IFCBK_EINVAL;
    if (lanes == c->ws_lanes) return IFCBK_OK;
    if (c->ws_base && lanes > c->ws_lanes)
        IFCBK_FAIL(c, IFCBK_EINVAL, "ctx_set_lanes: %d arenas are reserved; set the lane count before ifcbk_ctx_reserve", c->ws_lanes);
    if (!c->ws_base) c->ws_lanes = lanes;      // (fewer lanes than reserved: keep the arenas, nothing to do)
    return IFCBK_OK;
}

extern "C" int ifcbk_ctx_live_graphs(ifcbk_ctx* c) { return c ? c->n_graphs : 0; }

extern "C" const char* ifcbk_last_error(ifcbk_ctx* c) { return c ? c->err : g_create_err; }

static int run_one(ifcbk_ctx* c, const ifcbk_op* o, void* st) {
    void* const* p = o->p;
    const int acc = o->flags & 1, pacc = (o->flags >> 1) & 1;
    switch (o->kind) {
        case IFCBK_OP_CONV_FWD: return ifcbk_conv2d_fwd(c, &o->u.conv, p[0], p[1], p[2], (float*)p[3], st);
        case IFCBK_OP_CONV_FWD_AFFINE:
            return ifcbk_conv2d_fwd_affine(c, &o->u.conv, p[0], p[1], p[2], (const float*)p[3], (const float*)p[4], p[5],
                                           (int)o->i[0], (o->flags >> 2) & 1, st);
        case IFCBK_OP_CONV_DGRAD: return ifcbk_conv2d_dgrad(c, &o->u.conv, p[0], p[1], p[2], acc, st);
        case IFCBK_OP_CONV_WGRAD: return ifcbk_conv2d_wgrad(c, &o->u.conv, p[0], p[1], (float*)p[2], acc, st);
        case IFCBK_OP_CONV_FWD_AFFINE_MAXPOOL:
            return ifcbk_conv2d_fwd_affine_maxpool(c, &o->u.conv, p[0], p[1], p[2], (int)o->i[0], (const float*)p[3], (const float*)p[4],
                                                   (o->flags >> 2) & 1, st);
        case IFCBK_OP_STEM_U8_FWD:
            return ifcbk_stem_u8_fwd(c, &o->u.conv, (const uint8_t*)p[0], (const float*)p[1], (const float*)p[2], p[3], (float*)p[4],
                                     (const float*)p[5], (const float*)p[6], (o->flags >> 2) & 1, st);
        case IFCBK_OP_STEM_U8_WGRAD:
            return ifcbk_stem_u8_wgrad(c, &o->u.conv, (const uint8_t*)p[0], p[1], (const float*)p[2], (float*)p[3], acc, st);
        case IFCBK_OP_CONV_WGRAD_SEG: {
            float* dws[4];
            int32_t ks[4];
            int ns = 0;
            for (int k = 0; k < 4 && o->i[k] > 0; ++k) { dws[ns] = (float*)p[2 + k]; ks[ns] = (int32_t)o->i[k]; ++ns; }
            return ifcbk_conv2d_wgrad_segments(c, &o->u.conv, p[0], p[1], ns, dws, ks, acc, st);
        }
        case IFCBK_OP_WEIGHT_PACK_MULTI:
            return ifcbk_weight_pack_multi(c, (const ifcbk_pack_item*)p[0], (int)o->i[0], o->i[1], (int)o->i[2], st);
        case IFCBK_OP_WEIGHT_PACK: return ifcbk_weight_pack(c, &o->u.conv, (const float*)p[0], p[1], p[2], st);
        case IFCBK_OP_BN_FINALIZE:
            return ifcbk_bn_finalize_ld(c, &o->u.bn, (const float*)p[0], (int)o->i[0], (int)o->i[1], (const float*)p[1], (const float*)p[2],
                                     (float*)p[3], (float*)p[4], (float*)p[5], (float*)p[6], (float*)p[7], (float*)p[8], st);
        case IFCBK_OP_BN_APPLY:
            return ifcbk_bn_apply(c, &o->u.bn, p[0], (const float*)p[1], (const float*)p[2], p[3], (int)o->i[0], p[4], st);
        case IFCBK_OP_BN_BWD:
            return ifcbk_bn_bwd(c, &o->u.bn, p[0], p[1], p[2], (int)o->i[0], (const float*)p[3], (const float*)p[4],
                                (const float*)p[5], p[6], (int)o->i[1], p[7], (int)o->i[2], acc | (((o->flags >> 3) & 1) << 1), (float*)p[8], (float*)p[9],
                                pacc, (const float*)p[10], (const float*)p[11], st);
        case IFCBK_OP_CONV_DGRAD_BNSTAT:
            return ifcbk_conv2d_dgrad_bnstat(c, &o->u.conv, p[0], p[1], p[2], p[3], (int)o->i[0], (const float*)p[4], (const float*)p[5],
                                             (const float*)p[6], (const float*)p[7], (float*)p[8], st);
        case IFCBK_OP_BN_BWD_PARTIALS:
            return ifcbk_bn_bwd_partials_ld(c, &o->u.bn, p[0], p[1], (int)o->i[0], (const float*)p[2], (const float*)p[3], (const float*)p[4],
                                            (const float*)p[5], (const float*)p[6], (const float*)p[7], (int)o->i[1], (int)o->i[3], p[8],
                                            (int)o->i[2], (float*)p[9], (float*)p[10], pacc, st);
        case IFCBK_OP_CONV_DGRAD_BNSTAT_TAB:
            return ifcbk_conv2d_dgrad_bnstat_table(c, &o->u.conv, p[0], p[1], p[2], (const ifcbk_bs_chunk*)p[3], (float*)p[4], st);
        case IFCBK_OP_CONV_FWD_AFFINE_SEG: {
            // i[s] = channels | pixel stride << 20 | affine << 40 of segment s (0 = unused); p[2..5] = destinations
            void* ys[4];
            int32_t ld[4], ks[4], af[4];
            int ns = 0;
            for (int k = 0; k < 4 && (o->i[k] & 0xfffff) > 0; ++k) {
                ys[ns] = p[2 + k]; ks[ns] = (int32_t)(o->i[k] & 0xfffff); ld[ns] = (int32_t)((o->i[k] >> 20) & 0xfffff);
                af[ns] = (int32_t)((o->i[k] >> 40) & 1); ++ns;
            }
            return ifcbk_conv2d_fwd_affine_segments(c, &o->u.conv, p[0], p[1], ns, ys, ld, ks, af, (const float*)p[6], (const float*)p[7], st);
        }
        case IFCBK_OP_BN_STATS: return ifcbk_bn_stats(c, &o->u.bn, p[0], (float*)p[1], st);
        case IFCBK_OP_AVGPOOL_AFFINE:
            return ifcbk_avgpool3x3_affine(c, &o->u.pool, p[0], (const float*)p[1], (const float*)p[2], (o->flags >> 2) & 1, p[3], st);
        case IFCBK_OP_BN_APPLY_MAXPOOL:
            return ifcbk_bn_apply_maxpool(c, &o->u.pool, p[0], (const float*)p[1], (const float*)p[2], (int)o->i[0], p[3], (uint8_t*)p[4], st);
        case IFCBK_OP_BN_BWD_MAXPOOL:
            return ifcbk_bn_bwd_maxpool(c, &o->u.pool, p[0], p[1], (const uint8_t*)p[2], (const float*)p[3], (const float*)p[4],
                                        (const float*)p[5], (const float*)p[6], (const float*)p[7], (int)o->i[0], p[8], (int)o->i[1],
                                        (float*)p[9], (float*)p[10], pacc, st);
        case IFCBK_OP_MAXPOOL_FWD: return ifcbk_maxpool_fwd(c, &o->u.pool, p[0], p[1], (uint8_t*)p[2], st);
        case IFCBK_OP_MAXPOOL_BWD: return ifcbk_maxpool_bwd(c, &o->u.pool, p[0], (const uint8_t*)p[1], p[2], acc, st);
        case IFCBK_OP_AVGPOOL_FWD: return ifcbk_avgpool_fwd(c, &o->u.pool, p[0], p[1], st);
        case IFCBK_OP_AVGPOOL_BWD: return ifcbk_avgpool_bwd(c, &o->u.pool, p[0], p[1], acc, st);
        case IFCBK_OP_HEAD_FWD:
            return ifcbk_head_fwd(c, &o->u.head, p[0], (const uint8_t*)p[1], (const float*)p[2], (const float*)p[3],
                                  (float*)p[4], (float*)p[5], st);
        case IFCBK_OP_HEAD_BWD:
            return ifcbk_head_bwd(c, &o->u.head, (const float*)p[0], (const float*)p[1], (const uint8_t*)p[2],
                                  (const float*)p[3], (float*)p[4], (float*)p[5], p[6], (int)o->i[0], pacc, st);
        case IFCBK_OP_SOFTMAX_XENT:
            return ifcbk_softmax_xent(c, (const float*)p[0], (const int64_t*)p[1], (int)o->i[0], (int)o->i[1], o->f[0],
                                      (float*)p[2], acc, (float*)p[3], st);
        case IFCBK_OP_SOFTMAX: return ifcbk_softmax(c, (const float*)p[0], (int)o->i[0], (int)o->i[1], (float*)p[1], st);
        case IFCBK_OP_ADAM:
            return ifcbk_adam_flat(c, (float*)p[0], (const float*)p[1], (float*)p[2], (float*)p[3], o->i[0], o->f[0], o->f[1],
                                   o->f[2], o->f[3], o->f[4], (int)o->i[1], o->f[5], st);
        case IFCBK_OP_SGD:
            return ifcbk_sgd_flat(c, (float*)p[0], (const float*)p[1], (float*)p[2], o->i[0], o->f[0], o->f[1], o->f[2], o->f[3], st);
        case IFCBK_OP_MEMSET:
            IFCBK_HIP(c, hipMemsetAsync(p[0], (int)o->i[1], (size_t)o->i[0], (hipStream_t)st));
            return 0;
        case IFCBK_OP_COPY2D:
            IFCBK_HIP(c, hipMemcpy2DAsync(p[0], (size_t)o->i[0], p[1], (size_t)o->i[1], (size_t)o->i[2], (size_t)o->i[3],
                                          hipMemcpyDeviceToDevice, (hipStream_t)st));
            return 0;
        case IFCBK_OP_BIAS_RELU_BWD:
            return ifcbk_bias_relu_bwd(c, o->u.bn.M, o->u.bn.C, o->u.bn.dtype, p[0], o->u.bn.ldx, p[1], o->u.bn.ldy, p[2], (int)o->i[0],
                                       o->u.bn.relu, (float*)p[3], pacc, st);
        case IFCBK_OP_DROPOUT:
            return ifcbk_dropout_apply(c, o->i[0], (int)o->i[1], p[0], (const uint8_t*)p[1], o->f[0], p[2], acc, st);
        case IFCBK_OP_FLATTEN_CHW:
            return ifcbk_flatten_chw(c, (int)o->i[0], (int)o->i[1], (int)o->i[2], (int)(o->i[3] >> 32), p[0], (int)(o->i[3] & 0xffffffff),
                                     p[1], (o->flags >> 2) & 1, acc, st);
        case IFCBK_OP_CONV_WGRAD_GROUP: {
            const ifcbk_wgrad_item* it = (const ifcbk_wgrad_item*)p[0];
            const int n = (int)o->i[0];
            if (!it || n < 1 || n > 8) IFCBK_FAIL(c, IFCBK_EINVAL, "run_program: wgrad group of %d", n);
            ifcbk_conv_desc ds[8];
            const void *xs[8], *dys[8];
            float* dws[8];
            for (int k = 0; k < n; ++k) { ds[k] = it[k].d; xs[k] = it[k].x; dys[k] = it[k].dy; dws[k] = it[k].dw; }
            return ifcbk_conv2d_wgrad_group(c, n, ds, xs, dys, dws, acc, st);
        }
        case IFCBK_OP_STEP_COUNTERS:
            return ifcbk_step_counters(c, (int64_t*)p[0], (int)o->i[0], (float*)p[1], (const float*)p[2], st);
        case IFCBK_OP_DROPOUT_MASK:
            return ifcbk_dropout_mask(c, (uint8_t*)p[0], o->i[0], o->f[0], (uint64_t)o->i[1], (uint64_t)o->i[2], st);
        default: IFCBK_FAIL(c, IFCBK_EINVAL, "run_program: unknown op kind %d", o->kind);
    }
}

// ---------------------------------------------------------------- program runner with lanes
// An op carries its lane (flags bits 8-10) and a wait mask (bits 12-19): "before this op, make my lane wait for
// everything queued so far on those lanes".  Lane 0 is the caller's stream; lanes 1-7 are ctx-owned streams.  The host
// (engine.py) computes lanes and masks from the static data flow, so independent branches of the graph overlap --
// the tail of one kernel (a 578-block grid on 512 resident slots runs a nearly empty second round) is filled by the
// next branch's kernel.  Every lane joins lane 0 at the end of the program: to the caller it is one stream-ordered call.
static inline int op_lane(const ifcbk_op* o) { return (o->flags >> 8) & 7; }
static inline int op_wait(const ifcbk_op* o) { return (o->flags >> 12) & 255; }

// streams and ordering events of the lanes in `used` (created outside of any stream capture)
static int lane_resources(ifcbk_ctx* c, int used) {
    if (c->n_xev < 64) {
        for (int i = c->n_xev; i < 64; ++i) IFCBK_HIP(c, hipEventCreateWithFlags(&c->xev[i], hipEventDisableTiming));
        c->n_xev = 64;
    }
    for (int l = 1; l < IFCBK_MAX_LANES; ++l)
        if ((used >> l & 1) && !c->lane_st[l]) IFCBK_HIP(c, hipStreamCreateWithFlags(&c->lane_st[l], hipStreamNonBlocking));
    return IFCBK_OK;
}

static int lane_order(ifcbk_ctx* c, hipStream_t waiter, hipStream_t waited) {
    hipEvent_t ev;
    if (c->capturing) {
        // a stream capture must not record an event twice (re-recording one whose earlier record a captured wait still
        // refers to crashed the runtime with three lanes): one event per edge, pre-created by ifcbk_program_capture
        if (c->cev_next >= c->n_cev) IFCBK_FAIL(c, IFCBK_EINVAL, "lane_order: capture event pool exhausted");
        ev = c->cev[c->cev_next++];
    } else {
        ev = c->xev[c->xev_next];
        c->xev_next = (c->xev_next + 1) & 63;
    }
    IFCBK_HIP(c, hipEventRecord(ev, waited));
    IFCBK_HIP(c, hipStreamWaitEvent(waiter, ev, 0));
    return IFCBK_OK;
}

// ev: null, or 2n events (start, stop of every op, recorded on the op's own lane)
static int run_lanes(ifcbk_ctx* c, const ifcbk_op* ops, int n, hipStream_t s0, hipEvent_t* ev, unsigned char* rec = nullptr) {
    hipStream_t st[IFCBK_MAX_LANES] = {s0};
    int used = 1;
    for (int i = 0; i < n; ++i) used |= 1 << op_lane(&ops[i]);
    if (used >> c->ws_lanes) IFCBK_FAIL(c, IFCBK_EINVAL, "run_program: an op names a lane beyond the %d this ctx has arenas for (ifcbk_ctx_set_lanes)", c->ws_lanes);
    if (int e = lane_resources(c, used)) return e;
    for (int l = 1; l < IFCBK_MAX_LANES; ++l)
        if (used >> l & 1) {
            // Inside a stream capture the lanes run on capture-only streams: a stream that launches real work is never part of a
            // capture, so nothing a graph was recorded through is ever synchronised, re-created or given work later on.
            st[l] = c->capturing ? c->cap_st[l] : c->lane_st[l];
            if (int e = lane_order(c, st[l], s0)) return e;                 // fork: the lane starts after the caller's prior work
        }
    int rc = IFCBK_OK;
    for (int i = 0; i < n && !rc; ++i) {
        const int L = op_lane(&ops[i]);
        const int wm = op_wait(&ops[i]) & used & ~(1 << L);
        for (int j = 0; j < IFCBK_MAX_LANES && !rc; ++j)
            if (wm >> j & 1) {
                if (c->capturing && L != 0 && j != 0) {
                    // Stream capture: an event wait between two FORKED streams makes each the other's "parallel capture
                    // stream" inside the ROCm 7 runtime, and hipStreamEndCapture then recurses over that cycle until the stack
                    // overflows (backtrace: an unexported libamdhip64 function at +0x2d3450 that walks a stream's capture
                    // events, calls itself on every parallel stream and then clears the capture status -- Stream::EndCapture).
                    // Edges between the origin stream and a forked stream are fine (every 2-lane capture is made of them), so
                    // a lane-to-lane dependency is recorded as two of those: j -> origin -> L.  Over-constrains the origin
                    // (lane 0 also waits for lane j), never under-constrains.
                    rc = lane_order(c, st[0], st[j]);
                    if (!rc) rc = lane_order(c, st[L], st[0]);
                } else {
                    rc = lane_order(c, st[L], st[j]);
                }
            }
        if (rc) break;
        c->ws = (char*)c->ws_base + (size_t)L * c->ws_bytes;
        const bool timed = ev && (!rec || (ops[i].flags & 0x80));      // run_program_ev brackets only ops with flags bit 7
        if (rec) rec[i] = timed;
        if (timed) IFCBK_HIP(c, hipEventRecord(ev[2 * i], st[L]));
        rc = run_one(c, &ops[i], st[L]);
        if (rc) {
            size_t len = strlen(c->err);
            snprintf(c->err + len, sizeof(c->err) - len, " [op %d kind %d]", i, ops[i].kind);
            break;
        }
        if (timed) IFCBK_HIP(c, hipEventRecord(ev[2 * i + 1], st[L]));
    }
    c->ws = c->ws_base;
    for (int l = 1; l < IFCBK_MAX_LANES; ++l)                                // join, also on the error path
        if (used >> l & 1) (void)lane_order(c, s0, st[l]);
    return rc;
}

extern "C" int ifcbk_run_program(ifcbk_ctx* c, const ifcbk_op* ops, int n, void* stream, float* op_ms) {
    if (!c || (!ops && n > 0)) return IFCBK_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (op_ms) {
        if (c->n_ev < 2 * n) {
            hipEvent_t* ev = (hipEvent_t*)realloc(c->ev, sizeof(hipEvent_t) * (2 * n));
            if (!ev) IFCBK_FAIL(c, IFCBK_ENOMEM, "run_program: event pool");
            c->ev = ev;
            for (int i = c->n_ev; i < 2 * n; ++i) IFCBK_HIP(c, hipEventCreate(&c->ev[i]));
            c->n_ev = 2 * n;
        }
    }
    if (int e = run_lanes(c, ops, n, st, op_ms ? c->ev : nullptr)) return e;
    if (op_ms) {
        IFCBK_HIP(c, hipStreamSynchronize(st));
        for (int i = 0; i < n; ++i) IFCBK_HIP(c, hipEventElapsedTime(&op_ms[i], c->ev[2 * i], c->ev[2 * i + 1]));
    }
    return IFCBK_OK;
}
extern "C" int ifcbk_run_program_ev(ifcbk_ctx* c, const ifcbk_op* ops, int n, void* stream, int slot) {
    if (!c || slot < 0 || slot >= 256 || (!ops && n > 0)) return IFCBK_EINVAL;
    if (c->slot_n[slot] < 2 * n) {
        hipEvent_t* ev = (hipEvent_t*)realloc(c->slot_ev[slot], sizeof(hipEvent_t) * (2 * n));
        if (!ev) IFCBK_FAIL(c, IFCBK_ENOMEM, "run_program_ev: event pool");
        c->slot_ev[slot] = ev;
        for (int i = c->slot_n[slot]; i < 2 * n; ++i) IFCBK_HIP(c, hipEventCreate(&ev[i]));
        c->slot_n[slot] = 2 * n;
        unsigned char* r = (unsigned char*)realloc(c->slot_rec[slot], n > 0 ? n : 1);
        if (!r) IFCBK_FAIL(c, IFCBK_ENOMEM, "run_program_ev: event pool");
        c->slot_rec[slot] = r;
    }
    memset(c->slot_rec[slot], 0, n);
    return run_lanes(c, ops, n, (hipStream_t)stream, c->slot_ev[slot], c->slot_rec[slot]);
}
extern "C" int ifcbk_program_times(ifcbk_ctx* c, int slot, int n, float* op_ms) {
    if (!c || slot < 0 || slot >= 256 || !op_ms || c->slot_n[slot] < 2 * n) return IFCBK_EINVAL;
    for (int i = 0; i < n; ++i) {
        op_ms[i] = 0.f;
        if (c->slot_rec[slot][i]) IFCBK_HIP(c, hipEventElapsedTime(&op_ms[i], c->slot_ev[slot][2 * i], c->slot_ev[slot][2 * i + 1]));
    }
    return IFCBK_OK;
}

// ---------------------------------------------------------------- hipGraph replay
// Lifetime rule (round 5 audit of the round-4 hipGraphLaunch SIGSEGV): every graph is OWNED by the ctx it was captured
// through -- linked into ctx->graphs, launched and destroyed only through that ctx (checked), and destroyed by ifcbk_ctx_destroy
// before the arenas, streams and events it refers to.  Before this rule, handles the caller dropped leaked their hipGraphExec
// (with the runtime's per-exec parallel streams) for the life of the process.
struct ifcbk_graph {
    unsigned magic;              // 0x69666b67 while live
    ifcbk_ctx* owner;
    ifcbk_graph *prev, *next;
    hipGraph_t graph;
    hipGraphExec_t exec;
    unsigned ws_epoch;
    int n_ops;
};
constexpr unsigned GRAPH_MAGIC = 0x69666b67u;

static void graph_free(ifcbk_ctx* c, ifcbk_graph* g) {
    if (g->prev) g->prev->next = g->next; else c->graphs = g->next;
    if (g->next) g->next->prev = g->prev;
    --c->n_graphs;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    g->magic = 0; g->owner = nullptr;
    free(g);
}

extern "C" int ifcbk_program_capture(ifcbk_ctx* c, const ifcbk_op* ops, int n, ifcbk_graph** out) {
    if (!c || !out || !ops || n <= 0) return IFCBK_EINVAL;
    *out = nullptr;
    IFCBK_HIP(c, hipSetDevice(c->device));
    int used = 1;
    for (int i = 0; i < n; ++i) used |= 1 << op_lane(&ops[i]);
    if (int e = lane_resources(c, used)) return e;
    for (int l = 1; l < IFCBK_MAX_LANES; ++l)
        if ((used >> l & 1) && !c->cap_st[l]) IFCBK_HIP(c, hipStreamCreateWithFlags(&c->cap_st[l], hipStreamNonBlocking));
    // one ordering event per fork / wait / join edge of this program (upper bound: 3 per op + 2 per lane)
    {
        const int need = 6 * n + 2 * IFCBK_MAX_LANES;      // (a lane-to-lane edge is recorded as two edges through the origin)
        if (c->n_cev < need) {
            hipEvent_t* ev = (hipEvent_t*)realloc(c->cev, sizeof(hipEvent_t) * need);
            if (!ev) IFCBK_FAIL(c, IFCBK_ENOMEM, "program_capture: event pool");
            c->cev = ev;
            for (int i = c->n_cev; i < need; ++i) IFCBK_HIP(c, hipEventCreateWithFlags(&c->cev[i], hipEventDisableTiming));
            c->n_cev = need;
        }
        c->cev_next = 0;
    }
    // capture on a private stream: the caller's stream may be the legacy default stream, which cannot capture
    hipStream_t cs = nullptr;
    IFCBK_HIP(c, hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    hipError_t he = hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed);
    if (he != hipSuccess) {
        (void)hipStreamDestroy(cs);
        IFCBK_FAIL(c, IFCBK_EHIP, "program_capture: hipStreamBeginCapture: %s", hipGetErrorString(he));
    }
    c->capturing = 1;
    const int rc = run_lanes(c, ops, n, cs, nullptr);            // records; the lanes fork from and join `cs`
    c->capturing = 0;
    hipGraph_t g = nullptr;
    he = hipStreamEndCapture(cs, &g);                            // always ends the capture, also after a failed op
    (void)hipStreamDestroy(cs);
    if (rc) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
    }
    if (he != hipSuccess || !g) IFCBK_FAIL(c, IFCBK_EHIP, "program_capture: hipStreamEndCapture: %s", hipGetErrorString(he));
    hipGraphExec_t x = nullptr;
    he = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    if (he != hipSuccess) {
        (void)hipGraphDestroy(g);
        IFCBK_FAIL(c, IFCBK_EHIP, "program_capture: hipGraphInstantiate: %s", hipGetErrorString(he));
    }
    ifcbk_graph* r = (ifcbk_graph*)calloc(1, sizeof(ifcbk_graph));
    if (!r) {
        (void)hipGraphExecDestroy(x);
        (void)hipGraphDestroy(g);
        IFCBK_FAIL(c, IFCBK_ENOMEM, "program_capture: out of host memory");
    }
    r->magic = GRAPH_MAGIC; r->owner = c;
    r->graph = g; r->exec = x; r->ws_epoch = c->ws_epoch; r->n_ops = n;
    r->next = c->graphs;
    if (c->graphs) c->graphs->prev = r;
    c->graphs = r;
    ++c->n_graphs;
    *out = r;
    return IFCBK_OK;
}

extern "C" int ifcbk_graph_launch(ifcbk_ctx* c, ifcbk_graph* g, void* stream) {
    if (!c || !g) return IFCBK_EINVAL;
    if (g->magic != GRAPH_MAGIC || g->owner != c) IFCBK_FAIL(c, IFCBK_EINVAL, "graph_launch: not a live graph of this ctx");
    if (g->ws_epoch != c->ws_epoch) IFCBK_FAIL(c, IFCBK_EINVAL, "graph_launch: the workspace moved since this graph was captured; capture it again");
    IFCBK_HIP(c, hipGraphLaunch(g->exec, (hipStream_t)stream));
    return IFCBK_OK;
}

extern "C" int ifcbk_graph_destroy(ifcbk_ctx* c, ifcbk_graph* g) {
    if (!g) return IFCBK_OK;
    if (!c) return IFCBK_EINVAL;
    if (g->magic != GRAPH_MAGIC || g->owner != c) IFCBK_FAIL(c, IFCBK_EINVAL, "graph_destroy: not a live graph of this ctx");
    graph_free(c, g);
    return IFCBK_OK;
}

extern "C" int ifcbk_op_kernel(const ifcbk_op* o, char* name, size_t cap) {
    if (!o || !name || cap < 1) return IFCBK_EINVAL;
    name[0] = 0;
    switch (o->kind) {
        case IFCBK_OP_CONV_FWD_AFFINE_SEG: {
            const ifcbk_conv_desc& d = o->u.conv;
            int bmt = 0, btn = 0;
            if (ifcbk_conv_big_plan(d.dtype, d.N * d.P * d.Q, d.K, d.R * d.S * d.C, &bmt, &btn))
                snprintf(name, cap, "conv_pp2<%d, %d, %d, 4>", btn, bmt, bmt == 10 ? 4 : bmt / 2);
            else
                snprintf(name, cap, "conv_igemm<unsigned short, %d, 2, 2, 4>", ifcbk_conv_fwd_nt(d.K, d.N * d.P * d.Q));
            break;
        }
        case IFCBK_OP_CONV_FWD: case IFCBK_OP_CONV_FWD_AFFINE: {
            const ifcbk_conv_desc& d = o->u.conv;
            int wm = ifcbk_conv_fwd_wm(d.N * d.P * d.Q, d.K);
            int bmt = 0, btn = 0;
            const bool rows = ifcbk_conv_rows_ok(d.dtype, d.C, d.K, d.R, d.S, d.stride_h, d.stride_w, d.pad_h, d.pad_w, d.Q) && !(o->kind == IFCBK_OP_CONV_FWD_AFFINE && o->p[5]);
            if (!rows && d.stride_h == 1 && d.stride_w == 1 && !(o->kind == IFCBK_OP_CONV_FWD_AFFINE && o->p[5]) &&
                ifcbk_conv_flat_rows(d.dtype, d.N, d.H, d.W, d.C, d.K, d.R, d.S, d.pad_h, d.pad_w, d.P, d.Q)) {
                snprintf(name, cap, "conv_flat<%d, %d, %d, %d, 0>", d.C, d.K, d.R, d.S);
                break;
            }
            if (!rows && d.stride_h == 1 && d.stride_w == 1) {
                if (const int smt = ifcbk_conv_slab_plan(d.dtype, d.N, d.H, d.W, d.C, d.K, d.R, d.S, d.pad_h, d.pad_w, d.P, d.Q)) {
                    snprintf(name, cap, "conv_slab<%d, %d, %d, %d, 50, 0>", d.K <= 128 ? 2 : 3, smt, smt == 10 ? 4 : smt / 2, d.R * d.S);
                    break;
                }
            }
            if (!rows && !(o->kind == IFCBK_OP_CONV_FWD_AFFINE && o->p[5]) &&
                ifcbk_conv_pp3_plan(d.dtype, d.N * d.P * d.Q, d.K, d.R * d.S * d.C, o->kind == IFCBK_OP_CONV_FWD_AFFINE ? 1 : 0)) {
                ifcbk_conv_pp3_name(d.R * d.S * d.C, o->kind == IFCBK_OP_CONV_FWD_AFFINE,
                                    d.R == 1 && d.S == 1 && d.pad_h == 0 && d.pad_w == 0 && d.stride_h == 1 && d.stride_w == 1, name, cap);
                break;
            }
            if (!rows && ifcbk_conv_big_plan(d.dtype, d.N * d.P * d.Q, d.K, d.R * d.S * d.C, &bmt, &btn)) {
                snprintf(name, cap, "conv_pp2<%d, %d, %d, 0>", btn, bmt, bmt == 10 ? 4 : bmt / 2);
                break;
            }
            if (ifcbk_conv_rows_ok(d.dtype, d.C, d.K, d.R, d.S, d.stride_h, d.stride_w, d.pad_h, d.pad_w, d.Q) && !(o->kind == IFCBK_OP_CONV_FWD_AFFINE && o->p[5]))
                snprintf(name, cap, "conv_rows3x3<%d, %d>", d.C, d.K);
            else if (!(o->kind == IFCBK_OP_CONV_FWD_AFFINE && o->p[5]) && ifcbk_conv_ws_shape(d.dtype, d.N * d.P * d.Q, d.K, d.R * d.S * d.C))
                snprintf(name, cap, "conv_ws<%d>", ifcbk_conv_fwd_nt(d.K, d.N * d.P * d.Q));
            else
            snprintf(name, cap, "conv_igemm<unsigned short, %d, %d, %d, 0>", ifcbk_conv_fwd_nt(d.K, d.N * d.P * d.Q), wm, wm == 4 ? 3 : 2);
            break;
        }
        case IFCBK_OP_CONV_DGRAD: case IFCBK_OP_CONV_DGRAD_BNSTAT: case IFCBK_OP_CONV_DGRAD_BNSTAT_TAB: {
            const ifcbk_conv_desc& d = o->u.conv;
            int wm = ifcbk_conv_fwd_wm(d.N * d.H * d.W, d.C);
            {
                if (o->kind == IFCBK_OP_CONV_DGRAD && !(o->flags & 1) && ifcbk_conv_rows_ok(d.dtype, d.K, d.C, d.R, d.S, d.stride_h, d.stride_w, 2 - d.pad_h, 2 - d.pad_w, d.W)) {
                    snprintf(name, cap, "conv_rows3x3<%d, %d>", d.K, d.C);
                    break;
                }
                const bool s2 = d.stride_h == 2 || d.stride_w == 2;
                if (!s2 && o->kind != IFCBK_OP_CONV_DGRAD_BNSTAT_TAB && !(o->kind == IFCBK_OP_CONV_DGRAD && (o->flags & 1)) &&
                    ifcbk_conv_flat_rows(d.dtype, d.N, d.P, d.Q, d.K, d.C, d.R, d.S, d.R - 1 - d.pad_h, d.S - 1 - d.pad_w, d.H, d.W)) {
                    snprintf(name, cap, "conv_flat<%d, %d, %d, %d, %d>", d.K, d.C, d.R, d.S, o->kind == IFCBK_OP_CONV_DGRAD_BNSTAT ? 3 : 0);
                    break;
                }
                if (!s2 && o->kind != IFCBK_OP_CONV_DGRAD_BNSTAT_TAB) {
                    if (const int smt = ifcbk_conv_slab_plan(d.dtype, d.N, d.P, d.Q, d.K, d.C, d.R, d.S, d.R - 1 - d.pad_h, d.S - 1 - d.pad_w, d.H, d.W)) {
                        snprintf(name, cap, "conv_slab<%d, %d, %d, %d, 50, %d>", d.C <= 128 ? 2 : 3, smt, smt == 10 ? 4 : smt / 2, d.R * d.S,
                                 o->kind == IFCBK_OP_CONV_DGRAD_BNSTAT ? 3 : 0);
                        break;
                    }
                }
                int bmt = 0, btn = 0;
                if (!s2 && o->kind == IFCBK_OP_CONV_DGRAD && !(o->flags & 1) && ifcbk_conv_pp3_plan(d.dtype, d.N * d.H * d.W, d.C, d.R * d.S * d.K, 0)) {
                    ifcbk_conv_pp3_name(d.R * d.S * d.K, false, d.R == 1 && d.S == 1 && d.pad_h == 0 && d.pad_w == 0, name, cap);
                    break;
                }
                if (!s2 && ifcbk_conv_big_plan(d.dtype, d.N * d.H * d.W, d.C, d.R * d.S * d.K, &bmt, &btn)) {
                    snprintf(name, cap, "conv_pp2<%d, %d, %d, %d>", btn, bmt, bmt == 10 ? 4 : bmt / 2, o->kind == IFCBK_OP_CONV_DGRAD_BNSTAT ? 3 : o->kind == IFCBK_OP_CONV_DGRAD_BNSTAT_TAB ? 5 : 0);
                    break;
                }
                if (o->kind == IFCBK_OP_CONV_DGRAD && !(o->flags & 1) && !s2 && ifcbk_conv_ws_shape(d.dtype, d.N * d.H * d.W, d.C, d.R * d.S * d.K)) {
                    snprintf(name, cap, "conv_ws<%d>", ifcbk_conv_fwd_nt(d.C, d.N * d.H * d.W));
                    break;
                }
                const bool classes = d.stride_h == 2 && d.stride_w == 2 && d.R >= 2 && d.S >= 2 && d.H >= 2 && d.W >= 2;
                snprintf(name, cap, "conv_igemm<unsigned short, %d, %d, %d, %d>", ifcbk_conv_fwd_nt(d.C, classes ? d.N * ((d.H + 1) / 2) * ((d.W + 1) / 2) : d.N * d.H * d.W), wm, wm == 4 ? 3 : 2,
                         o->kind == IFCBK_OP_CONV_DGRAD_BNSTAT ? 3 : o->kind == IFCBK_OP_CONV_DGRAD_BNSTAT_TAB ? 5 : classes ? 2 : (s2 ? 1 : 0));
            }
            break;
        }
        case IFCBK_OP_CONV_WGRAD: case IFCBK_OP_CONV_WGRAD_SEG: {
            int mt = 0, cols = 0;
            ifcbk_conv_wgrad_shape(&o->u.conv, &mt, &cols);
            if (mt < 0) {
                snprintf(name, cap, "conv_wgrad_pp<%d, 0>", -mt);          // the name rocprofv3 lists: <KH, DM>
            }
            else if (o->u.conv.dtype == IFCBK_F32) snprintf(name, cap, "conv_wgrad_f32<%d>", mt);
            else if (cols == 2) snprintf(name, cap, "conv_wgrad_flat");
            else if (mt == 0) snprintf(name, cap, "conv_wgrad_stem");
            else if (cols) snprintf(name, cap, "conv_wgrad_cols<%d, 4>", mt);
            else snprintf(name, cap, "conv_wgrad_rows<%d>", mt);
            break;
        }
        case IFCBK_OP_CONV_WGRAD_GROUP: {
            const ifcbk_wgrad_item* it = (const ifcbk_wgrad_item*)o->p[0];
            const int n = (int)o->i[0];
            ifcbk_conv_desc ds[8];
            int kh = 0;
            if (it && n >= 1 && n <= 8) {
                for (int k = 0; k < n; ++k) ds[k] = it[k].d;
                if (ifcbk_conv2d_wgrad_group_info(n, ds, &kh, nullptr, nullptr) == IFCBK_OK) {
                    if (kh == 16) snprintf(name, cap, "conv_wgrad_flatg");
                    else snprintf(name, cap, "conv_wgrad_ppg<%d>", kh);
                }
            }
            break;
        }
        case IFCBK_OP_CONV_FWD_AFFINE_MAXPOOL: snprintf(name, cap, "conv_rows3x3<%d, %d, true>", o->u.conv.C, o->u.conv.K); break;
        case IFCBK_OP_STEM_U8_FWD: snprintf(name, cap, "stem_u8_fwd_kernel"); break;
        case IFCBK_OP_STEM_U8_WGRAD: snprintf(name, cap, "stem_u8_wgrad_kernel"); break;
        case IFCBK_OP_BN_APPLY: snprintf(name, cap, "bn_apply_kernel"); break;
        case IFCBK_OP_BN_BWD: snprintf(name, cap, "bn_bwd"); break;
        case IFCBK_OP_BN_BWD_PARTIALS: snprintf(name, cap, "bn_bwd(partials)"); break;
        case IFCBK_OP_BN_APPLY_MAXPOOL: snprintf(name, cap, "bn_apply_maxpool_kernel"); break;
        case IFCBK_OP_BN_BWD_MAXPOOL: snprintf(name, cap, "bn_bwd(maxpool)"); break;
        case IFCBK_OP_BN_FINALIZE: snprintf(name, cap, "bn_finalize_kernel"); break;
        case IFCBK_OP_BN_STATS: snprintf(name, cap, "bn_stats_kernel"); break;
        case IFCBK_OP_MAXPOOL_FWD: snprintf(name, cap, "maxpool_fwd_kernel"); break;
        case IFCBK_OP_MAXPOOL_BWD: snprintf(name, cap, "maxpool_bwd_kernel"); break;
        case IFCBK_OP_AVGPOOL_FWD: case IFCBK_OP_AVGPOOL_AFFINE: snprintf(name, cap, "avgpool_fwd_kernel"); break;
        case IFCBK_OP_AVGPOOL_BWD: snprintf(name, cap, "avgpool_bwd_kernel"); break;
        case IFCBK_OP_ADAM: snprintf(name, cap, "adam_kernel"); break;
        case IFCBK_OP_SGD: snprintf(name, cap, "sgd_kernel"); break;
        case IFCBK_OP_BIAS_RELU_BWD: snprintf(name, cap, "bias_relu_bwd_kernel"); break;
        case IFCBK_OP_DROPOUT: snprintf(name, cap, "dropout_apply_kernel"); break;
        case IFCBK_OP_FLATTEN_CHW: snprintf(name, cap, "flatten_chw_kernel"); break;
        case IFCBK_OP_WEIGHT_PACK: snprintf(name, cap, "weight_pack_kernel"); break;
        case IFCBK_OP_WEIGHT_PACK_MULTI: snprintf(name, cap, "weight_pack_multi_kernel"); break;
        default: break;
    }
    return IFCBK_OK;
}

extern "C" int ifcbk_op_cost(const ifcbk_op* o, double* flops, double* bytes) {
    double fl = 0, by = 0;
    switch (o->kind) {
        case IFCBK_OP_CONV_FWD: case IFCBK_OP_CONV_FWD_AFFINE: case IFCBK_OP_CONV_FWD_AFFINE_SEG: case IFCBK_OP_CONV_DGRAD: case IFCBK_OP_CONV_WGRAD:
        case IFCBK_OP_CONV_WGRAD_SEG: case IFCBK_OP_CONV_DGRAD_BNSTAT: case IFCBK_OP_CONV_DGRAD_BNSTAT_TAB: {
            const ifcbk_conv_desc& d = o->u.conv;
            double mac = (double)d.N * d.P * d.Q * d.K * d.R * d.S * d.Cw;
            fl = 2.0 * mac;
            double xin = (double)d.N * d.H * d.W * d.C * 2, yout = (double)d.N * d.P * d.Q * d.K * 2,
                   wb = (double)d.K * d.R * d.S * d.C * 2;
            by = xin + yout + wb + ((o->kind == IFCBK_OP_CONV_DGRAD_BNSTAT || o->kind == IFCBK_OP_CONV_DGRAD_BNSTAT_TAB) ? xin : 0);      // + one read of the producer's raw output
            break;
        }
        case IFCBK_OP_CONV_WGRAD_GROUP: {
            const ifcbk_wgrad_item* it = (const ifcbk_wgrad_item*)o->p[0];
            for (int k = 0; it && k < (int)o->i[0] && k < 8; ++k) {
                const ifcbk_conv_desc& d = it[k].d;
                fl += 2.0 * d.N * d.P * d.Q * d.K * d.R * d.S * d.Cw;
                by += (double)d.N * d.H * d.W * d.C * 2 + (double)d.N * d.P * d.Q * d.K * 2 + (double)d.K * d.R * d.S * d.C * 2;
            }
            break;
        }
        case IFCBK_OP_CONV_FWD_AFFINE_MAXPOOL: {
            const ifcbk_conv_desc& d = o->u.conv;
            fl = 2.0 * d.N * d.P * d.Q * d.K * d.R * d.S * d.Cw;
            by = (double)d.N * d.H * d.W * d.C * 2 + (double)d.N * ((d.P - 3) / 2 + 1) * ((d.Q - 3) / 2 + 1) * d.K * 2;
            break;
        }
        case IFCBK_OP_STEM_U8_FWD: case IFCBK_OP_STEM_U8_WGRAD: {
            const ifcbk_conv_desc& d = o->u.conv;      // the conv's own MACs (3 input channels); bytes: the u8 plane + the output / its gradient
            fl = 2.0 * d.N * d.P * d.Q * d.K * d.R * d.S * d.Cw;
            by = (double)d.N * d.H * d.W + (double)d.N * d.P * d.Q * d.K * 2;
            break;
        }
        case IFCBK_OP_BN_APPLY: by = (double)o->u.bn.M * o->u.bn.C * (2 + 2 + (o->p[3] ? 2 : 0)); break;
        case IFCBK_OP_BN_STATS: by = (double)o->u.bn.M * o->u.bn.C * 2; break;
        case IFCBK_OP_BN_BWD: by = (double)o->u.bn.M * o->u.bn.C * (2.0 * (o->u.bn.relu ? 6 : 4) + 2 + (o->p[7] ? 2 : 0)); break;
        case IFCBK_OP_BN_BWD_PARTIALS: by = (double)o->u.bn.M * o->u.bn.C * (2 + 2 + 2); break;      // x, dy in; dx out
        case IFCBK_OP_BN_APPLY_MAXPOOL: {
            const ifcbk_pool_desc& d = o->u.pool;
            by = ((double)d.N * d.H * d.W * 2 + (double)d.N * d.P * d.Q * 3) * d.C;      // x in; pooled y + u8 arg-max out
            break;
        }
        case IFCBK_OP_BN_BWD_MAXPOOL: {
            const ifcbk_pool_desc& d = o->u.pool;
            by = ((double)d.N * d.H * d.W * (2 * 2 + 2) + (double)d.N * d.P * d.Q * 3 * 2) * d.C;   // x twice, dx once; pooled grad + arg-max twice
            break;
        }
        case IFCBK_OP_MAXPOOL_FWD: case IFCBK_OP_AVGPOOL_FWD: case IFCBK_OP_MAXPOOL_BWD: case IFCBK_OP_AVGPOOL_BWD: case IFCBK_OP_AVGPOOL_AFFINE: {
            const ifcbk_pool_desc& d = o->u.pool;
            by = ((double)d.N * d.H * d.W + (double)d.N * d.P * d.Q) * d.C * 2;
            break;
        }
        case IFCBK_OP_HEAD_FWD: case IFCBK_OP_HEAD_BWD: {
            const ifcbk_head_desc& d = o->u.head;
            fl = 2.0 * d.N * d.C * d.NC * (o->kind == IFCBK_OP_HEAD_BWD ? 2 : 1);
            by = (double)d.N * d.HW * d.C * 2;
            break;
        }
        case IFCBK_OP_ADAM: by = (double)o->i[0] * 28; break;
        case IFCBK_OP_SGD: by = (double)o->i[0] * (o->p[2] ? 20 : 12); break;
        default: break;
    }
    if (flops) *flops = fl;
    if (bytes) *bytes = by;
    return 0;
}
