"""dev tool (timing only, WRONG training results): what would the overlapped step gain if a class of ops cost nothing?  The ops
of the class are turned into 4-byte memsets of their own output inside the live step program (lane and wait flags stay), the step
is timed before and after on the same engine.  Classes: wgrad35 (weight gradients of the 35x35 stage), wgrad17, wgrad8, wgradstem,
bnfin (bn_finalize + bn_bwd partials' finalize are inside their ops: only the forward finalize op can be dropped), all_wgrad.
    python scripts/ceiling_drop_ops.py wgrad35 wgrad17 ..."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ifcb_classifier_amd import graph, _lib
from ifcb_classifier_amd.engine import Engine

B = 256
WG = (_lib.OP_CONV_WGRAD, _lib.OP_CONV_WGRAD_SEG)


def build():
    eng = Engine(graph.build('inception_v3', 100, pretrained=False), device=0, max_batch=B)
    eng.init_weights(seed=1234)
    rois, _ = bench.synth_rois(B, 1234, eng.dev)
    eng.target[:B].copy_(torch.randint(0, 100, (B,), generator=torch.Generator().manual_seed(99)))
    return eng, rois


def timeit(eng, rois, steps=40, warm=10):
    for _ in range(warm):
        eng.load_rois(**rois)
        eng.train_step(B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.load_rois(**rois)
        eng.train_step(B)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


def match(o, cls):
    k, d = o.kind, o.u.conv
    if cls == 'all_wgrad':
        return k in WG or k in (_lib.OP_CONV_WGRAD_GROUP, _lib.OP_STEM_U8_WGRAD)
    if cls == 'wgrad35':
        return k in WG and d.P == 35
    if cls == 'wgrad17':
        return (k in WG and d.P == 17) or (k == _lib.OP_CONV_WGRAD_GROUP)
    if cls == 'wgrad8':
        return k in WG and d.P == 8
    if cls == 'wgradstem':
        return (k in WG and d.P > 35) or k == _lib.OP_STEM_U8_WGRAD
    if cls == 'bnfin':
        return k == _lib.OP_BN_FINALIZE
    if cls == 'dgrad':
        return k in (_lib.OP_CONV_DGRAD, _lib.OP_CONV_DGRAD_BNSTAT, _lib.OP_CONV_DGRAD_BNSTAT_TAB)
    if cls == 'bnbwd':
        return k in (_lib.OP_BN_BWD, _lib.OP_BN_BWD_PARTIALS, _lib.OP_BN_BWD_MAXPOOL)
    if cls == 'bnapply':
        return k in (_lib.OP_BN_APPLY, _lib.OP_BN_APPLY_MAXPOOL)
    if cls in ('dgrad35', 'dgrad17', 'dgrad8', 'dgradstem'):
        if k not in (_lib.OP_CONV_DGRAD, _lib.OP_CONV_DGRAD_BNSTAT, _lib.OP_CONV_DGRAD_BNSTAT_TAB):
            return False
        hw = d.H                        # the tensor the input gradient is written for
        return {'dgrad35': hw == 35, 'dgrad17': hw == 17, 'dgrad8': hw == 8, 'dgradstem': hw > 35}[cls]
    if cls in ('fwd35', 'fwd17', 'fwd8', 'fwdstem'):
        if k not in (_lib.OP_CONV_FWD, _lib.OP_STEM_U8_FWD):
            return False
        hw = d.P
        return {'fwd35': hw == 35, 'fwd17': hw == 17, 'fwd8': hw == 8, 'fwdstem': hw > 35}[cls]
    if cls == 'pools':
        return k in (_lib.OP_MAXPOOL_FWD, _lib.OP_MAXPOOL_BWD, _lib.OP_AVGPOOL_FWD, _lib.OP_AVGPOOL_BWD)
    if cls == 'fwdconv':
        return k in (_lib.OP_CONV_FWD, _lib.OP_STEM_U8_FWD)
    raise SystemExit('unknown class ' + cls)


def drop(prog, cls):
    n = 0
    for i in range(prog.n):
        o = prog.arr[i]
        if not match(o, cls):
            continue
        if o.kind == _lib.OP_CONV_WGRAD_GROUP:
            tgt = o.p[1]
        elif o.kind == _lib.OP_STEM_U8_WGRAD:
            tgt = o.p[3]
        elif o.kind in WG:
            tgt = o.p[2]
        else:
            tgt = next(v for v in o.p if v)
        o.kind = _lib.OP_MEMSET
        o.p[0] = tgt
        o.i[0], o.i[1] = 4, 0
        n += 1
    return n


for cls in sys.argv[1:] or ['wgrad35']:
    eng, rois = build()
    eng.load_rois(**rois)
    pl = eng.train_step(B)                # the plan of the input kind load_rois selected (u8 plane)
    base = timeit(eng, rois)
    n = drop(pl.step, cls)
    t = timeit(eng, rois)
    print('%-10s %3d ops dropped: %.3f -> %.3f ms/step (%.3f)' % (cls, n, base, t, base - t), flush=True)
    del eng, pl
    torch.cuda.empty_cache()
