"""dev tool: which activation of a RUN batch beyond the engine's verified capacity first differs from the same images run 256 at a
time?  (IFCBK_DEV_INDEX_GIB lifts the capacity cap; every activation of the eval forward keeps its own buffer, so exact per-image
checksums of all of them can be compared after the fact.)  usage: IFCBK_DEV_INDEX_GIB=8 python scripts/big_batch_bisect.py 1536"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ifcb_classifier_amd import graph
from ifcb_classifier_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
use_graph = os.environ.get('BISECT_GRAPH', '1') != '0'
net = graph.build('inception_v3', 100, pretrained=False)
eng = Engine(net, device=0, max_batch=B, train_batch=1)
eng.init_weights(seed=1234)
rois, _ = bench.synth_rois(B, 77, eng.dev)
print('engine up: window', eng.window_batch, 'capacity', eng.max_batch, flush=True)


def run(i0, n):
    eng.load_rois(rois['pixels'], rois['offs'][i0:i0 + n], rois['hs'][i0:i0 + n], rois['ws'][i0:i0 + n], rois['max_h'], rois['max_w'])
    p = eng.forward_eval(n)
    eng.run(p.softmax)
    torch.cuda.synchronize()


def sums(n):
    out = {}
    for bid, t in eng.act.items():
        if t.dtype != torch.bfloat16:
            continue
        acc = torch.empty(n, dtype=torch.int64, device=t.device)
        for j in range(0, n, 64):
            m = min(64, n - j)
            acc[j:j + m] = t[j:j + m].contiguous().view(torch.int16).to(torch.int32).view(m, -1).sum(1, dtype=torch.int64)
        out[bid] = acc.cpu()
    out['probs'] = eng.probs[:n].double().sum(1).cpu()
    return out


run(0, B)
print('whole ran', flush=True)
whole = sums(B)
print('whole summed', flush=True)
bad = {}
for i0 in range(0, B, 256):
    n = min(256, B - i0)
    run(i0, n)
    part = sums(n)
    for k, v in part.items():
        d = (whole[k][i0:i0 + n] != v).nonzero().flatten()
        if len(d):
            bad.setdefault(k, []).extend((d + i0).tolist())
names = {b.id: b.name for b in net.bufs}
order = [b.id for b in net.bufs if b.id in bad] + [k for k in bad if k not in names]
print('batch', B, ':', len(bad), 'of', len(whole), 'tensors differ')
for k in order[:40]:
    v = bad[k]
    t = eng.act.get(k)
    print('  %-34s %-22s images differing: %d, first %d, last %d' % (names.get(k, k), tuple(t.shape[1:]) if t is not None else '', len(v), v[0], v[-1]))
