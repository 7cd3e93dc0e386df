"""dev tool: is a RUN batch far beyond the descriptor window (2048 images: three image groups per convolution, tensors of more than
2^31 elements) the same as its 256-image parts, bit for bit?  ROI path + hipGraph-replayed eval forward + softmax, as bench.py's RUN leg."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from ifcb_classifier_amd import graph
from ifcb_classifier_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
eng = Engine(graph.build('inception_v3', 100, pretrained=False), device=0, max_batch=B, train_batch=1)
eng.init_weights(seed=1234)
rois, _ = bench.synth_rois(B, 77, eng.dev)


def run(i0, n):
    eng.load_rois(rois['pixels'], rois['offs'][i0:i0 + n], rois['hs'][i0:i0 + n], rois['ws'][i0:i0 + n], rois['max_h'], rois['max_w'])
    p = eng.forward_eval(n)
    eng.run(p.softmax)
    torch.cuda.synchronize()
    return eng.probs[:n].clone()


run(0, B)
whole = run(0, B)
print('whole done', flush=True)
parts = torch.cat([run(i, min(256, B - i)) for i in range(0, B, 256)], 0)
print('batch', B, 'finite', bool(torch.isfinite(whole).all()), 'equal to 256-image parts', bool(torch.equal(whole, parts)),
      'max abs diff %.3e' % float((whole - parts).abs().max()), flush=True)
