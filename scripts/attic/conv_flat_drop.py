"""dev tool: where a conv_flat block's time goes.  Needs a library built with -DIFCBK_EXPERIMENT_FLAT (IFCBK_LIB=...): IFCBK_DEBUG_DROP
bits 1 = no image load, 2 = no filter DMA, 4 = no fragment reads / MFMAs, 8 = no epilogue (results are wrong by design)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ifcb_classifier_amd import _lib
from ifcb_classifier_amd._lib import ConvDesc

LAYERS = {'5b_5x5': (256, 48, 35, 35, 64, 5, 5, 2, 2), '5c_3x3a': (256, 64, 35, 35, 96, 3, 3, 1, 1), '5c_3x3b': (256, 96, 35, 35, 96, 3, 3, 1, 1)}
ctx = _lib.Context(0)
ctx.reserve(1 << 30)
st = _lib.cur_stream()
os.environ['IFCBK_CONV_FLAT'] = '2'
for name, (N, Cc, H, W, K, R, S, ph, pw) in LAYERS.items():
    N = int(os.environ.get('CONV_LAYERS_N', N))
    d = ConvDesc(N, H, W, Cc, Cc, K, R, S, 1, 1, ph, pw, H, W, K, Cc, 0)
    x = torch.randn(N, H, W, Cc, device='cuda').bfloat16()
    w = (torch.randn(K, R, S, Cc, device='cuda') * 0.05).bfloat16()
    y = torch.empty(N, H, W, K, device='cuda', dtype=torch.bfloat16)
    mb = ctx.lib.ifcbk_conv2d_fwd_mblocks(C.byref(d))
    part = torch.empty(mb, 2, K, device='cuda')
    run = lambda: ctx.call('ifcbk_conv2d_fwd', C.byref(d), _lib.ptr(x), _lib.ptr(w), _lib.ptr(y), _lib.ptr(part), st)
    res = {}
    variants = [int(v, 0) for v in os.environ.get('DROPS', '0,8,4,12,7,15,2,1,3').split(',')]
    for r in range(5):
        for v in variants:
            os.environ['IFCBK_DEBUG_DROP'] = str(v)
            run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) / 5 * 1e3)
    print(name, mb, 'segments |', ' | '.join('drop %4d: %5.1f us' % (v, min(res[v])) for v in variants), flush=True)
