"""time the pooling entry points in isolation on the inception_v3 shapes (batch 256, bf16)"""
import ctypes as C
import sys
import torch
sys.path.insert(0, '.')
from ifcb_classifier_amd import _lib

ctx = _lib.Context(0)
dev = torch.device('cuda:0')
N = 256
cases = [('max', 147, 64, 2, 0), ('max', 71, 192, 2, 0), ('max', 35, 288, 2, 0), ('max', 17, 768, 2, 0),
         ('avg', 35, 192, 1, 1), ('avg', 35, 288, 1, 1), ('avg', 17, 768, 1, 1), ('avg', 8, 1280, 1, 1), ('avg', 8, 2048, 1, 1)]
for kind, H, Cc, s, p in cases:
    P = (H + 2 * p - 3) // s + 1
    x = torch.randn(N, H, H, Cc, device=dev).to(torch.bfloat16)
    y = torch.empty(N, P, P, Cc, device=dev, dtype=torch.bfloat16)
    dy = torch.randn_like(y)
    dx = torch.empty_like(x)
    arg = torch.empty(N, P, P, Cc, device=dev, dtype=torch.uint8)
    d = _lib.PoolDesc(N, H, H, Cc, Cc, 3, 3, s, s, p, p, P, P, Cc, 0)
    st = _lib.cur_stream()
    def fwd():
        if kind == 'max':
            ctx.call('ifcbk_maxpool_fwd', C.byref(d), _lib.ptr(x), _lib.ptr(y), _lib.ptr(arg), st)
        else:
            ctx.call('ifcbk_avgpool_fwd', C.byref(d), _lib.ptr(x), _lib.ptr(y), st)
    def bwd():
        if kind == 'max':
            ctx.call('ifcbk_maxpool_bwd', C.byref(d), _lib.ptr(dy), _lib.ptr(arg), _lib.ptr(dx), 0, st)
        else:
            ctx.call('ifcbk_avgpool_bwd', C.byref(d), _lib.ptr(dy), _lib.ptr(dx), 0, st)
    for name, fn in (('fwd', fwd), ('bwd', bwd)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        mb = (x.numel() + y.numel()) * 2 / 1e6 + (arg.numel() / 1e6 if kind == 'max' else 0)
        print('%s %s H=%d C=%d: %7.1f us  %6.0f GB/s' % (kind, name, H, Cc, us, mb / us * 1e3))
