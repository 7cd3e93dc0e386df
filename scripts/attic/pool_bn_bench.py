"""dev tool: the pool-fused BatchNorm kernels of the stem alone, at batch 256 (Conv2d_2b -> maxpool1, Conv2d_4a -> maxpool2):
microseconds and TB/s of their minimum bytes.  With rocprofv3 --kernel-trace --stats around it the three kernels of the backward
(reduce, finalize, dx) show up separately."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ifcb_classifier_amd import _lib

ctx = _lib.Context(0)
ctx.reserve(1 << 28)
st = _lib.cur_stream()
for name, (N, H, W, Cc) in {'2b+maxpool1': (256, 147, 147, 64), '4a+maxpool2': (256, 71, 71, 192)}.items():
    P, Q = (H - 3) // 2 + 1, (W - 3) // 2 + 1
    g = torch.Generator(device='cuda').manual_seed(1)
    raw = torch.randn(N, H, W, Cc, generator=g, device='cuda').bfloat16()
    scale, shift = torch.rand(Cc, device='cuda') + 0.5, torch.randn(Cc, device='cuda') * 0.3
    gamma, mean, invstd = torch.rand(Cc, device='cuda') + 0.5, torch.randn(Cc, device='cuda') * 0.2, torch.rand(Cc, device='cuda') + 0.5
    dpool = torch.randn(N, P, Q, Cc, generator=g, device='cuda').bfloat16()
    yp = torch.empty(N, P, Q, Cc, dtype=torch.bfloat16, device='cuda')
    arg = torch.empty(N, P, Q, Cc, dtype=torch.uint8, device='cuda')
    dx = torch.empty_like(raw)
    dg, db = torch.zeros(Cc, device='cuda'), torch.zeros(Cc, device='cuda')
    pd = _lib.PoolDesc(N, H, W, Cc, Cc, 3, 3, 2, 2, 0, 0, P, Q, Cc, 0)
    fwd = lambda: ctx.call('ifcbk_bn_apply_maxpool', C.byref(pd), _lib.ptr(raw), _lib.ptr(scale), _lib.ptr(shift), 1, _lib.ptr(yp), _lib.ptr(arg), st)
    bwd = lambda: ctx.call('ifcbk_bn_bwd_maxpool', C.byref(pd), _lib.ptr(raw), _lib.ptr(dpool), _lib.ptr(arg), _lib.ptr(gamma), _lib.ptr(mean),
                           _lib.ptr(invstd), _lib.ptr(scale), _lib.ptr(shift), 1, _lib.ptr(dx), Cc, _lib.ptr(dg), _lib.ptr(db), 0, st)
    nin, nout = raw.numel(), yp.numel()
    for what, fn, nbytes in (('apply_maxpool', fwd, 2 * nin + 3 * nout), ('bwd_maxpool', bwd, 2 * (2 * nin + 3 * nout) + 2 * nin)):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 3)
        print('%-12s %-14s %7.1f us  %5.2f TB/s of %6.1f MB' % (name, what, best * 1e3, nbytes / best / 1e9, nbytes / 1e6), flush=True)
