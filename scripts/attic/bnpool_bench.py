"""time bn_bwd_maxpool / bn_apply_maxpool in isolation on the two inception_v3 stem shapes (batch 256, bf16);
run under `rocprofv3 --kernel-trace --stats` to split the backward into its reduce and dx kernels"""
import ctypes as C
import sys
import torch
sys.path.insert(0, '.')
from ifcb_classifier_amd import _lib

ctx = _lib.Context(0)
dev = torch.device('cuda:0')
N = 256
ctx.reserve(256 << 20)
for H, Cc in ((147, 64), (71, 192)):
    P = (H - 3) // 2 + 1
    x = torch.randn(N, H, H, Cc, device=dev).to(torch.bfloat16)
    yp = torch.empty(N, P, P, Cc, device=dev, dtype=torch.bfloat16)
    dp = torch.randn_like(yp)
    dx = torch.empty_like(x)
    arg = torch.empty(N, P, P, Cc, device=dev, dtype=torch.uint8)
    gamma = torch.rand(Cc, device=dev) + 0.5
    mean = torch.zeros(Cc, device=dev)
    invstd = torch.ones(Cc, device=dev)
    scale, shift = gamma.clone(), torch.zeros(Cc, device=dev)
    dg, db = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
    d = _lib.PoolDesc(N, H, H, Cc, Cc, 3, 3, 2, 2, 0, 0, P, P, Cc, 0)
    st = _lib.cur_stream()
    p = _lib.ptr
    def fwd():
        ctx.call('ifcbk_bn_apply_maxpool', C.byref(d), p(x), p(scale), p(shift), 1, p(yp), p(arg), st)
    def bwd():
        ctx.call('ifcbk_bn_bwd_maxpool', C.byref(d), p(x), p(dp), p(arg), p(gamma), p(mean), p(invstd), p(scale), p(shift), 1,
                 p(dx), Cc, p(dg), p(db), 0, st)
    for name, fn, mb in (('apply+pool', fwd, (x.numel() * 2 + yp.numel() * 3) / 1e6),
                         ('bn_bwd(pool)', bwd, (x.numel() * 6 + yp.numel() * 6) / 1e6)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print('%s H=%d C=%d: %7.1f us  %6.0f GB/s' % (name, H, Cc, us, mb / us * 1e3), flush=True)
