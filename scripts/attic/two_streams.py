"""experiment: do two independent kernels of the step overlap when issued on two streams? (wave-quantization / tail recovery)"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ifcb_classifier_amd import _lib
from ifcb_classifier_amd._lib import ConvDesc, BnDesc

ctx = _lib.Context(0)
ctx.reserve(1 << 28)
N, Cc, H, W, K, R, S, ph, pw = 256, 192, 17, 17, 192, 7, 1, 3, 0
d = ConvDesc(N, H, W, Cc, Cc, K, R, S, 1, 1, ph, pw, H, W, K, Cc, 0)
mk = lambda *s: torch.randn(*s, device='cuda').bfloat16()
xs = [mk(N, H, W, Cc) for _ in range(2)]
ws = [(torch.randn(K, R, S, Cc, device='cuda') * 0.05).bfloat16() for _ in range(2)]
ys = [torch.empty(N, H, W, K, device='cuda', dtype=torch.bfloat16) for _ in range(2)]
parts = [torch.empty(ctx.lib.ifcbk_conv2d_fwd_mblocks(C.byref(d)), 2, K, device='cuda') for _ in range(2)]
bd = BnDesc(N * H * W, K, K, K, 1, 0, 1e-3, 0.1)
sc, sh = torch.rand(K, device='cuda'), torch.rand(K, device='cuda')
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
h = lambda s: C.c_void_p(s.cuda_stream)

def conv(i, s):
    ctx.call('ifcbk_conv2d_fwd', C.byref(d), _lib.ptr(xs[i]), _lib.ptr(ws[i]), _lib.ptr(ys[i]), _lib.ptr(parts[i]), h(s))

def bn(i, s):
    ctx.call('ifcbk_bn_apply', C.byref(bd), _lib.ptr(ys[i]), _lib.ptr(sc), _lib.ptr(sh), None, 0, _lib.ptr(xs[i]), h(s))

def timeit(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(s0)
    for _ in range(reps):
        fn()
    s0.wait_stream(s1)
    e1.record(s0)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

def seq():
    conv(0, s0); bn(0, s0); conv(1, s0); bn(1, s0)

def par():
    s1.wait_stream(s0)
    conv(0, s0); bn(0, s0)
    conv(1, s1); bn(1, s1)
    s0.wait_stream(s1)

print('sequential  (conv+bn) x2: %.1f us' % timeit(seq))
print('two streams (conv+bn) x2: %.1f us' % timeit(par))
