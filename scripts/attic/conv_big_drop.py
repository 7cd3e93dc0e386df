"""dev tool (timing only): conv_big with the pixel and/or filter operand's loads dropped (IFCBK_DEBUG_DROP=a|b|ab)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ifcb_classifier_amd import _lib
from ifcb_classifier_amd._lib import ConvDesc
L = {'k64_7x1': (256, 64, 17, 17, 192, 7, 1, 1, 1, 3, 0), 'k64_1x1': (256, 64, 17, 17, 192, 1, 1, 1, 1, 0, 0), '6e_7x1': (256, 192, 17, 17, 192, 7, 1, 1, 1, 3, 0), '6e_1x1g': (256, 768, 17, 17, 768, 1, 1, 1, 1, 0, 0),
     '4a_3x3': (256, 80, 73, 73, 192, 3, 3, 1, 1, 0, 0), '5c_3x3b': (256, 96, 35, 35, 96, 3, 3, 1, 1, 1, 1)}
ctx = _lib.Context(0); ctx.reserve(1 << 30); st = _lib.cur_stream()
for name, (N, Cc, H, W, K, R, S, sh, sw, ph, pw) in L.items():
    P = (H + 2 * ph - R) // sh + 1; Q = (W + 2 * pw - S) // sw + 1
    d = ConvDesc(N, H, W, Cc, Cc, K, R, S, sh, sw, ph, pw, P, Q, K, Cc, 0)
    x = torch.randn(N, H, W, Cc, device='cuda').bfloat16(); w = (torch.randn(K, R, S, Cc, device='cuda') * 0.05).bfloat16()
    y = torch.empty(N, P, Q, K, device='cuda', dtype=torch.bfloat16)
    part = torch.empty(4096, 2, K, device='cuda')
    flops = 2.0 * N * P * Q * K * R * S * Cc
    os.environ['IFCBK_CONV_BIG'] = '2'
    res = []
    for drop in (sys.argv[1].split(',') if len(sys.argv) > 1 else ('', 'e', 'mrd', 'mrde', 'rd', 'rde', 'm', 'me')):
        os.environ['IFCBK_DEBUG_DROP'] = drop
        run = lambda: ctx.call('ifcbk_conv2d_fwd', C.byref(d), _lib.ptr(x), _lib.ptr(w), _lib.ptr(y), _lib.ptr(part), st)
        run(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): run()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 3)
        res.append('%s %.1f us' % (drop if drop not in ('', 'x') else 'full', best * 1e3))
    print(name, ' | '.join(res), '| full = %.0f TF/s' % (flops / float(res[0].split()[1]) / 1e6), flush=True)
