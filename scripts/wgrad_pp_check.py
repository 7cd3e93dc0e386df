"""dev tool: wide-tile ping-pong weight gradient (conv_wgrad_pp) against conv_wgrad_rows on inception_v3 shapes at batch 256,
in one process (IFCBK_WGRAD_PP is read per call): correctness against torch's fp32 GPU weight gradient, interleaved timing.
    python scripts/wgrad_pp_check.py [layers|all] [reps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ifcb_classifier_amd import _lib
from ifcb_classifier_amd._lib import ConvDesc

LAYERS = {
    '6e_7x1':    (256, 192, 17, 17, 192, 7, 1, 1, 1, 3, 0),
    '6e_1x7':    (256, 192, 17, 17, 192, 1, 7, 1, 1, 0, 3),
    '6c_1x7':    (256, 160, 17, 17, 160, 1, 7, 1, 1, 0, 3),
    '6c_7x1o':   (256, 160, 17, 17, 192, 7, 1, 1, 1, 3, 0),
    '6b_1x7':    (256, 128, 17, 17, 128, 1, 7, 1, 1, 0, 3),
    '6e_1x1g':   (256, 768, 17, 17, 768, 1, 1, 1, 1, 0, 0),
    '6b_1x1g':   (256, 768, 17, 17, 640, 1, 1, 1, 1, 0, 0),
    '6a_3x3s2':  (256, 288, 35, 35, 384, 3, 3, 2, 2, 0, 0),
    '7a_3x3s2':  (256, 192, 17, 17, 320, 3, 3, 2, 2, 0, 0),
    '5c_1x1g':   (256, 256, 35, 35, 240, 1, 1, 1, 1, 0, 0),
    '5c_3x3b':   (256, 96, 35, 35, 96, 3, 3, 1, 1, 1, 1),
    '7b_3x3':    (256, 448, 8, 8, 384, 3, 3, 1, 1, 1, 1),
    '7b_1x3':    (256, 384, 8, 8, 384, 1, 3, 1, 1, 0, 1),
    '7c_1x1g':   (256, 2048, 8, 8, 1344, 1, 1, 1, 1, 0, 0),
}
which = sys.argv[1].split(',') if len(sys.argv) > 1 and sys.argv[1] != 'all' else list(LAYERS)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = _lib.Context(0)
ctx.reserve(2 << 30)
st = _lib.cur_stream()


def rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def kname(d):
    op = _lib.Op()
    op.kind = _lib.OP_CONV_WGRAD
    op.u.conv = d
    buf = C.create_string_buffer(96)
    ctx.lib.ifcbk_op_kernel(C.byref(op), buf, 96)
    return buf.value.decode()


tot = [0.0, 0.0, 0.0]
bad = 0
for name in which:
    N, Cc, H, W, K, R, S, sh, sw, ph, pw = LAYERS[name]
    P = (H + 2 * ph - R) // sh + 1
    Q = (W + 2 * pw - S) // sw + 1
    d = ConvDesc(N, H, W, Cc, Cc, K, R, S, sh, sw, ph, pw, P, Q, K, Cc, 0)
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(N, H, W, Cc, device='cuda', generator=g).bfloat16()
    dy = (torch.randn(N, P, Q, K, device='cuda', generator=g) * 0.1).bfloat16()
    ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (K, Cc, R, S), dy.float().permute(0, 3, 1, 2), (sh, sw), (ph, pw)).permute(0, 2, 3, 1)
    flops = 2.0 * N * P * Q * K * R * S * Cc
    outs = {}
    for pp in ('0', '1'):
        os.environ['IFCBK_WGRAD_PP'] = pp
        kn = kname(d)
        dw = torch.full((K, R, S, Cc), float('nan'), device='cuda')
        run = lambda dw=dw: ctx.call('ifcbk_conv2d_wgrad', C.byref(d), _lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), 0, st)
        run(); torch.cuda.synchronize()
        outs[pp] = (kn, run, rel(dw, ref), dw)
    ms = {'0': [], '1': []}
    for r in range(reps):
        for pp in ('0', '1'):
            os.environ['IFCBK_WGRAD_PP'] = pp
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                outs[pp][1]()
            e1.record(); torch.cuda.synchronize()
            ms[pp].append(e0.elapsed_time(e1) / 3)
    m0, m1 = min(ms['0']), min(ms['1'])
    cross = rel(outs['1'][3], outs['0'][3])
    ok = outs['1'][2] < 2e-3 or cross < 1e-4
    bad += 0 if ok else 1
    same = outs['0'][0] == outs['1'][0]
    print('%-9s old %-22s %7.3f ms %5.0f TF | pp %-18s %7.3f ms %5.0f TF x%.2f | err old %.1e new %.1e cross %.1e %s'
          % (name, outs['0'][0], m0, flops / m0 / 1e9, '(same)' if same else outs['1'][0], m1, flops / m1 / 1e9, m0 / m1,
             outs['0'][2], outs['1'][2], cross, 'OK' if ok else 'BAD'), flush=True)
    tot[0] += m0; tot[1] += m1; tot[2] += flops
print('TOTAL old %.3f ms %.0f TF/s   new %.3f ms %.0f TF/s' % (tot[0], tot[2] / tot[0] / 1e9, tot[1], tot[2] / tot[1] / 1e9))
sys.exit(1 if bad else 0)
