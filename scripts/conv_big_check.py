"""dev tool: the wide-tile ping-pong kernel (conv_big) against conv_igemm / conv_ws on the inception_v3 layer shapes at batch
256, in ONE process (IFCBK_CONV_BIG is read per launch): correctness of forward (+ BatchNorm partial sums) and input gradient
against torch's fp32 GPU convolution of the same bf16 operands, then interleaved timing rounds.
    python scripts/conv_big_check.py [layers|all] [reps]
CHECK_SWITCH=IFCBK_CONV_SLAB compares the pixel-slab kernel (conv_slab.hip) with what the dispatcher picks without it."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from ifcb_classifier_amd import _lib
from ifcb_classifier_amd._lib import ConvDesc

LAYERS = {
    # name: N, C, H, W, K, R, S, sh, sw, ph, pw
    '4a_3x3':    (256, 80, 73, 73, 192, 3, 3, 1, 1, 0, 0),
    '6a_3x3s2':  (256, 288, 35, 35, 384, 3, 3, 2, 2, 0, 0),
    '6b_1x7':    (256, 128, 17, 17, 128, 1, 7, 1, 1, 0, 3),
    '6b_7x1o':   (256, 128, 17, 17, 192, 7, 1, 1, 1, 3, 0),
    '6c_1x7':    (256, 160, 17, 17, 160, 1, 7, 1, 1, 0, 3),
    '6c_7x1o':   (256, 160, 17, 17, 192, 7, 1, 1, 1, 3, 0),
    '6e_7x1':    (256, 192, 17, 17, 192, 7, 1, 1, 1, 3, 0),
    '6e_1x7':    (256, 192, 17, 17, 192, 1, 7, 1, 1, 0, 3),
    '6c_7x1':    (256, 160, 17, 17, 160, 7, 1, 1, 1, 3, 0),
    '6b_7x1':    (256, 128, 17, 17, 128, 7, 1, 1, 1, 3, 0),
    '6b_1x1g':   (256, 768, 17, 17, 640, 1, 1, 1, 1, 0, 0),
    '6e_1x1g':   (256, 768, 17, 17, 768, 1, 1, 1, 1, 0, 0),
    '7a_1x1g':   (256, 768, 17, 17, 384, 1, 1, 1, 1, 0, 0),
    '5b_5x5':    (256, 48, 35, 35, 64, 5, 5, 1, 1, 2, 2),
    '5c_3x3b':   (256, 96, 35, 35, 96, 3, 3, 1, 1, 1, 1),
    '5c_3x3a':   (256, 64, 35, 35, 96, 3, 3, 1, 1, 1, 1),
    '5c_1x1g':   (256, 256, 35, 35, 240, 1, 1, 1, 1, 0, 0),
    '7b_3x3':    (256, 448, 8, 8, 384, 3, 3, 1, 1, 1, 1),
    '7c_1x1g':   (256, 2048, 8, 8, 1344, 1, 1, 1, 1, 0, 0),
}
which = sys.argv[1].split(',') if len(sys.argv) > 1 and sys.argv[1] != 'all' else list(LAYERS)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
NOVR = int(os.environ.get('CONV_LAYERS_N', '0'))
SW = os.environ.get('CHECK_SWITCH', 'IFCBK_CONV_BIG')
ctx = _lib.Context(0)
ctx.reserve(1 << 30)
st = _lib.cur_stream()


def rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def kname(d, kind):
    op = _lib.Op()
    op.kind = kind
    op.u.conv = d
    buf = C.create_string_buffer(96)
    ctx.lib.ifcbk_op_kernel(C.byref(op), buf, 96)
    return buf.value.decode()


tot = {}
bad = 0
for name in which:
    N, Cc, H, W, K, R, S, sh, sw, ph, pw = LAYERS[name]
    if NOVR:
        N = NOVR
    P = (H + 2 * ph - R) // sh + 1
    Q = (W + 2 * pw - S) // sw + 1
    d = ConvDesc(N, H, W, Cc, Cc, K, R, S, sh, sw, ph, pw, P, Q, K, Cc, 0)
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(N, H, W, Cc, device='cuda', generator=g).bfloat16()
    w = (torch.randn(K, R, S, Cc, device='cuda', generator=g) * (1.0 / (R * S * Cc) ** 0.5)).bfloat16()
    wT = w.permute(3, 1, 2, 0).flip(1, 2).contiguous()          # [C][R'][S'][K]: flipped, transposed filter of the input gradient
    dy = torch.randn(N, P, Q, K, device='cuda', generator=g).bfloat16()
    flops = 2.0 * N * P * Q * K * R * S * Cc
    ref_y = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), None, (sh, sw), (ph, pw)).permute(0, 2, 3, 1)
    modes = ['fwd'] + (['dgrad'] if sh == 1 else [])
    if 'dgrad' in modes:
        ref_dx = torch.nn.grad.conv2d_input((N, Cc, H, W), w.float().permute(0, 3, 1, 2), dy.float().permute(0, 3, 1, 2), (sh, sw), (ph, pw)).permute(0, 2, 3, 1)
    res = {}
    for mode in modes:
        outs = {}
        for big in ('0', '1'):
            os.environ[SW] = big
            kn = kname(d, _lib.OP_CONV_FWD if mode == 'fwd' else _lib.OP_CONV_DGRAD)
            if mode == 'fwd':
                y = torch.full((N, P, Q, K), float('nan'), device='cuda', dtype=torch.bfloat16)
                mb = ctx.lib.ifcbk_conv2d_fwd_mblocks(C.byref(d))
                part = torch.full((mb, 2, K), float('nan'), device='cuda')
                run = lambda y=y, part=part: ctx.call('ifcbk_conv2d_fwd', C.byref(d), _lib.ptr(x), _lib.ptr(w), _lib.ptr(y), _lib.ptr(part), st)
                run(); torch.cuda.synchronize()
                e = rel(y.float(), ref_y)
                s1 = part[:, 0].double().sum(0)
                es = rel(s1, y.float().double().sum((0, 1, 2)))
                outs[big] = (kn, run, e, es, y)
            else:
                dx = torch.full((N, H, W, Cc), float('nan'), device='cuda', dtype=torch.bfloat16)
                run = lambda dx=dx: ctx.call('ifcbk_conv2d_dgrad', C.byref(d), _lib.ptr(dy), _lib.ptr(wT), _lib.ptr(dx), 0, st)
                run(); torch.cuda.synchronize()
                outs[big] = (kn, run, rel(dx.float(), ref_dx), 0.0, dx)
        # interleaved timing rounds
        ms = {'0': [], '1': []}
        for r in range(reps):
            for big in ('0', '1'):
                os.environ[SW] = big
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    outs[big][1]()
                e1.record(); torch.cuda.synchronize()
                ms[big].append(e0.elapsed_time(e1) / 3)
        m0, m1 = min(ms['0']), min(ms['1'])
        same = outs['0'][0] == outs['1'][0]
        cross = rel(outs['1'][4].float(), outs['0'][4].float())            # new vs old kernel directly
        if not (outs['1'][2] < 5e-3) and cross < 3e-3 and not (outs['0'][2] < 5e-3):
            outs['1'] = outs['1'][:2] + (cross,) + outs['1'][3:]              # the torch reference is off for this shape: both kernels agree
        print('%-9s %-5s old %-38s %7.3f ms %5.0f TF | pp2 %-18s %7.3f ms %5.0f TF x%.2f | err old %.1e new %.1e stat %.1e %s'
              % (name, mode, outs['0'][0], m0, flops / m0 / 1e9, '(same)' if same else outs['1'][0], m1, flops / m1 / 1e9, m0 / m1,
                 outs['0'][2], outs['1'][2], outs['1'][3], 'OK' if outs['1'][2] < 5e-3 and outs['1'][3] < 1e-4 else 'BAD'), flush=True)
        bad += 0 if (outs['1'][2] < 5e-3 and outs['1'][3] < 1e-4) else 1
        t = tot.setdefault(mode, [0.0, 0.0, 0.0])
        t[0] += m0; t[1] += m1; t[2] += flops
for m, t in tot.items():
    print('TOTAL %-5s old %.3f ms %.0f TF/s   new %.3f ms %.0f TF/s' % (m, t[0], t[2] / t[0] / 1e9, t[1], t[2] / t[1] / 1e9))
sys.exit(1 if bad else 0)
