"""dev tool: the persistent wide-tile kernel (conv_pp3) against what the dispatcher picks without it (conv_pp2 / conv_igemm / conv_ws),
in ONE process (IFCBK_CONV_PP3 is read per launch), on inception_v3 layer shapes: forward + BatchNorm sums, the eval affine epilogue,
the first-writer input gradient; interleaved timing rounds.  CONV_LAYERS_N overrides the batch (1024 = the RUN headline batch).
    python scripts/conv_pp3_check.py [layers|all] [reps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ifcb_classifier_amd import _lib
from ifcb_classifier_amd._lib import ConvDesc

LAYERS = {
    # name: N, C, H, W, K, R, S, sh, sw, ph, pw
    '4a_3x3':    (256, 80, 73, 73, 192, 3, 3, 1, 1, 0, 0),
    '6a_3x3s2':  (256, 288, 35, 35, 384, 3, 3, 2, 2, 0, 0),
    '6e_1x1g':   (256, 768, 17, 17, 768, 1, 1, 1, 1, 0, 0),
    '7a_1x1g':   (256, 768, 17, 17, 384, 1, 1, 1, 1, 0, 0),
    '6e_7x1':    (256, 192, 17, 17, 192, 7, 1, 1, 1, 3, 0),
    '6e_1x7':    (256, 192, 17, 17, 192, 1, 7, 1, 1, 0, 3),
    '6c_1x7':    (256, 160, 17, 17, 160, 1, 7, 1, 1, 0, 3),
    '7b_3x3':    (256, 448, 8, 8, 384, 3, 3, 1, 1, 1, 1),
    '7b_1x3':    (256, 384, 8, 8, 384, 1, 3, 1, 1, 0, 1),
    '7c_1x1a':   (256, 2048, 8, 8, 384, 1, 1, 1, 1, 0, 0),
}
which = sys.argv[1].split(',') if len(sys.argv) > 1 and sys.argv[1] != 'all' else list(LAYERS)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
NOVR = int(os.environ.get('CONV_LAYERS_N', '0'))
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
    wT = w.permute(3, 1, 2, 0).flip(1, 2).contiguous()
    dy = torch.randn(N, P, Q, K, device='cuda', generator=g).bfloat16()
    scale = (torch.rand(K, device='cuda', generator=g) + 0.5)
    shift = torch.randn(K, device='cuda', generator=g) * 0.3
    flops = 2.0 * N * P * Q * K * R * S * Cc
    for mode in ['fwd', 'affine'] + (['dgrad'] if sh == 1 else []):
        outs = {}
        for pp in ('0', '2'):
            os.environ['IFCBK_CONV_PP3'] = pp
            kn = kname(d, {'fwd': _lib.OP_CONV_FWD, 'affine': _lib.OP_CONV_FWD_AFFINE, 'dgrad': _lib.OP_CONV_DGRAD}[mode])
            if mode == 'fwd':
                y = torch.full((N, P, Q, K), float('nan'), device='cuda', dtype=torch.bfloat16)
                mb = ctx.lib.ifcbk_conv2d_fwd_mblocks(C.byref(d))
                part = torch.full((mb, 2, K), float('nan'), device='cuda')
                run = lambda y=y, part=part: ctx.call('ifcbk_conv2d_fwd', C.byref(d), _lib.ptr(x), _lib.ptr(w), _lib.ptr(y), _lib.ptr(part), st)
                run(); torch.cuda.synchronize()
                outs[pp] = (kn, run, y, part[:, 0].double().sum(0))
            elif mode == 'affine':
                y = torch.full((N, P, Q, K), float('nan'), device='cuda', dtype=torch.bfloat16)
                run = lambda y=y: ctx.call('ifcbk_conv2d_fwd_affine', C.byref(d), _lib.ptr(x), _lib.ptr(w), _lib.ptr(y), _lib.ptr(scale), _lib.ptr(shift), None, 0, 1, st)
                run(); torch.cuda.synchronize()
                outs[pp] = (kn, run, y, None)
            else:
                dx = torch.full((N, H, W, Cc), float('nan'), device='cuda', dtype=torch.bfloat16)
                run = lambda dx=dx: ctx.call('ifcbk_conv2d_dgrad', C.byref(d), _lib.ptr(dy), _lib.ptr(wT), _lib.ptr(dx), 0, st)
                run(); torch.cuda.synchronize()
                outs[pp] = (kn, run, dx, None)
        ms = {'0': [], '2': []}
        for r in range(reps):
            for pp in ('0', '2'):
                os.environ['IFCBK_CONV_PP3'] = pp
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    outs[pp][1]()
                e1.record(); torch.cuda.synchronize()
                ms[pp].append(e0.elapsed_time(e1) / 3)
        m0, m1 = min(ms['0']), min(ms['2'])
        cross = rel(outs['2'][2].float(), outs['0'][2].float())
        es = rel(outs['2'][3], outs['0'][3]) if outs['0'][3] is not None else 0.0
        print('%-9s N=%-4d %-6s old %-36s %7.3f ms %5.0f TF | %-34s %7.3f ms %5.0f TF x%.2f | vs old %.1e stat %.1e'
              % (name, N, mode, outs['0'][0], m0, flops / m0 / 1e9, outs['2'][0], m1, flops / m1 / 1e9, m0 / m1, cross, es), flush=True)
