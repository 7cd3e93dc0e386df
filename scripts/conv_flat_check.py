"""dev tool: the flat-image kernel (conv_flat.hip) against conv_igemm on the 35x35-stage layer shapes, in ONE process
(IFCBK_CONV_FLAT is read per launch): forward (+ BatchNorm partial sums), plain / accumulating input gradient and the input
gradient with fused BN-backward sums, against torch's fp32 GPU convolution of the same bf16 operands; then interleaved timing.
    python scripts/conv_flat_check.py [layers|all] [reps]          (CONV_LAYERS_N overrides the batch)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from ifcb_classifier_amd import _lib
from ifcb_classifier_amd._lib import ConvDesc

LAYERS = {
    # name: N, C, H, W, K, R, S, ph, pw
    '5b_5x5':   (256, 48, 35, 35, 64, 5, 5, 2, 2),
    '5c_3x3a':  (256, 64, 35, 35, 96, 3, 3, 1, 1),
    '5c_3x3b':  (256, 96, 35, 35, 96, 3, 3, 1, 1),
    # odd shapes: tails, unpadded / asymmetric padding, tiny maps
    't_3x3b':   (3, 96, 17, 13, 96, 3, 3, 1, 1),
    't_3x3p0':  (5, 64, 19, 23, 96, 3, 3, 0, 0),
    't_5x5':    (2, 48, 9, 31, 64, 5, 5, 2, 2),
    't_5x5p1':  (7, 48, 12, 12, 64, 5, 5, 1, 2),
    't_3x3p2':  (4, 96, 8, 8, 96, 3, 3, 2, 1),
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


bad = 0
tot = {}
for name in which:
    N, Cc, H, W, K, R, S, ph, pw = LAYERS[name]
    if NOVR and N == 256:
        N = NOVR
    P, Q = H + 2 * ph - R + 1, W + 2 * pw - S + 1
    # the tensors are channel slices of wider buffers (ldx / ldy > channels), as the Inception concatenations are
    LDX, LDY = Cc + 16, K + 8
    d = ConvDesc(N, H, W, Cc, LDX, K, R, S, 1, 1, ph, pw, P, Q, LDY, Cc, 0)
    g = torch.Generator(device='cuda').manual_seed(1)
    xb = torch.randn(N, H, W, LDX, device='cuda', generator=g).bfloat16()
    x = xb[..., 8:8 + Cc]
    w = (torch.randn(K, R, S, Cc, device='cuda', generator=g) * (1.0 / (R * S * Cc) ** 0.5)).bfloat16()
    wT = w.permute(3, 1, 2, 0).flip(1, 2).contiguous()
    dyb = torch.randn(N, P, Q, LDY, device='cuda', generator=g).bfloat16()
    dy = dyb[..., 8:8 + K]
    flops = 2.0 * N * P * Q * K * R * S * Cc
    xf, wf, dyf = x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), dy.float().permute(0, 3, 1, 2)
    ref_y = F.conv2d(xf, wf, None, 1, (ph, pw)).permute(0, 2, 3, 1)
    ref_dx = torch.nn.grad.conv2d_input((N, Cc, H, W), wf, dyf, 1, (ph, pw)).permute(0, 2, 3, 1)
    # the producing BatchNorm of the fused-sums variant: raw input, statistics
    raw = torch.randn(N, H, W, Cc, device='cuda', generator=g).bfloat16()
    mean, invstd = torch.randn(Cc, device='cuda', generator=g) * 0.1, torch.rand(Cc, device='cuda', generator=g) + 0.5
    scale, shift = torch.rand(Cc, device='cuda', generator=g) + 0.5, torch.randn(Cc, device='cuda', generator=g) * 0.3
    act = raw.float() * scale + shift > 0
    xhat = (raw.float() - mean) * invstd
    for mode in ('fwd', 'dgrad', 'dgrad_acc', 'dgrad_bs'):
        outs = {}
        for flat in ('0', '2'):
            os.environ['IFCBK_CONV_FLAT'] = flat
            kn = kname(d, {'fwd': _lib.OP_CONV_FWD, 'dgrad_bs': _lib.OP_CONV_DGRAD_BNSTAT}.get(mode, _lib.OP_CONV_DGRAD))
            if mode == 'fwd':
                yb = torch.full((N, P, Q, LDY), float('nan'), device='cuda', dtype=torch.bfloat16)
                y = yb[..., 8:8 + K]
                mb = ctx.lib.ifcbk_conv2d_fwd_mblocks(C.byref(d))
                part = torch.full((mb, 2, K), float('nan'), device='cuda')
                run = lambda y=y, part=part: ctx.call('ifcbk_conv2d_fwd', C.byref(d), _lib.ptr(x), _lib.ptr(w), _lib.ptr(y), _lib.ptr(part), st)
                run(); torch.cuda.synchronize()
                e = rel(y.float(), ref_y)
                es = max(rel(part[:, 0].double().sum(0), y.float().double().sum((0, 1, 2))),
                         rel(part[:, 1].double().sum(0), (y.float().double() ** 2).sum((0, 1, 2))))
                untouched = bool(torch.isnan(yb[..., :8].float()).all() and torch.isnan(yb[..., 8 + K:].float()).all())
                outs[flat] = (kn, run, e, es if untouched else 1.0, y)
            elif mode in ('dgrad', 'dgrad_acc'):
                acc = mode == 'dgrad_acc'
                base = torch.randn(N, H, W, LDX, device='cuda', generator=torch.Generator(device='cuda').manual_seed(7)).bfloat16()
                dxb = base.clone() if acc else torch.full((N, H, W, LDX), float('nan'), device='cuda', dtype=torch.bfloat16)
                dx = dxb[..., 8:8 + Cc]
                run = lambda dx=dx, acc=acc: ctx.call('ifcbk_conv2d_dgrad', C.byref(d), _lib.ptr(dy), _lib.ptr(wT), _lib.ptr(dx), int(acc), st)
                run(); torch.cuda.synchronize()
                want = ref_dx + base[..., 8:8 + Cc].float() if acc else ref_dx
                outs[flat] = (kn, run, rel(dx.float(), want), 0.0, dx)
            else:
                mb = ctx.lib.ifcbk_conv2d_dgrad_bnstat_mblocks(C.byref(d))
                dx = torch.full((N, H, W, Cc), float('nan'), device='cuda', dtype=torch.bfloat16)
                d2 = ConvDesc(N, H, W, Cc, Cc, K, R, S, 1, 1, ph, pw, P, Q, LDY, Cc, 0)
                part = torch.full((max(mb, 1), 2, Cc), float('nan'), device='cuda')
                run = lambda dx=dx, part=part, d2=d2: ctx.call('ifcbk_conv2d_dgrad_bnstat', C.byref(d2), _lib.ptr(dy), _lib.ptr(wT), _lib.ptr(dx), _lib.ptr(raw), Cc,
                                                               _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(part), st)
                if mb == 0:
                    outs[flat] = (kn + ' (no fused variant)', lambda: None, 0.0, 0.0, dx)
                    continue
                run(); torch.cuda.synchronize()
                dz = torch.where(act, dx.float(), torch.zeros((), device='cuda'))
                es = max(rel(part[:, 0].double().sum(0), dz.double().sum((0, 1, 2))),
                         rel(part[:, 1].double().sum(0), (dz.double() * xhat.double()).sum((0, 1, 2))))
                outs[flat] = (kn, run, rel(dx.float(), ref_dx), es, dx)
        ms = {'0': [], '2': []}
        for r in range(reps):
            for flat in ('0', '2'):
                os.environ['IFCBK_CONV_FLAT'] = flat
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    outs[flat][1]()
                e1.record(); torch.cuda.synchronize()
                ms[flat].append(e0.elapsed_time(e1) / 3)
        m0, m1 = min(ms['0']), min(ms['2'])
        ok = outs['2'][2] < 5e-3 and outs['2'][3] < 2e-4 and ('conv_flat' in outs['2'][0] or mode == 'dgrad_acc')
        print('%-8s %-9s old %-36s %7.3f ms %5.0f TF | new %-30s %7.3f ms %5.0f TF x%.2f | err old %.1e new %.1e stat %.1e / %.1e %s'
              % (name, mode, outs['0'][0], m0, flops / m0 / 1e9, outs['2'][0], m1, flops / m1 / 1e9, m0 / m1, outs['0'][2], outs['2'][2],
                 outs['0'][3], outs['2'][3], 'OK' if ok else 'BAD'), flush=True)
        bad += 0 if ok else 1
        if N >= 64:
            t = tot.setdefault(mode, [0.0, 0.0, 0.0])
            t[0] += m0; t[1] += m1; t[2] += flops
for m, t in tot.items():
    print('TOTAL %-9s old %.3f ms %.0f TF/s   new %.3f ms %.0f TF/s' % (m, t[0], t[2] / t[0] / 1e9, t[1], t[2] / t[1] / 1e9))
sys.exit(1 if bad else 0)
