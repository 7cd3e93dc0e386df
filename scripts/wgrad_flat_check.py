"""dev tool: the flat-slot weight gradient (conv_wgrad_flat.hip) against what the dispatcher picks without it, on the inception_v3
shapes it serves at batch 256, in ONE process (IFCBK_WGRAD_FLAT is read per call): error against torch's fp32 GPU weight gradient of
the same bf16 operands, interleaved timing rounds (op = split-K kernel + reduce); then one Inception-A block's three layers as single
launches vs ONE grouped launch.      python scripts/wgrad_flat_check.py [reps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ifcb_classifier_amd import _lib
from ifcb_classifier_amd._lib import ConvDesc

LAYERS = {
    '5x_3x3a':  (256, 64, 35, 35, 96, 3, 3, 1, 1),
    '5x_3x3b':  (256, 96, 35, 35, 96, 3, 3, 1, 1),
    '5x_5x5':   (256, 48, 35, 35, 64, 5, 5, 2, 2),
    '4a_3x3':   (256, 80, 73, 73, 192, 3, 3, 0, 0),
}
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = _lib.Context(0)
ctx.reserve(1 << 30)
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


def timeit(fn, n=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


members = {}
for name, (N, Cc, H, W, K, R, S, ph, pw) in LAYERS.items():
    P, Q = H + 2 * ph - R + 1, W + 2 * pw - S + 1
    d = ConvDesc(N, H, W, Cc, Cc, K, R, S, 1, 1, ph, pw, P, Q, K, Cc, 0)
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(N, H, W, Cc, device='cuda', generator=g).bfloat16()
    dy = torch.randn(N, P, Q, K, device='cuda', generator=g).bfloat16()
    ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (K, Cc, R, S), dy.float().permute(0, 3, 1, 2), 1, (ph, pw)).permute(0, 2, 3, 1)
    flops = 2.0 * N * P * Q * K * R * S * Cc
    outs = {}
    for sw in ('0', '1'):
        os.environ['IFCBK_WGRAD_FLAT'] = sw
        ctx.reserve(ctx.lib.ifcbk_conv2d_wgrad_workspace(C.byref(d)))
        dw = torch.full((K, R, S, Cc), float('nan'), device='cuda')
        run = lambda dw=dw: ctx.call('ifcbk_conv2d_wgrad', C.byref(d), _lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), 0, st)
        run()
        torch.cuda.synchronize()
        outs[sw] = (kname(d), run, rel(dw, ref))
    ms = {'0': [], '1': []}
    for r in range(reps):
        for sw in ('0', '1'):
            os.environ['IFCBK_WGRAD_FLAT'] = sw
            ms[sw].append(timeit(outs[sw][1]))
    m0, m1 = min(ms['0']), min(ms['1'])
    print('%-9s old %-24s %7.3f ms %5.0f TF | new %-18s %7.3f ms %5.0f TF x%.2f | err old %.1e new %.1e'
          % (name, outs['0'][0], m0, flops / m0 / 1e9, outs['1'][0], m1, flops / m1 / 1e9, m0 / m1, outs['0'][2], outs['1'][2]), flush=True)
    members[name] = (d, x, dy, flops)

# one Inception-A block: its three multi-tap layers as single launches vs one grouped launch
os.environ['IFCBK_WGRAD_FLAT'] = '1'
grp = ['5x_5x5', '5x_3x3a', '5x_3x3b']
n = len(grp)
descs = (ConvDesc * n)(*[members[k][0] for k in grp])
xs, dys, dws = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
keep = []
for i, k in enumerate(grp):
    d, x, dy, _ = members[k]
    dw = torch.zeros(d.K, d.R, d.S, d.C, device='cuda')
    xs[i], dys[i], dws[i] = x.data_ptr(), dy.data_ptr(), dw.data_ptr()
    keep.append(dw)
need = ctx.lib.ifcbk_conv2d_wgrad_group_workspace(n, descs)
print('group workspace', need)
if need:
    ctx.reserve(max(need, 1 << 30))

    def singles():
        for i, k in enumerate(grp):
            d, x, dy, _ = members[k]
            ctx.call('ifcbk_conv2d_wgrad', C.byref(d), _lib.ptr(x), _lib.ptr(dy), _lib.ptr(keep[i]), 0, st)

    def grouped():
        ctx.call('ifcbk_conv2d_wgrad_group', n, descs, xs, dys, dws, 0, st)
    singles(); grouped(); torch.cuda.synchronize()
    a, b = [], []
    for r in range(reps):
        a.append(timeit(singles))
        b.append(timeit(grouped))
    fl = sum(members[k][3] for k in grp)
    print('Inception-A block: 3 single launches %.3f ms %.0f TF | one group %.3f ms %.0f TF x%.2f'
          % (min(a), fl / min(a) / 1e9, min(b), fl / min(b) / 1e9, min(a) / min(b)))
