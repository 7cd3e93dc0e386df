"""dev tool: time representative inception_v3 conv layers (fwd/dgrad/wgrad) through the C-ABI."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ifcb_classifier_amd import _lib
from ifcb_classifier_amd._lib import ConvDesc

LAYERS = {
    # name: N, C, H, W, K, R, S, sh, sw, ph, pw
    '2a_3x3':   (256, 32, 149, 149, 32, 3, 3, 1, 1, 0, 0),
    '2b_3x3':   (256, 32, 147, 147, 64, 3, 3, 1, 1, 1, 1),
    '3b_1x1':   (256, 64, 73, 73, 80, 1, 1, 1, 1, 0, 0),
    '4a_3x3':   (256, 80, 73, 73, 192, 3, 3, 1, 1, 0, 0),
    '5b_5x5':   (256, 48, 35, 35, 64, 5, 5, 1, 1, 2, 2),
    '5c_3x3b':  (256, 96, 35, 35, 96, 3, 3, 1, 1, 1, 1),
    '5c_1x1':   (256, 256, 35, 35, 64, 1, 1, 1, 1, 0, 0),
    '6a_3x3s2': (256, 288, 35, 35, 384, 3, 3, 2, 2, 0, 0),
    '6c_1x7':   (256, 160, 17, 17, 160, 1, 7, 1, 1, 0, 3),
    '6e_7x1':   (256, 192, 17, 17, 192, 7, 1, 1, 1, 3, 0),
    '6b_1x1':   (256, 768, 17, 17, 192, 1, 1, 1, 1, 0, 0),
    '7b_3x3':   (256, 448, 8, 8, 384, 3, 3, 1, 1, 1, 1),
    '7c_1x1':   (256, 2048, 8, 8, 320, 1, 1, 1, 1, 0, 0),
}
which = sys.argv[1].split(',') if len(sys.argv) > 1 and sys.argv[1] != 'all' else list(LAYERS)
modes = sys.argv[2].split(',') if len(sys.argv) > 2 else ['fwd', 'dgrad', 'wgrad']
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
NOVR = int(os.environ.get('CONV_LAYERS_N', '0'))
ctx = _lib.Context(0)
ctx.reserve(1 << 30)
st = _lib.cur_stream()
tot = {m: [0.0, 0.0] for m in modes}
for name in which:
    N, Cc, H, W, K, R, S, sh, sw, ph, pw = LAYERS[name]
    if NOVR: N = NOVR
    P = (H + 2 * ph - R) // sh + 1; Q = (W + 2 * pw - S) // sw + 1
    d = ConvDesc(N, H, W, Cc, Cc, K, R, S, sh, sw, ph, pw, P, Q, K, Cc, 0)
    x = torch.randn(N, H, W, Cc, device='cuda').bfloat16()
    w = (torch.randn(K, R, S, Cc, device='cuda') * 0.05).bfloat16()
    wT = (torch.randn(Cc, R, S, K, device='cuda') * 0.05).bfloat16()
    y = torch.empty(N, P, Q, K, device='cuda', dtype=torch.bfloat16)
    dy = torch.randn(N, P, Q, K, device='cuda').bfloat16()
    dx = torch.empty_like(x)
    dw = torch.empty(K, R, S, Cc, device='cuda')
    part = torch.empty(ctx.lib.ifcbk_conv2d_fwd_mblocks(C.byref(d)), 2, K, device='cuda')
    flops = 2.0 * N * P * Q * K * R * S * Cc
    for mode in modes:
        def run():
            if mode == 'fwd':
                ctx.call('ifcbk_conv2d_fwd', C.byref(d), _lib.ptr(x), _lib.ptr(w), _lib.ptr(y), _lib.ptr(part), st)
            elif mode == 'dgrad':
                ctx.call('ifcbk_conv2d_dgrad', C.byref(d), _lib.ptr(dy), _lib.ptr(wT), _lib.ptr(dx), 0, st)
            else:
                ctx.call('ifcbk_conv2d_wgrad', C.byref(d), _lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), 0, st)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tot[mode][0] += ms; tot[mode][1] += flops
        print('%-10s %-6s %8.3f ms  %7.1f TF/s' % (name, mode, ms, flops / ms / 1e9), flush=True)
for m in modes:
    print('TOTAL %-6s %8.3f ms %7.1f TF/s' % (m, tot[m][0], tot[m][1] / tot[m][0] / 1e9))
