"""dev tool: s_memtime stamps of conv_rows3x3's per-row phases (a library whose conv_rows.hip was built with the stamp patch, IFCBK_LIB):
prologue, A = wait for the row DMA + barrier, B = MFMAs + C-row write, C = barrier, D = epilogue; wave 0 of every block, summed per block."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ifcb_classifier_amd import _lib
from ifcb_classifier_amd._lib import ConvDesc
for name, (N, Cc, H, W, K, pad) in {'2a fwd': (256, 32, 149, 149, 32, 0), '2b fwd': (256, 32, 147, 147, 64, 1)}.items():
    ctx = _lib.Context(0); ctx.reserve(1 << 28); st = _lib.cur_stream()
    P, Q = H + 2 * pad - 2, W + 2 * pad - 2
    d = ConvDesc(N, H, W, Cc, Cc, K, 3, 3, 1, 1, pad, pad, P, Q, K, Cc, 0)
    x = torch.randn(N, H, W, Cc, device='cuda').bfloat16(); w = (torch.randn(K, 3, 3, Cc, device='cuda') * 0.05).bfloat16()
    y = torch.empty(N, P, Q, K, device='cuda', dtype=torch.bfloat16)
    nb = ctx.lib.ifcbk_conv2d_fwd_mblocks(C.byref(d))
    part = torch.zeros(nb, 2, K, device='cuda')
    for i in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.call('ifcbk_conv2d_fwd', C.byref(d), _lib.ptr(x), _lib.ptr(w), _lib.ptr(y), _lib.ptr(part), st)
        e1.record(); torch.cuda.synchronize()
    v = part.view(nb, -1)[:, :7].double().cpu()
    rows = v[:, 6]
    full = rows == 16
    m = v[full].mean(0)
    print('%s: %.3f ms, %d blocks; per block (ticks of s_memtime): prologue %.0f, total %.0f; per row: A wait+barrier %.0f, B mfma %.0f, C barrier %.0f, D epilogue %.0f'
          % (name, e0.elapsed_time(e1), nb, m[0], m[5], m[1] / 16, m[2] / 16, m[3] / 16, m[4] / 16))
