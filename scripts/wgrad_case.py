"""dev tool: one weight-gradient case through the C-ABI against torch (fp32 and bf16), per-tap error
usage: python scripts/wgrad_case.py N Cw H W K R S stride pad [dtype]"""
import ctypes as C
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, '.')
from ifcb_classifier_amd import _lib

N, Cw, H, W, K, R, S, st, pad = [int(v) for v in sys.argv[1:10]]
dt = sys.argv[10] if len(sys.argv) > 10 else 'fp32'
td = torch.float32 if dt == 'fp32' else torch.bfloat16
Cp = (Cw + 7) // 8 * 8
ctx = _lib.Context(0)
g = torch.Generator().manual_seed(0)
x = torch.randn(N, Cw, H, W, generator=g).to(td).float()
P, Q = (H + 2 * pad - R) // st + 1, (W + 2 * pad - S) // st + 1
dy = torch.randn(N, K, P, Q, generator=g).to(td).float()
w = (torch.randn(K, Cw, R, S, generator=g) / (Cw * R * S) ** 0.5).to(td).float().requires_grad_(True)
xr = x.clone().requires_grad_(True)
yref = F.conv2d(xr, w, None, st, pad)
yref.backward(dy)
d = _lib.ConvDesc(N, H, W, Cp, Cp, K, R, S, st, st, pad, pad, P, Q, K, Cw, _lib.F32 if dt == 'fp32' else _lib.BF16)
ctx.reserve(max(1 << 20, ctx.lib.ifcbk_conv2d_wgrad_workspace(C.byref(d))))
xd = torch.zeros(N, H, W, Cp, dtype=td)
xd[..., :Cw] = x.permute(0, 2, 3, 1).to(td)
xd = xd.cuda()
dyd = dy.permute(0, 2, 3, 1).contiguous().to(td).cuda()
dw = torch.full((K, R, S, Cw), float('nan'), device='cuda')
ctx.call('ifcbk_conv2d_wgrad', C.byref(d), _lib.ptr(xd), _lib.ptr(dyd), _lib.ptr(dw), 0, _lib.cur_stream())
torch.cuda.synchronize()
got = dw.cpu().permute(0, 3, 1, 2)
ref = w.grad
print('rel', float((got - ref).norm() / ref.norm()))
for r in range(R):
    print(['%.1e' % float((got[:, :, r, s] - ref[:, :, r, s]).norm() / ref[:, :, r, s].norm()) for s in range(S)])
op = _lib.Op()
op.kind = _lib.OP_CONV_WGRAD
op.u.conv = d
buf = C.create_string_buffer(256)
ctx.lib.ifcbk_op_kernel(C.byref(op), buf, 256)
print('kernel', buf.value.decode())

if Cw == Cp:
    wm = w.detach().permute(0, 2, 3, 1).contiguous().cuda()
    wk = torch.empty(K, R, S, Cp, dtype=td, device='cuda')
    wT = torch.empty(Cp, R, S, K, dtype=td, device='cuda')
    ctx.call('ifcbk_weight_pack', C.byref(d), _lib.ptr(wm), _lib.ptr(wk), _lib.ptr(wT), _lib.cur_stream())
    y = torch.full((N, P, Q, K), float('nan'), dtype=td, device='cuda')
    ctx.call('ifcbk_conv2d_fwd', C.byref(d), _lib.ptr(xd), _lib.ptr(wk), _lib.ptr(y), None, _lib.cur_stream())
    dx = torch.full((N, H, W, Cp), float('nan'), dtype=td, device='cuda')
    ctx.call('ifcbk_conv2d_dgrad', C.byref(d), _lib.ptr(dyd), _lib.ptr(wT), _lib.ptr(dx), 0, _lib.cur_stream())
    torch.cuda.synchronize()
    yh = y.float().cpu().permute(0, 3, 1, 2)
    dxh = dx.float().cpu().permute(0, 3, 1, 2)
    print('fwd rel', float((yh - yref.detach()).norm() / yref.detach().norm()), 'dgrad rel', float((dxh - xr.grad).norm() / xr.grad.norm()))
    e = (dxh - xr.grad).abs()
    print('dgrad worst', float(e.max()), 'at', [int(v) for v in torch.nonzero(e == e.max())[0]], 'count >1e-3', int((e > 1e-3).sum()))
    op.kind = _lib.OP_CONV_DGRAD
    ctx.lib.ifcbk_op_kernel(C.byref(op), buf, 256)
    print('dgrad kernel', buf.value.decode())
    eg = (got - ref.detach()).abs()
    print('wgrad worst', float(eg.max()), 'count >1e-3', int((eg > 1e-3 * float(ref.abs().max())).sum()), 'of', eg.numel())
