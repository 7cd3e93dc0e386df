"""dev tool: the u8-plane stem path (roi_preprocess -> u8 plane, ifcbk_stem_u8_fwd / _wgrad) against the dense path
(roi_preprocess -> [N,S,S,8] tensor, ifcbk_conv2d_fwd / _wgrad) at the benchmark's shape: per-kernel times, interleaved.
    python scripts/stem_u8_check.py [N=256] [reps=7]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ifcb_classifier_amd import _lib
from ifcb_classifier_amd._lib import ConvDesc, RoiDesc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
S, K = 299, 32
P = Q = (S - 3) // 2 + 1
ctx = _lib.Context(0)
st = _lib.cur_stream()
rng = np.random.default_rng(1)
hs = rng.integers(32, 300, N).astype(np.int32)
ws = rng.integers(32, 300, N).astype(np.int32)
sizes = hs.astype(np.int64) * ws
offs = np.zeros(N, np.int64)
offs[1:] = np.cumsum(sizes)[:-1]
pix = torch.from_numpy(rng.integers(0, 256, int(sizes.sum()), dtype=np.uint8)).cuda()
offs_d, hs_d, ws_d = torch.from_numpy(offs).cuda(), torch.from_numpy(hs).cuda(), torch.from_numpy(ws).cuda()
rd = RoiDesc()
rd.n_img, rd.S, rd.in_channels, rd.out_channels, rd.flip_bits_valid, rd.dtype = N, S, 1, 8, 0, _lib.BF16
for k in range(3):
    rd.mean[k], rd.std[k], rd.tin_scale[k], rd.tin_shift[k] = 0.0, 1.0, 1.0, 0.0
d = ConvDesc(N, S, S, 8, 8, K, 3, 3, 2, 2, 0, 0, P, Q, K, 3, _lib.BF16)
need = max(ctx.lib.ifcbk_roi_preprocess_workspace(C.byref(rd), int(hs.max()), int(ws.max())), ctx.lib.ifcbk_conv2d_wgrad_workspace(C.byref(d)),
           ctx.lib.ifcbk_stem_u8_wgrad_workspace(C.byref(d)))
ctx.reserve(need)
x8 = torch.zeros(N, S, S, 8, device='cuda', dtype=torch.bfloat16)
g8 = torch.zeros(N, S, S, device='cuda', dtype=torch.uint8)
wm = torch.randn(K, 3, 3, 3, device='cuda') * 0.2
wsh = torch.zeros(K, 3, 3, 8, device='cuda', dtype=torch.bfloat16)
wsh[..., :3] = wm.bfloat16()
ab = torch.tensor([1 / 255.0] * 3 + [0.0] * 3, device='cuda')
y = torch.zeros(N, P, Q, K, device='cuda', dtype=torch.bfloat16)
dy = torch.randn(N, P, Q, K, device='cuda').bfloat16()
mb = ctx.lib.ifcbk_conv2d_fwd_mblocks(C.byref(d))
part = torch.zeros(max(mb, ctx.lib.ifcbk_stem_u8_rows(C.byref(d))) * 2 * K, device='cuda')
dw = torch.zeros(K, 3, 3, 3, device='cuda')
scale, shift = torch.ones(K, device='cuda'), torch.zeros(K, device='cuda')

runs = {
    'roi -> [N,S,S,8] bf16': lambda: ctx.call('ifcbk_roi_preprocess', C.byref(rd), _lib.ptr(pix), _lib.ptr(offs_d), _lib.ptr(hs_d), _lib.ptr(ws_d),
                                              None, int(hs.max()), int(ws.max()), _lib.ptr(x8), None, st),
    'roi -> u8 plane': lambda: ctx.call('ifcbk_roi_preprocess', C.byref(rd), _lib.ptr(pix), _lib.ptr(offs_d), _lib.ptr(hs_d), _lib.ptr(ws_d),
                                        None, int(hs.max()), int(ws.max()), None, _lib.ptr(g8), st),
    'conv2d_fwd (dense)': lambda: ctx.call('ifcbk_conv2d_fwd', C.byref(d), _lib.ptr(x8), _lib.ptr(wsh), _lib.ptr(y), _lib.ptr(part), st),
    'stem_u8_fwd': lambda: ctx.call('ifcbk_stem_u8_fwd', C.byref(d), _lib.ptr(g8), _lib.ptr(wm), _lib.ptr(ab), _lib.ptr(y), _lib.ptr(part), None, None, 0, st),
    'stem_u8_fwd (eval affine)': lambda: ctx.call('ifcbk_stem_u8_fwd', C.byref(d), _lib.ptr(g8), _lib.ptr(wm), _lib.ptr(ab), _lib.ptr(y), None,
                                                  _lib.ptr(scale), _lib.ptr(shift), 1, st),
    'conv2d_wgrad (dense)': lambda: ctx.call('ifcbk_conv2d_wgrad', C.byref(d), _lib.ptr(x8), _lib.ptr(dy), _lib.ptr(dw), 0, st),
    'stem_u8_wgrad': lambda: ctx.call('ifcbk_stem_u8_wgrad', C.byref(d), _lib.ptr(g8), _lib.ptr(dy), _lib.ptr(ab), _lib.ptr(dw), 0, st),
}
ms = {k: [] for k in runs}
for r in range(reps + 1):
    for k, f in runs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            f()
        e1.record()
        torch.cuda.synchronize()
        if r:
            ms[k].append(e0.elapsed_time(e1) / 3)
out_b = N * P * Q * K * 2
for k, v in ms.items():
    print('%-28s %7.1f us   (%.2f TB/s against the %d MB tensor)' % (k, 1e3 * min(v), out_b / min(v) / 1e9, out_b >> 20))
