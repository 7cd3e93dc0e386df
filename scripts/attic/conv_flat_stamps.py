"""dev tool: s_memtime stamps of conv_flat's phases (library built with -DIFCBK_EXPERIMENT_FLAT, IFCBK_FLAT_STAMPS=1)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ifcb_classifier_amd import _lib
from ifcb_classifier_amd._lib import ConvDesc
N, Cc, H, W, K, R, S, ph, pw = 256, 96, 35, 35, 96, 3, 3, 1, 1
ctx = _lib.Context(0); ctx.reserve(1 << 30); st = _lib.cur_stream()
os.environ['IFCBK_CONV_FLAT'] = '2'
d = ConvDesc(N, H, W, Cc, Cc, K, R, S, 1, 1, ph, pw, H, W, K, Cc, 0)
x = torch.randn(N, H, W, Cc, device='cuda').bfloat16(); w = (torch.randn(K, R, S, Cc, device='cuda') * 0.05).bfloat16()
y = torch.empty(N, H, W, K, device='cuda', dtype=torch.bfloat16)
part = torch.empty(ctx.lib.ifcbk_conv2d_fwd_mblocks(C.byref(d)), 2, K, device='cuda')
for i in range(3):
    if i == 2: os.environ['IFCBK_FLAT_STAMPS'] = '1'
    ctx.call('ifcbk_conv2d_fwd', C.byref(d), _lib.ptr(x), _lib.ptr(w), _lib.ptr(y), _lib.ptr(part), st)
    torch.cuda.synchronize()
