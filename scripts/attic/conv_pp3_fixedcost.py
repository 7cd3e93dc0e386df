"""dev tool: per-K-tile time and fixed per-tile cost of conv_pp2<3,8,4,0> and conv_pp3 on the SAME 256 x 192 tile: a plain 1x1 GEMM
with exactly one tile per CU (64 images of 16x16 pixels x 4 column tiles of 192 channels), reduction lengths nk = 8 .. 64 K-tiles; and with TWO / FOUR tiles
per CU (128 / 256 / 512 images), where conv_pp3 walks tiles inside a block and conv_pp2 starts a block per tile.
    python scripts/conv_pp3_fixedcost.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ifcb_classifier_amd import _lib
from ifcb_classifier_amd._lib import ConvDesc

KO = 768          # four 192-channel column tiles: the pixel operand is re-read from L2, the GEMM is MFMA-bound
ctx = _lib.Context(0)
ctx.reserve(1 << 28)
st = _lib.cur_stream()


def kname(d):
    op = _lib.Op()
    op.kind = _lib.OP_CONV_FWD
    op.u.conv = d
    buf = C.create_string_buffer(96)
    ctx.lib.ifcbk_op_kernel(C.byref(op), buf, 96)
    return buf.value.decode()


for N in (64, 128, 256, 512):
    for Cc in (512, 1024, 2048):
        d = ConvDesc(N, 16, 16, Cc, Cc, KO, 1, 1, 1, 1, 0, 0, 16, 16, KO, Cc, 0)
        x = torch.randn(N, 16, 16, Cc, device='cuda').bfloat16()
        w = (torch.randn(KO, 1, 1, Cc, device='cuda') / Cc ** 0.5).bfloat16()
        y = torch.empty(N, 16, 16, KO, device='cuda', dtype=torch.bfloat16)
        res = {}
        for tag, env in (('pp2', dict(IFCBK_CONV_PP3='0', IFCBK_CONV_BIG='2', IFCBK_CONV_BIG_MT='8', IFCBK_CONV_BIG_TN='3')),
                         ('pp3', dict(IFCBK_CONV_PP3='2', IFCBK_CONV_BIG='0'))):
            for k in ('IFCBK_CONV_PP3', 'IFCBK_CONV_BIG', 'IFCBK_CONV_BIG_MT', 'IFCBK_CONV_BIG_TN'):
                os.environ.pop(k, None)
            os.environ.update(env)
            kn = kname(d)
            mb = ctx.lib.ifcbk_conv2d_fwd_mblocks(C.byref(d))
            part = torch.empty(mb, 2, KO, device='cuda')
            run = lambda: ctx.call('ifcbk_conv2d_fwd', C.byref(d), _lib.ptr(x), _lib.ptr(w), _lib.ptr(y), _lib.ptr(part), st)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            best = 1e9
            for r in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    run()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 5)
            res[tag] = (kn, best)
        print('N=%-5d tiles/CU=%d nk=%-3d  %s %.1f us   %s %.1f us' % (N, N // 64, Cc // 64, res['pp2'][0], 1e3 * res['pp2'][1], res['pp3'][0], 1e3 * res['pp3'][1]), flush=True)
