"""Eval-mode conv + folded BatchNorm affine + ReLU + 3x3 / stride-2 max pool in one kernel (ifcbk_conv2d_fwd_affine_maxpool, the
row-streaming kernel's pooled epilogue) against the two calls it replaces -- ifcbk_conv2d_fwd_affine then ifcbk_maxpool_fwd -- bit for
bit, and against torch ([TV] Inception3: Conv2d_2b_3x3 -> maxpool1 in eval mode; reference call site neuston_models.py:94-103)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('N,H,W,pad,relu', [(2, 147, 147, 1, 1), (3, 40, 37, 1, 1), (1, 35, 150, 0, 1), (5, 18, 21, 1, 0), (2, 3, 3, 1, 1),
                                            (1, 160, 160, 1, 1)])
def test_conv_affine_maxpool_equals_the_two_kernels_it_replaces(N, H, W, pad, relu):
    from ifcb_classifier_amd import _lib
    from ifcb_classifier_amd._lib import ConvDesc, PoolDesc
    ctx = _lib.Context(0)
    st = _lib.cur_stream()
    Cc, K = 32, 64
    P, Q = H + 2 * pad - 2, W + 2 * pad - 2
    if P < 3 or Q < 3:
        pytest.skip('no pooled output')
    Pp, Qp = (P - 3) // 2 + 1, (Q - 3) // 2 + 1
    LDX, LDP = Cc + 8, K + 16
    d = ConvDesc(N, H, W, Cc, LDX, K, 3, 3, 1, 1, pad, pad, P, Q, K, Cc, _lib.BF16)
    assert ctx.lib.ifcbk_conv2d_fwd_affine_maxpool_ok(C.byref(d)) == 1
    g = torch.Generator(device='cuda').manual_seed(5)
    xb = torch.randn(N, H, W, LDX, device='cuda', generator=g).bfloat16()
    x = xb[..., 4:4 + Cc]
    w = (torch.randn(K, 3, 3, Cc, device='cuda', generator=g) * 0.08).bfloat16()
    scale = torch.randn(K, device='cuda', generator=g) * 0.7            # (negative scales too: the affine is not monotone)
    shift = torch.randn(K, device='cuda', generator=g) * 0.3
    # the two kernels
    act = torch.empty(N, P, Q, K, device='cuda', dtype=torch.bfloat16)
    ctx.call('ifcbk_conv2d_fwd_affine', C.byref(d), _lib.ptr(x), _lib.ptr(w), _lib.ptr(act), _lib.ptr(scale), _lib.ptr(shift), None, 0,
             relu, st)
    want = torch.full((N, Pp, Qp, LDP), float('nan'), device='cuda', dtype=torch.bfloat16)
    pd = PoolDesc(N, P, Q, K, K, 3, 3, 2, 2, 0, 0, Pp, Qp, LDP, _lib.BF16)
    ctx.call('ifcbk_maxpool_fwd', C.byref(pd), _lib.ptr(act), _lib.ptr(want), None, st)
    # the one kernel
    got = torch.full((N, Pp, Qp, LDP), float('nan'), device='cuda', dtype=torch.bfloat16)
    ctx.call('ifcbk_conv2d_fwd_affine_maxpool', C.byref(d), _lib.ptr(x), _lib.ptr(w), _lib.ptr(got), LDP, _lib.ptr(scale), _lib.ptr(shift),
             relu, st)
    torch.cuda.synchronize()
    assert torch.isnan(got[..., K:].float()).all()
    assert torch.equal(got[..., :K], want[..., :K])
    # and torch
    y = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), None, 1, pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if relu:
        y = y.relu()
    ref = F.max_pool2d(y, 3, 2).permute(0, 2, 3, 1)
    assert ((got[..., :K].float() - ref).abs() <= 2.0 ** -7 * ref.abs() + 2e-2).all()


def test_conv_affine_maxpool_refuses_other_layers():
    from ifcb_classifier_amd import _lib
    from ifcb_classifier_amd._lib import ConvDesc
    ctx = _lib.Context(0)
    for d in (ConvDesc(2, 40, 40, 32, 32, 32, 3, 3, 1, 1, 1, 1, 40, 40, 32, 32, _lib.BF16),         # 32 -> 32
              ConvDesc(2, 40, 40, 32, 32, 64, 3, 3, 1, 1, 1, 1, 40, 40, 64, 32, _lib.F32),          # fp32 storage
              ConvDesc(2, 40, 200, 32, 32, 64, 3, 3, 1, 1, 1, 1, 40, 200, 64, 32, _lib.BF16),       # rows wider than 160
              ConvDesc(2, 40, 40, 80, 80, 192, 3, 3, 1, 1, 0, 0, 38, 38, 192, 80, _lib.BF16)):      # Conv2d_4a's shape
        assert ctx.lib.ifcbk_conv2d_fwd_affine_maxpool_ok(C.byref(d)) == 0
        t = torch.zeros(64, device='cuda')
        rc = ctx.lib.ifcbk_conv2d_fwd_affine_maxpool(ctx.h, C.byref(d), _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), 64, _lib.ptr(t), _lib.ptr(t), 1,
                                                     _lib.cur_stream())
        assert rc == -4


def test_eval_forward_with_and_without_the_fused_pool_is_bit_identical(monkeypatch):
    from ifcb_classifier_amd import _lib, graph
    from ifcb_classifier_amd.engine import Engine
    B = 3
    x = torch.rand(B, 3, 299, 299, device='cuda')
    out = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('IFCBK_FUSE_POOL_EVAL', mode)
        e = Engine(graph.build('inception_v3', 6), 0, max_batch=B)
        e.init_weights(seed=4)
        e.load_input_nchw(x)
        pl = e.forward_eval(B)
        torch.cuda.synchronize()
        kinds = [pl.fwd_eval.arr[k].kind for k in range(pl.fwd_eval.n)]
        assert (_lib.OP_CONV_FWD_AFFINE_MAXPOOL in kinds) == (mode == '1')
        out[mode] = [h for h in e.heads if not h.aux][0].logits[:B].clone()
        del e
    assert torch.equal(out['1'], out['0'])
