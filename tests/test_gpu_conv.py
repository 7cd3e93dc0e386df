"""HIP implicit-GEMM conv fwd / dgrad / wgrad through the C-ABI vs torch-CPU F.conv2d (the reference's
arithmetic: aten::conv2d reached from neuston_models.py:66-68) on bf16-representable inputs."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def nhwc(x, cpad=None):
    """fp32 NCHW cpu -> bf16 NHWC cuda (channel padded)"""
    n, c, h, w = x.shape
    cp = cpad or c
    out = torch.zeros(n, h, w, cp, dtype=torch.bfloat16)
    out[..., :c] = x.permute(0, 2, 3, 1).to(torch.bfloat16)
    return out.cuda()


def from_nhwc(y, c=None):
    y = y.float().cpu()
    if c is not None:
        y = y[..., :c]
    return y.permute(0, 3, 1, 2).contiguous()


CASES = [
    # N, C, H, W, K, R, S, sh, sw, ph, pw
    (2, 32, 9, 9, 32, 3, 3, 1, 1, 0, 0),
    (2, 16, 8, 8, 48, 1, 1, 1, 1, 0, 0),
    (3, 64, 12, 12, 96, 3, 3, 1, 1, 1, 1),
    (2, 48, 10, 10, 64, 5, 5, 1, 1, 2, 2),
    (2, 128, 9, 9, 128, 1, 7, 1, 1, 0, 3),
    (2, 128, 9, 9, 192, 7, 1, 1, 1, 3, 0),
    (2, 96, 11, 11, 96, 3, 3, 2, 2, 0, 0),
    (1, 288, 9, 9, 384, 3, 3, 2, 2, 0, 0),
    (2, 80, 15, 15, 192, 3, 3, 1, 1, 0, 0),
    (2, 8, 31, 31, 32, 3, 3, 2, 2, 0, 0),          # stem (C padded 3->8)
    (1, 768, 5, 5, 160, 1, 1, 1, 1, 0, 0),
    (2, 448, 8, 8, 384, 3, 3, 1, 1, 1, 1),
    (1, 64, 16, 16, 64, 7, 7, 2, 2, 3, 3),         # resnet stem-like
    (1, 64, 8, 8, 128, 1, 1, 2, 2, 0, 0),          # resnet downsample
    (1, 128, 5, 5, 768, 5, 5, 1, 1, 0, 0),         # aux conv1
    (5, 2048, 8, 8, 320, 1, 1, 1, 1, 0, 0),
    (2, 64, 10, 10, 64, 3, 3, 2, 2, 1, 1),         # resnet 3x3 stride 2 pad 1 (even input): parity-class dgrad
    (2, 32, 13, 11, 48, 3, 3, 2, 2, 1, 1),         # odd, non-square input
    (1, 16, 8, 8, 16, 2, 2, 2, 2, 0, 0),           # 2x2 stride 2: one tap per class
    (1, 24, 9, 9, 40, 5, 5, 2, 2, 2, 2),           # 5x5 stride 2: 9 / 6 / 6 / 4 taps
    (2, 32, 21, 19, 64, 3, 3, 1, 1, 1, 1),         # row-streaming stem kernel: 32 -> 64, pad 1 (dgrad 64 -> 32)
    (1, 32, 20, 149, 32, 3, 3, 1, 1, 0, 0),        # ... full 149-wide rows (10 pixel tiles), two row segments
    (3, 32, 35, 18, 32, 3, 3, 1, 1, 0, 0),         # ... three row segments per image
    (2, 32, 37, 147, 64, 3, 3, 1, 1, 1, 1),        # row-streaming stem weight gradient: full-width rows, pad 1, two K halves, 3 strips
    (2, 64, 19, 23, 32, 3, 3, 1, 1, 1, 1),         # row-streaming kernel, 64 -> 32 forward with statistics (its input gradient: 32 -> 64 raw)
    (2, 80, 21, 73, 192, 3, 3, 1, 1, 0, 0),        # ... its 80-channel form (Conv2d_4a): six K blocks, 104-pixel row images
]


def _desc(lib, N, Cc, H, W, K, R, S, sh, sw, ph, pw, ldx=None, ldy=None, Cw=None):
    from ifcb_classifier_amd._lib import ConvDesc
    P = (H + 2 * ph - R) // sh + 1
    Q = (W + 2 * pw - S) // sw + 1
    return ConvDesc(N, H, W, Cc, ldx or Cc, K, R, S, sh, sw, ph, pw, P, Q, ldy or K, Cw or Cc, 0)


@pytest.mark.parametrize('case', CASES)
def test_conv_fwd_dgrad_wgrad(ctx, case):
    from ifcb_classifier_amd import _lib
    N, Cc, H, W, K, R, S, sh, sw, ph, pw = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = _bf(torch.randn(N, Cc, H, W, generator=g))
    w = _bf(torch.randn(K, Cc, R, S, generator=g) * (1.0 / (Cc * R * S) ** 0.5))
    d = _desc(_lib, *case)
    P, Q = d.P, d.Q
    dy = _bf(torch.randn(N, K, P, Q, generator=g))
    # oracle (torch CPU fp32)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, (sh, sw), (ph, pw))
    yr.backward(dy)
    st = _lib.cur_stream()
    # ---- weight pack from the fp32 master (KRSC)
    wm = w.permute(0, 2, 3, 1).contiguous().cuda()
    wk = torch.empty(K, R, S, Cc, dtype=torch.bfloat16, device='cuda')
    wT = torch.empty(Cc, R, S, K, dtype=torch.bfloat16, device='cuda')
    ctx.call('ifcbk_weight_pack', C.byref(d), _lib.ptr(wm), _lib.ptr(wk), _lib.ptr(wT), st)
    assert torch.equal(wk.float().cpu(), w.permute(0, 2, 3, 1))
    # ---- forward (+ BN partial sums)
    xd = nhwc(x)
    y = torch.full((N, P, Q, K), float('nan'), dtype=torch.bfloat16, device='cuda')
    mb = ctx.lib.ifcbk_conv2d_fwd_mblocks(C.byref(d))
    part = torch.full((mb, 2, K), float('nan'), dtype=torch.float32, device='cuda')
    ctx.call('ifcbk_conv2d_fwd', C.byref(d), _lib.ptr(xd), _lib.ptr(wk), _lib.ptr(y), _lib.ptr(part), st)
    torch.cuda.synchronize()
    yh = from_nhwc(y)
    ref = yr.detach()
    tol = 1e-2 * ref.abs().max().item() + 1e-6       # bf16 output rounding (2^-8 rel) dominates
    assert (yh - _bf(ref)).abs().max().item() <= tol * 0.5, (yh - _bf(ref)).abs().max().item()
    # statistics are those of the ROUNDED outputs, exactly
    s1 = part[:, 0].sum(0).cpu()
    s2 = part[:, 1].sum(0).cpu()
    assert torch.allclose(s1, yh.sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
    assert torch.allclose(s2, (yh * yh).sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
    # ---- dgrad
    dyd = nhwc(dy)
    dx = torch.full((N, H, W, Cc), float('nan'), dtype=torch.bfloat16, device='cuda')
    ctx.call('ifcbk_conv2d_dgrad', C.byref(d), _lib.ptr(dyd), _lib.ptr(wT), _lib.ptr(dx), 0, st)
    torch.cuda.synchronize()
    dxh = from_nhwc(dx)
    rdx = xr.grad
    assert (dxh - _bf(rdx)).abs().max().item() <= 0.5e-2 * rdx.abs().max().item() + 1e-6
    # accumulate: second call doubles it
    ctx.call('ifcbk_conv2d_dgrad', C.byref(d), _lib.ptr(dyd), _lib.ptr(wT), _lib.ptr(dx), 1, st)
    torch.cuda.synchronize()
    assert (from_nhwc(dx) - 2 * _bf(rdx)).abs().max().item() <= 2e-2 * rdx.abs().max().item() + 1e-6
    # ---- wgrad (fp32 KRSC)
    dw = torch.full((K, R, S, Cc), float('nan'), dtype=torch.float32, device='cuda')
    ctx.reserve(ctx.lib.ifcbk_conv2d_wgrad_workspace(C.byref(d)))
    ctx.call('ifcbk_conv2d_wgrad', C.byref(d), _lib.ptr(xd), _lib.ptr(dyd), _lib.ptr(dw), 0, st)
    torch.cuda.synchronize()
    rdw = wr.grad.permute(0, 2, 3, 1)
    err = (dw.cpu() - rdw).abs().max().item()
    assert err <= 2e-5 * max(1.0, rdw.abs().max().item()) * (N * P * Q) ** 0.5, err


@pytest.mark.parametrize('case', [(2, 32, 20, 149, 32, 3, 3, 1, 1, 0, 0), (2, 32, 21, 19, 64, 3, 3, 1, 1, 1, 1),
                                  (2, 64, 19, 23, 32, 3, 3, 1, 1, 1, 1)])
@pytest.mark.parametrize('relu', [0, 1])
def test_row_streaming_kernel_affine_epilogue(ctx, case, relu):
    """conv_rows3x3<.., .., 2> (the eval epilogue of the row-streaming kernel, every channel pair it serves): y = act(conv * scale +
    shift) must be the raw forward's ROUNDED output put through the same affine -- same kernel body, only the epilogue differs"""
    from ifcb_classifier_amd import _lib
    N, Cc, H, W, K, R, S, sh, sw, ph, pw = case
    g = torch.Generator().manual_seed(7 + K + relu)
    x = nhwc(_bf(torch.randn(N, Cc, H, W, generator=g)))
    wk = (torch.randn(K, R, S, Cc, generator=g) * (1.0 / (Cc * R * S) ** 0.5)).to(torch.bfloat16).cuda()
    scale = (torch.rand(K, generator=g) + 0.5).cuda()
    shift = torch.randn(K, generator=g).cuda()
    d = _desc(_lib, *case)
    st = _lib.cur_stream()
    raw = torch.full((N, d.P, d.Q, K), float('nan'), dtype=torch.bfloat16, device='cuda')
    mb = ctx.lib.ifcbk_conv2d_fwd_mblocks(C.byref(d))
    part = torch.zeros(mb, 2, K, dtype=torch.float32, device='cuda')
    ctx.call('ifcbk_conv2d_fwd', C.byref(d), _lib.ptr(x), _lib.ptr(wk), _lib.ptr(raw), _lib.ptr(part), st)
    got = torch.full((N, d.P, d.Q, K), float('nan'), dtype=torch.bfloat16, device='cuda')
    ctx.call('ifcbk_conv2d_fwd_affine', C.byref(d), _lib.ptr(x), _lib.ptr(wk), _lib.ptr(got), _lib.ptr(scale), _lib.ptr(shift), None, 0, relu, st)
    torch.cuda.synchronize()
    # (the kernel's affine is one fused multiply-add: emulate the single rounding through double)
    want = (raw.double() * scale.double() + shift.double()).float()
    if relu:
        want = want.clamp_min(0)
    assert torch.equal(got, want.to(torch.bfloat16))


def test_conv_channel_slices(ctx):
    """output written into a channel slice of a wider concat buffer; input read from a slice."""
    from ifcb_classifier_amd import _lib
    N, Cc, H, W, K = 2, 32, 7, 7, 64
    g = torch.Generator().manual_seed(5)
    xfull = _bf(torch.randn(N, 96, H, W, generator=g))
    w = _bf(torch.randn(K, Cc, 1, 1, generator=g) * 0.2)
    d = _desc(_lib, N, Cc, H, W, K, 1, 1, 1, 1, 0, 0, ldx=96, ldy=160)
    xd = nhwc(xfull)
    wk = w.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()
    y = torch.zeros(N, H, W, 160, dtype=torch.bfloat16, device='cuda')
    st = _lib.cur_stream()
    ctx.call('ifcbk_conv2d_fwd', C.byref(d), C.c_void_p(xd.data_ptr() + 2 * 40), _lib.ptr(wk),
             C.c_void_p(y.data_ptr() + 2 * 64), None, st)
    torch.cuda.synchronize()
    ref = F.conv2d(xfull[:, 40:72], w)
    yh = from_nhwc(y)
    assert (yh[:, 64:128] - _bf(ref)).abs().max().item() <= 1e-2 * ref.abs().max().item()
    assert yh[:, :64].abs().max().item() == 0 and yh[:, 128:].abs().max().item() == 0


def test_conv_rejects_bad_desc(ctx):
    from ifcb_classifier_amd import _lib
    d = _desc(_lib, 1, 12, 8, 8, 16, 3, 3, 1, 1, 0, 0)      # C not a multiple of 8
    with pytest.raises(RuntimeError, match='multiples of 8'):
        ctx.call('ifcbk_conv2d_fwd', C.byref(d), None, None, None, None, None)


@pytest.mark.parametrize('case', CASES[:10] + CASES[12:14])
def test_conv_fp32_parity_mode(ctx, case):
    """fp32 storage / v_mfma_f32_16x16x4_f32 path vs torch-CPU fp32 conv: 1e-5 relative."""
    from ifcb_classifier_amd import _lib
    from ifcb_classifier_amd._lib import ConvDesc
    N, Cc, H, W, K, R, S, sh, sw, ph, pw = case
    g = torch.Generator().manual_seed(7 + hash(case) % 1000)
    x = torch.randn(N, Cc, H, W, generator=g)
    w = torch.randn(K, Cc, R, S, generator=g) * (1.0 / (Cc * R * S) ** 0.5)
    P = (H + 2 * ph - R) // sh + 1
    Q = (W + 2 * pw - S) // sw + 1
    d = ConvDesc(N, H, W, Cc, Cc, K, R, S, sh, sw, ph, pw, P, Q, K, Cc, 1)
    dy = torch.randn(N, K, P, Q, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, (sh, sw), (ph, pw))
    yr.backward(dy)
    st = _lib.cur_stream()
    wm = w.permute(0, 2, 3, 1).contiguous().cuda()
    wk = torch.empty(K, R, S, Cc, device='cuda')
    wT = torch.empty(Cc, R, S, K, device='cuda')
    ctx.call('ifcbk_weight_pack', C.byref(d), _lib.ptr(wm), _lib.ptr(wk), _lib.ptr(wT), st)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    dyd = dy.permute(0, 2, 3, 1).contiguous().cuda()
    y = torch.full((N, P, Q, K), float('nan'), device='cuda')
    mb = ctx.lib.ifcbk_conv2d_fwd_mblocks(C.byref(d))
    part = torch.zeros(mb, 2, K, device='cuda')
    ctx.call('ifcbk_conv2d_fwd', C.byref(d), _lib.ptr(xd), _lib.ptr(wk), _lib.ptr(y), _lib.ptr(part), st)
    dx = torch.full((N, H, W, Cc), float('nan'), device='cuda')
    ctx.call('ifcbk_conv2d_dgrad', C.byref(d), _lib.ptr(dyd), _lib.ptr(wT), _lib.ptr(dx), 0, st)
    dw = torch.full((K, R, S, Cc), float('nan'), device='cuda')
    ctx.reserve(ctx.lib.ifcbk_conv2d_wgrad_workspace(C.byref(d)))
    ctx.call('ifcbk_conv2d_wgrad', C.byref(d), _lib.ptr(xd), _lib.ptr(dyd), _lib.ptr(dw), 0, st)
    torch.cuda.synchronize()
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
    yh = y.cpu().permute(0, 3, 1, 2)
    assert rel(yh, yr.detach()) < 1e-5
    assert torch.allclose(part[:, 0].sum(0).cpu(), yh.sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
    assert rel(dx.cpu().permute(0, 3, 1, 2), xr.grad) < 1e-5
    assert rel(dw.cpu(), wr.grad.permute(0, 2, 3, 1)) < 1e-5


@pytest.mark.parametrize('case', [(2, 64, 9, 9, 96, 3, 3, 1, 1, 1, 1), (3, 192, 7, 7, 192, 1, 7, 1, 1, 0, 3),
                                  (2, 48, 10, 10, 64, 5, 5, 1, 1, 2, 2), (5, 768, 5, 5, 128, 1, 1, 1, 1, 0, 0)])
def test_dgrad_bnstat_equals_dgrad_then_bn_bwd(ctx, case):
    """ifcbk_conv2d_dgrad_bnstat + ifcbk_bn_bwd_partials == ifcbk_conv2d_dgrad + ifcbk_bn_bwd: same dx of the conv (bit for
    bit) and the same BatchNorm backward of the producing layer up to the summation order of the two channel sums."""
    from ifcb_classifier_amd import _lib
    N, Cc, H, W, K, R, S, sh, sw, ph, pw = case
    g = torch.Generator().manual_seed(hash(case) % 997)
    d = _desc(_lib, *case)
    P, Q = d.P, d.Q
    assert ctx.lib.ifcbk_conv2d_dgrad_bnstat_mblocks(C.byref(d)) > 0
    dy = _bf(torch.randn(N, P, Q, K, generator=g)).to(torch.bfloat16).cuda()
    wT = _bf(torch.randn(Cc, R, S, K, generator=g) * 0.1).to(torch.bfloat16).cuda()
    raw = _bf(torch.randn(N, H, W, Cc, generator=g) * 1.5).to(torch.bfloat16).cuda()       # the producer's raw conv output
    gamma = (torch.rand(Cc, generator=g) + 0.5).cuda()
    mean = (torch.randn(Cc, generator=g) * 0.2).cuda()
    invstd = (torch.rand(Cc, generator=g) + 0.5).cuda()
    scale = (gamma * invstd).contiguous()
    shift = (torch.randn(Cc, generator=g) * 0.3).cuda()
    st = _lib.cur_stream()
    M = N * H * W
    bd = _lib.BnDesc(M, Cc, Cc, Cc, 1, 0, 1e-3, 0.1)
    ctx.reserve(1 << 24)
    # reference: two separate steps
    dx = torch.empty(N, H, W, Cc, dtype=torch.bfloat16, device='cuda')
    ctx.call('ifcbk_conv2d_dgrad', C.byref(d), _lib.ptr(dy), _lib.ptr(wT), _lib.ptr(dx), 0, st)
    draw = torch.empty_like(dx)
    dg, db = torch.zeros(Cc).cuda(), torch.zeros(Cc).cuda()
    ctx.call('ifcbk_bn_bwd', C.byref(bd), _lib.ptr(raw), None, _lib.ptr(dx), Cc, _lib.ptr(gamma), _lib.ptr(mean),
             _lib.ptr(invstd), _lib.ptr(draw), Cc, None, 0, 0, _lib.ptr(dg), _lib.ptr(db), 0, _lib.ptr(scale), _lib.ptr(shift), st)
    # fused
    nrow = ctx.lib.ifcbk_conv2d_dgrad_bnstat_mblocks(C.byref(d))
    part = torch.zeros(nrow, 2, Cc, device='cuda')
    dx2 = torch.empty_like(dx)
    ctx.call('ifcbk_conv2d_dgrad_bnstat', C.byref(d), _lib.ptr(dy), _lib.ptr(wT), _lib.ptr(dx2), _lib.ptr(raw), Cc,
             _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(part), st)
    draw2 = torch.empty_like(dx)
    dg2, db2 = torch.zeros(Cc).cuda(), torch.zeros(Cc).cuda()
    ctx.call('ifcbk_bn_bwd_partials', C.byref(bd), _lib.ptr(raw), _lib.ptr(dx2), Cc, _lib.ptr(gamma), _lib.ptr(mean),
             _lib.ptr(invstd), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(part), nrow, _lib.ptr(draw2), Cc, _lib.ptr(dg2),
             _lib.ptr(db2), 0, st)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx2)
    for a, b in ((dg, dg2), (db, db2), (draw.float(), draw2.float())):
        assert (a - b).abs().max().item() <= 2e-3 * a.abs().max().item() + 1e-6, (a - b).abs().max().item()
