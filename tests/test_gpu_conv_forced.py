"""The kernels that carry the benchmark -- the wide-tile ping-pong kernels (conv_pp2 / conv_big / conv_wgrad_pp) and the flat-image
kernel (conv_flat) -- FORCED onto shapes off inception's batch-256 set (``IFCBK_CONV_BIG=2``, ``IFCBK_WGRAD_PP=2``,
``IFCBK_CONV_FLAT=2``: "wherever the kernel applies"), through the C-ABI, against torch-CPU ``F.conv2d`` autograd (the reference's
arithmetic: aten::conv2d under neuston_models.py:66-68 for every backbone, with --batch of neuston_net.py:324): M tails, K tails
(K % (64*TN)), reduction tails (Kg % 64), 1x1 / 3x3 / 1x7 / 7x1 / stride-2 forward, padding, channel slices (ldx / ldy), every tile
(MT 8 / 10, TN 2..4, KH 4..6) and every epilogue of conv_common.h: statistics (MODE 0), eval affine (+residual, +ReLU), accumulate,
BN-backward sums of one producer (MODE 3), of a producer table (MODE 5), segmented destinations (MODE 4).  Each test asserts through
``ifcbk_op_kernel`` that the forced kernel is what the dispatcher picked."""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def _kname(ctx, d, kind, flags=0, residual=False):
    from ifcb_classifier_amd import _lib
    op = _lib.Op()
    op.kind = kind
    op.flags = flags
    op.u.conv = d
    if residual:
        op.p[5] = 1
    buf = C.create_string_buffer(128)
    ctx.lib.ifcbk_op_kernel(C.byref(op), buf, 128)
    return buf.value.decode()


@pytest.fixture
def forced(monkeypatch):
    def set_(**kw):
        for k in ('IFCBK_CONV_BIG', 'IFCBK_CONV_BIG_MT', 'IFCBK_CONV_BIG_TN', 'IFCBK_WGRAD_PP', 'IFCBK_WGRAD_PP_KH',
                  'IFCBK_CONV_FLAT', 'IFCBK_CONV_PP3', 'IFCBK_CONV_PP3_GRID', 'IFCBK_CONV_SLAB', 'IFCBK_WGRAD_FLAT'):
            monkeypatch.delenv(k, raising=False)
        for k, v in kw.items():
            monkeypatch.setenv(k, str(v))
    return set_


def _tensors(case, seed, ldx_extra=0, ldy_extra=0):
    """bf16-representable operands: NCHW fp32 on the CPU (the oracle's) and NHWC bf16 channel slices on the GPU"""
    N, Cc, H, W, K, R, S, sh, sw, ph, pw = case
    g = torch.Generator().manual_seed(seed)
    P, Q = (H + 2 * ph - R) // sh + 1, (W + 2 * pw - S) // sw + 1
    x = _bf(torch.randn(N, Cc, H, W, generator=g))
    w = _bf(torch.randn(K, Cc, R, S, generator=g) * (1.0 / (Cc * R * S) ** 0.5))
    dy = _bf(torch.randn(N, K, P, Q, generator=g))
    LDX, LDY = Cc + ldx_extra, K + ldy_extra
    xb = torch.full((N, H, W, LDX), float('nan'), dtype=torch.bfloat16)
    xb[..., ldx_extra // 2:ldx_extra // 2 + Cc] = x.permute(0, 2, 3, 1).to(torch.bfloat16)
    dyb = torch.full((N, P, Q, LDY), float('nan'), dtype=torch.bfloat16)
    dyb[..., ldy_extra // 2:ldy_extra // 2 + K] = dy.permute(0, 2, 3, 1).to(torch.bfloat16)
    xb, dyb = xb.cuda(), dyb.cuda()
    wk = w.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()                       # [K][R][S][C]
    wT = w.permute(1, 2, 3, 0).flip(1, 2).contiguous().to(torch.bfloat16).cuda()            # [C][R'][S'][K], flipped
    return x, w, dy, xb, dyb, wk, wT, P, Q, LDX, LDY


def _desc(case, P, Q, ldx, ldy):
    from ifcb_classifier_amd._lib import ConvDesc
    N, Cc, H, W, K, R, S, sh, sw, ph, pw = case
    return ConvDesc(N, H, W, Cc, ldx, K, R, S, sh, sw, ph, pw, P, Q, ldy, Cc, 0)


def _slice(t, off, n):
    return C.c_void_p(t.data_ptr() + 2 * off), t[..., off:off + n]


# N, C, H, W, K, R, S, sh, sw, ph, pw | forced MT, TN (0 = the plan's choice) | ldx / ldy padding
WIDE = [
    ((3, 192, 17, 17, 192, 1, 7, 1, 1, 0, 3), 10, 3, 0, 0),      # inception's own tile on an M tail (867 pixels)
    ((2, 160, 17, 17, 200, 7, 1, 1, 1, 3, 0), 8, 4, 16, 24),     # K tail (200 of 256), Kg tail (1120 = 17.5 steps), slices
    ((2, 768, 9, 9, 328, 1, 1, 1, 1, 0, 0), 8, 3, 0, 8),         # 1x1, K tail over two N tiles
    ((2, 96, 19, 19, 136, 3, 3, 2, 2, 0, 0), 10, 2, 8, 0),       # stride-2 forward (wide kernel), 136 of 2 x 128 channels
    ((5, 40, 15, 13, 72, 3, 3, 1, 1, 1, 1), 8, 2, 0, 0),         # narrow: 72 output channels, Kg = 360
    ((1, 256, 14, 14, 256, 3, 3, 1, 1, 1, 1), 0, 0, 0, 0),       # resnet50 layer3 3x3 (the plan's own tile)
    ((2, 64, 23, 9, 264, 5, 5, 1, 1, 2, 2), 10, 3, 8, 16),       # 5x5, K = 264: tail of 72 in the second tile
]


@pytest.mark.parametrize('case,mt,tn,lx,ly', WIDE)
def test_wide_tile_forward_and_input_gradient_forced(ctx, forced, case, mt, tn, lx, ly):
    env = dict(IFCBK_CONV_BIG=2, IFCBK_CONV_FLAT=0, IFCBK_CONV_PP3=0, IFCBK_CONV_SLAB=0)      # (conv_pp3 / conv_slab have their own tests below)
    if mt:
        env.update(IFCBK_CONV_BIG_MT=mt, IFCBK_CONV_BIG_TN=tn)
    forced(**env)
    _check_forward_and_input_gradient(ctx, case, lx, ly, 'conv_pp2<')


# the pixel-slab kernel (conv_slab.hip, round 5): 7-tap filters along either axis (row-major 1x7, column-major 7x1 tiles), M tails
# (867 = 2.7 tiles of 320; 578 = 1.8), channel chunks with a tail (160 = 64 + 64 + 32, 72 = 64 + 8), both column tiles (128 / 192),
# K tails (136 of 192, 104 of 128), channel slices of wider tensors, lines shorter and longer than inception's 17
SLAB = [
    ((3, 192, 17, 17, 192, 1, 7, 1, 1, 0, 3), 0, 0),         # inception's own 1x7 on an M tail
    ((3, 192, 17, 17, 192, 7, 1, 1, 1, 3, 0), 0, 0),         # ... and its 7x1: column-major tiles, transposed epilogue
    ((2, 160, 17, 17, 136, 7, 1, 1, 1, 3, 0), 16, 24),       # chunk tail (32 of 64), K tail, slices
    ((2, 128, 17, 17, 128, 1, 7, 1, 1, 0, 3), 8, 0),         # the 128-channel column tile
    ((4, 72, 12, 23, 104, 1, 7, 1, 1, 0, 3), 0, 8),          # 8-channel chunk tail, long lines, K tail of the 128-wide tile
    ((5, 64, 21, 11, 192, 7, 1, 1, 1, 3, 0), 0, 0),          # one chunk; column-major with 21-pixel lines
]


@pytest.mark.parametrize('case,lx,ly', SLAB)
def test_pixel_slab_kernel_forced(ctx, forced, case, lx, ly):
    forced(IFCBK_CONV_SLAB=2, IFCBK_CONV_BIG=0, IFCBK_CONV_FLAT=0, IFCBK_CONV_PP3=0)
    _check_forward_and_input_gradient(ctx, case, lx, ly, 'conv_slab<')


def test_pixel_slab_kernel_declines_what_its_slab_cannot_hold(ctx, forced):
    """a 1x7 filter over 9-pixel lines: 320 pixels touch 37 lines, 438 slab rows > the 400 of the instantiation -- the plan must say
    no (and the implicit GEMM serves the layer), never launch with a slab that is too short"""
    from ifcb_classifier_amd import _lib
    forced(IFCBK_CONV_SLAB=2, IFCBK_CONV_BIG=0, IFCBK_CONV_FLAT=0, IFCBK_CONV_PP3=0)
    case = (8, 64, 9, 9, 192, 1, 7, 1, 1, 0, 3)
    d = _desc(case, 9, 9, 64, 192)
    assert not _kname(ctx, d, _lib.OP_CONV_FWD).startswith('conv_slab<')
    _check_forward_and_input_gradient(ctx, case, 0, 0, 'conv_')


def _check_forward_and_input_gradient(ctx, case, lx, ly, prefix):
    from ifcb_classifier_amd import _lib
    N, Cc, H, W, K, R, S, sh, sw, ph, pw = case
    x, w, dy, xb, dyb, wk, wT, P, Q, LDX, LDY = _tensors(case, 11, lx, ly)
    d = _desc(case, P, Q, LDX, LDY)
    st = _lib.cur_stream()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, (sh, sw), (ph, pw))
    yr.backward(dy)
    yref = yr.detach().permute(0, 2, 3, 1)
    xp, _ = _slice(xb, lx // 2, Cc)
    # ---- forward + BatchNorm partial sums (MODE 0)
    assert _kname(ctx, d, _lib.OP_CONV_FWD).startswith(prefix), _kname(ctx, d, _lib.OP_CONV_FWD)
    yb = torch.full((N, P, Q, LDY), float('nan'), dtype=torch.bfloat16, device='cuda')
    yp, yv = _slice(yb, ly // 2, K)
    mb = ctx.lib.ifcbk_conv2d_fwd_mblocks(C.byref(d))
    part = torch.full((mb, 2, K), float('nan'), device='cuda')
    ctx.call('ifcbk_conv2d_fwd', C.byref(d), xp, _lib.ptr(wk), yp, _lib.ptr(part), st)
    torch.cuda.synchronize()
    yh = yv.float().cpu()
    assert _rel(yh, yref) < 3e-3
    assert torch.allclose(part[:, 0].sum(0).cpu(), yh.sum((0, 1, 2)), rtol=1e-4, atol=1e-3)
    assert torch.allclose(part[:, 1].sum(0).cpu(), (yh * yh).sum((0, 1, 2)), rtol=1e-4, atol=1e-3)
    if ly:
        assert torch.isnan(yb[..., :ly // 2].float()).all() and torch.isnan(yb[..., ly // 2 + K:].float()).all()
    # ---- eval epilogue: affine + residual + ReLU
    g = torch.Generator().manual_seed(3)
    scale, shift = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.3
    res = _bf(torch.randn(N, P, Q, K, generator=g))
    y2 = torch.full((N, P, Q, LDY), float('nan'), dtype=torch.bfloat16, device='cuda')
    y2p, y2v = _slice(y2, ly // 2, K)
    resd, scd, shd = res.to(torch.bfloat16).cuda(), scale.cuda(), shift.cuda()     # (device copies stay referenced until the sync)
    ctx.call('ifcbk_conv2d_fwd_affine', C.byref(d), xp, _lib.ptr(wk), y2p, _lib.ptr(scd), _lib.ptr(shd), _lib.ptr(resd), K, 1, st)
    torch.cuda.synchronize()
    want = torch.relu(_bf(yref) * scale + shift + res)          # the affine acts on the conv output as stored (rounded)
    assert _rel(y2v.float().cpu(), want) < 4e-3
    if sh != 1:
        return
    # ---- input gradient: first writer, accumulating, with the BN-backward sums of one producer (MODE 3)
    assert _kname(ctx, d, _lib.OP_CONV_DGRAD).startswith(prefix), _kname(ctx, d, _lib.OP_CONV_DGRAD)
    dyp, _ = _slice(dyb, ly // 2, K)
    dxb = torch.full((N, H, W, LDX), float('nan'), dtype=torch.bfloat16, device='cuda')
    dxp, dxv = _slice(dxb, lx // 2, Cc)
    ctx.call('ifcbk_conv2d_dgrad', C.byref(d), dyp, _lib.ptr(wT), dxp, 0, st)
    torch.cuda.synchronize()
    rdx = xr.grad.permute(0, 2, 3, 1)
    assert _rel(dxv.float().cpu(), rdx) < 3e-3
    first = dxv.float().cpu().clone()
    ctx.call('ifcbk_conv2d_dgrad', C.byref(d), dyp, _lib.ptr(wT), dxp, 1, st)
    torch.cuda.synchronize()
    assert _rel(dxv.float().cpu(), 2 * first) < 6e-3
    assert _kname(ctx, d, _lib.OP_CONV_DGRAD_BNSTAT).startswith(prefix) and _kname(ctx, d, _lib.OP_CONV_DGRAD_BNSTAT).endswith(', 3>')
    raw = _bf(torch.randn(N, H, W, Cc, generator=g) * 1.5)
    mean, invstd = torch.randn(Cc, generator=g) * 0.2, torch.rand(Cc, generator=g) + 0.5
    bsc, bsh = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    nrow = ctx.lib.ifcbk_conv2d_dgrad_bnstat_mblocks(C.byref(d))
    assert nrow > 0
    part2 = torch.full((nrow, 2, Cc), float('nan'), device='cuda')
    dx3 = torch.full((N, H, W, LDX), float('nan'), dtype=torch.bfloat16, device='cuda')
    dx3p, dx3v = _slice(dx3, lx // 2, Cc)
    rawd, dev = raw.to(torch.bfloat16).cuda(), [t.cuda() for t in (mean, invstd, bsc, bsh)]
    ctx.call('ifcbk_conv2d_dgrad_bnstat', C.byref(d), dyp, _lib.ptr(wT), dx3p, _lib.ptr(rawd), Cc, _lib.ptr(dev[0]), _lib.ptr(dev[1]),
             _lib.ptr(dev[2]), _lib.ptr(dev[3]), _lib.ptr(part2), st)
    torch.cuda.synchronize()
    dxs = dx3v.float().cpu()
    assert torch.equal(dxs, first)
    dz = torch.where(raw * bsc + bsh > 0, dxs, torch.zeros(()))
    xhat = (raw - mean) * invstd
    assert _rel(part2[:, 0].sum(0).cpu(), dz.sum((0, 1, 2))) < 1e-4
    assert _rel(part2[:, 1].sum(0).cpu(), (dz * xhat).sum((0, 1, 2))) < 1e-4


def test_wide_tile_producer_table_and_segmented_epilogues(ctx, forced):
    """MODE 5 (BN-backward sums of a CONCATENATION's producers through a per-chunk table) and MODE 4 (eval sibling GEMM: segments
    with their own destination, stride and affine-or-raw switch) on the wide-tile kernel, shapes with M and K tails."""
    from ifcb_classifier_amd import _lib
    forced(IFCBK_CONV_BIG=2, IFCBK_CONV_BIG_MT=8, IFCBK_CONV_BIG_TN=3, IFCBK_CONV_FLAT=0, IFCBK_CONV_SLAB=0)
    case = (3, 168, 13, 11, 152, 1, 1, 1, 1, 0, 0)
    N, Cc, H, W, K, R, S, sh, sw, ph, pw = case
    x, w, dy, xb, dyb, wk, wT, P, Q, LDX, LDY = _tensors(case, 5)
    d = _desc(case, P, Q, LDX, LDY)
    st = _lib.cur_stream()
    g = torch.Generator().manual_seed(9)
    # ---- MODE 5: dx channels [0,64) come from producer A, [64,104) have no BatchNorm producer (a pooled slice), [104,168) from B
    assert _kname(ctx, d, _lib.OP_CONV_DGRAD_BNSTAT_TAB).startswith('conv_pp2<') and _kname(ctx, d, _lib.OP_CONV_DGRAD_BNSTAT_TAB).endswith(', 5>')
    rawA = _bf(torch.randn(N, H, W, 64, generator=g)).to(torch.bfloat16).cuda()
    rawB = _bf(torch.randn(N, H, W, 80, generator=g)).to(torch.bfloat16).cuda()        # B's tensor is wider: its slice [8, 72) is used
    statA = torch.stack([torch.randn(64, generator=g) * 0.2, torch.rand(64, generator=g) + 0.5, torch.rand(64, generator=g) + 0.5,
                         torch.randn(64, generator=g) * 0.3]).cuda()                 # rows: mean, invstd, scale, shift
    statB = torch.stack([torch.randn(64, generator=g) * 0.2, torch.rand(64, generator=g) + 0.5, torch.rand(64, generator=g) + 0.5,
                         torch.randn(64, generator=g) * 0.3]).cuda()
    tab = (_lib.BsChunk * (Cc // 8))()
    for c8 in range(Cc // 8):
        c = c8 * 8
        if c < 64:
            tab[c8].raw, tab[c8].stat, tab[c8].raw_ld, tab[c8].stat_ld = rawA.data_ptr() + 2 * c, statA.data_ptr() + 4 * c, 64, 64
        elif c < 104:
            tab[c8].raw, tab[c8].stat, tab[c8].raw_ld, tab[c8].stat_ld = None, None, 0, 0
        else:
            cb = c - 104
            tab[c8].raw, tab[c8].stat, tab[c8].raw_ld, tab[c8].stat_ld = rawB.data_ptr() + 2 * (8 + cb), statB.data_ptr() + 4 * cb, 80, 64
    tabd = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).cuda()
    nrow = ctx.lib.ifcbk_conv2d_dgrad_bnstat_mblocks(C.byref(d))
    part = torch.full((nrow, 2, Cc), float('nan'), device='cuda')
    dx = torch.full((N, H, W, Cc), float('nan'), dtype=torch.bfloat16, device='cuda')
    ctx.call('ifcbk_conv2d_dgrad_bnstat_table', C.byref(d), _lib.ptr(dyb), _lib.ptr(wT), _lib.ptr(dx), _lib.ptr(tabd), _lib.ptr(part), st)
    torch.cuda.synchronize()
    rdx = torch.nn.grad.conv2d_input((N, Cc, H, W), w, dy, 1, 0).permute(0, 2, 3, 1)
    dxs = dx.float().cpu()
    assert _rel(dxs, rdx) < 3e-3
    s1, s2 = part[:, 0].sum(0).cpu(), part[:, 1].sum(0).cpu()
    for lo, hi, raw, stt in ((0, 64, rawA.float().cpu(), statA.cpu()), (104, 168, rawB.float().cpu()[..., 8:72], statB.cpu())):
        dz = torch.where(raw * stt[2] + stt[3] > 0, dxs[..., lo:hi], torch.zeros(()))
        assert _rel(s1[lo:hi], dz.sum((0, 1, 2))) < 1e-4
        assert _rel(s2[lo:hi], (dz * ((raw - stt[0]) * stt[1])).sum((0, 1, 2))) < 1e-4
    assert s1[64:104].abs().max().item() == 0 and s2[64:104].abs().max().item() == 0
    # ---- MODE 4: three segments (affine+ReLU into a slice, raw, affine+ReLU), their own tensors and strides
    assert _kname(ctx, d, _lib.OP_CONV_FWD_AFFINE_SEG).startswith('conv_pp2<') and _kname(ctx, d, _lib.OP_CONV_FWD_AFFINE_SEG).endswith(', 4>')
    ksegs = [64, 40, 48]
    ys = [torch.full((N, P, Q, 96), float('nan'), dtype=torch.bfloat16, device='cuda'),
          torch.full((N, P, Q, 40), float('nan'), dtype=torch.bfloat16, device='cuda'),
          torch.full((N, P, Q, 48), float('nan'), dtype=torch.bfloat16, device='cuda')]
    ptrs = (C.c_void_p * 3)(ys[0].data_ptr() + 2 * 16, ys[1].data_ptr(), ys[2].data_ptr())
    ldys = (C.c_int32 * 3)(96, 40, 48)
    ks = (C.c_int32 * 3)(*ksegs)
    aff = (C.c_int32 * 3)(1, 0, 1)
    scale, shift = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.3
    scd, shd = scale.cuda(), shift.cuda()
    ctx.call('ifcbk_conv2d_fwd_affine_segments', C.byref(d), _lib.ptr(xb), _lib.ptr(wk), 3, ptrs, ldys, ks, aff, _lib.ptr(scd), _lib.ptr(shd), st)
    torch.cuda.synchronize()
    yref = _bf(F.conv2d(x, w).permute(0, 2, 3, 1))
    act = torch.relu(yref * scale + shift)
    assert _rel(ys[0][..., 16:80].float().cpu(), act[..., :64]) < 4e-3
    assert torch.isnan(ys[0][..., :16].float()).all() and torch.isnan(ys[0][..., 80:].float()).all()
    assert _rel(ys[1].float().cpu(), yref[..., 64:104]) < 3e-3
    assert _rel(ys[2].float().cpu(), act[..., 104:152]) < 4e-3


WGRAD = [
    ((3, 192, 17, 17, 192, 7, 1, 1, 1, 3, 0), 6, 0, 0),          # inception's tile, pixel tail
    ((2, 96, 19, 19, 136, 3, 3, 2, 2, 0, 0), 5, 8, 16),          # stride 2, K tail (136 of 160), slices
    ((2, 768, 9, 9, 328, 1, 1, 1, 1, 0, 0), 4, 0, 8),            # 1x1, three K tiles of 128 with a tail, RSC = 768 = 3 column tiles
    ((5, 40, 15, 13, 72, 3, 3, 1, 1, 1, 1), 4, 0, 0),            # one narrow tile, RSC = 360 (column tail)
    ((4, 64, 23, 9, 200, 5, 5, 1, 1, 2, 2), 6, 8, 0),            # 5x5 with padding, K tail (200 of 2 x 192)
]


@pytest.mark.parametrize('case,kh,lx,ly', WGRAD)
def test_wide_tile_weight_gradient_forced(ctx, forced, case, kh, lx, ly):
    from ifcb_classifier_amd import _lib
    forced(IFCBK_WGRAD_PP=2, IFCBK_WGRAD_PP_KH=kh)
    N, Cc, H, W, K, R, S, sh, sw, ph, pw = case
    x, w, dy, xb, dyb, wk, wT, P, Q, LDX, LDY = _tensors(case, 21, lx, ly)
    d = _desc(case, P, Q, LDX, LDY)
    assert _kname(ctx, d, _lib.OP_CONV_WGRAD) == 'conv_wgrad_pp<%d, 0>' % kh, _kname(ctx, d, _lib.OP_CONV_WGRAD)
    st = _lib.cur_stream()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    F.conv2d(xr, wr, None, (sh, sw), (ph, pw)).backward(dy)
    rdw = wr.grad.permute(0, 2, 3, 1)
    xp, _ = _slice(xb, lx // 2, Cc)
    dyp, _ = _slice(dyb, ly // 2, K)
    ctx.reserve(ctx.lib.ifcbk_conv2d_wgrad_workspace(C.byref(d)))
    dw = torch.full((K, R, S, Cc), float('nan'), device='cuda')
    ctx.call('ifcbk_conv2d_wgrad', C.byref(d), xp, dyp, _lib.ptr(dw), 0, st)
    torch.cuda.synchronize()
    assert _rel(dw.cpu(), rdw) < 1e-4
    err = (dw.cpu() - rdw).abs().max().item()
    assert err <= 2e-5 * max(1.0, rdw.abs().max().item()) * (N * P * Q) ** 0.5, err
    ctx.call('ifcbk_conv2d_wgrad', C.byref(d), xp, dyp, _lib.ptr(dw), 1, st)      # accumulate
    torch.cuda.synchronize()
    assert _rel(dw.cpu(), 2 * rdw) < 1e-4


# the flat-slot weight gradient (conv_wgrad_flat.hip, round 5): one filter row per block, the x slab shared by the row's taps.  3x3
# and 5x5 with padding (band slots between rows / images), the unpadded Conv2d_4a shape (two K tiles of 96), K and column tails
# (K = 40: 48-channel tile; 15 column tiles: an odd count over the two wave columns), slices, pixel counts off the 64-slot step
WFLAT = [
    ((4, 64, 19, 17, 96, 3, 3, 1, 1, 1, 1), 0, 0),               # the 35x35 stage's 64 -> 96
    ((3, 96, 14, 21, 96, 3, 3, 1, 1, 1, 1), 8, 16),              # 96 -> 96: 18 column tiles, slices
    ((3, 48, 15, 15, 64, 5, 5, 1, 1, 2, 2), 0, 8),               # 5x5: 15 column tiles, five filter rows
    ((2, 80, 23, 25, 192, 3, 3, 1, 1, 0, 0), 0, 0),              # Conv2d_4a: no padding, two K tiles
    ((3, 64, 13, 13, 40, 3, 3, 1, 1, 1, 1), 0, 0),               # K tail: 40 of a 48-channel tile
    ((2, 16, 10, 11, 24, 3, 3, 1, 1, 1, 1), 8, 0),               # tiny: 3 column tiles, one wave column partly empty
    ((2, 32, 9, 30, 104, 1, 7, 1, 1, 0, 3), 0, 0),               # one filter row of seven taps, K = 104: two K tiles of 64 (52 -> 64)
]


@pytest.mark.parametrize('case,lx,ly', WFLAT)
def test_flat_slot_weight_gradient_forced(ctx, forced, case, lx, ly):
    from ifcb_classifier_amd import _lib
    forced(IFCBK_WGRAD_FLAT=2)
    N, Cc, H, W, K, R, S, sh, sw, ph, pw = case
    x, w, dy, xb, dyb, wk, wT, P, Q, LDX, LDY = _tensors(case, 41, lx, ly)
    d = _desc(case, P, Q, LDX, LDY)
    assert _kname(ctx, d, _lib.OP_CONV_WGRAD) == 'conv_wgrad_flat', _kname(ctx, d, _lib.OP_CONV_WGRAD)
    st = _lib.cur_stream()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    F.conv2d(xr, wr, None, (sh, sw), (ph, pw)).backward(dy)
    rdw = wr.grad.permute(0, 2, 3, 1)
    xp, _ = _slice(xb, lx // 2, Cc)
    dyp, _ = _slice(dyb, ly // 2, K)
    ctx.reserve(ctx.lib.ifcbk_conv2d_wgrad_workspace(C.byref(d)))
    outs = []
    for rep in range(2):                                               # (twice: bitwise repeatable)
        dw = torch.full((K, R, S, Cc), float('nan'), device='cuda')
        ctx.call('ifcbk_conv2d_wgrad', C.byref(d), xp, dyp, _lib.ptr(dw), 0, st)
        torch.cuda.synchronize()
        outs.append(dw.cpu())
    assert torch.equal(outs[0], outs[1])
    assert _rel(outs[0], rdw) < 1e-4
    err = (outs[0] - rdw).abs().max().item()
    assert err <= 2e-5 * max(1.0, rdw.abs().max().item()) * (N * P * Q) ** 0.5, err
    ctx.call('ifcbk_conv2d_wgrad', C.byref(d), xp, dyp, _lib.ptr(dw), 1, st)      # accumulate
    torch.cuda.synchronize()
    assert _rel(dw.cpu(), 2 * rdw) < 1e-4


def test_flat_slot_weight_gradients_as_one_group(ctx, forced):
    """three layers of one block (5x5 48 -> 64, 3x3 64 -> 96, 3x3 96 -> 96) as ONE grid + one reduce: equal, bit for bit, to their
    single launches (same per-block sums, same reduce order); ifcbk_op_kernel names the grouped kernel"""
    from ifcb_classifier_amd import _lib
    forced(IFCBK_WGRAD_FLAT=2, IFCBK_WGRAD_PP=2)
    cases = [(6, 48, 15, 15, 64, 5, 5, 1, 1, 2, 2), (6, 64, 15, 15, 96, 3, 3, 1, 1, 1, 1), (6, 96, 15, 15, 96, 3, 3, 1, 1, 1, 1)]
    n = len(cases)
    st = _lib.cur_stream()
    descs = (_lib.ConvDesc * n)()
    xs, dys, dws, keep, single = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)(), [], []
    for i, case in enumerate(cases):
        N, Cc, H, W, K, R, S, sh, sw, ph, pw = case
        x, w, dy, xb, dyb, wk, wT, P, Q, LDX, LDY = _tensors(case, 50 + i)
        d = _desc(case, P, Q, LDX, LDY)
        descs[i] = d
        assert ctx.lib.ifcbk_conv2d_wgrad_group_member_kh(C.byref(d)) == 16
        ctx.reserve(ctx.lib.ifcbk_conv2d_wgrad_workspace(C.byref(d)))
        dw1 = torch.full((K, R, S, Cc), float('nan'), device='cuda')
        ctx.call('ifcbk_conv2d_wgrad', C.byref(d), _lib.ptr(xb), _lib.ptr(dyb), _lib.ptr(dw1), 0, st)
        torch.cuda.synchronize()
        single.append(dw1.cpu())
        dwg = torch.full((K, R, S, Cc), float('nan'), device='cuda')
        xs[i], dys[i], dws[i] = xb.data_ptr(), dyb.data_ptr(), dwg.data_ptr()
        keep.append((xb, dyb, dwg))
    need = ctx.lib.ifcbk_conv2d_wgrad_group_workspace(n, descs)
    assert need > 0
    ctx.reserve(need)
    kh = C.c_int(0)
    assert ctx.lib.ifcbk_conv2d_wgrad_group_info(n, descs, C.byref(kh), None, None) == 0 and kh.value == 16
    ctx.call('ifcbk_conv2d_wgrad_group', n, descs, xs, dys, dws, 0, st)
    torch.cuda.synchronize()
    for i in range(n):
        got = keep[i][2].cpu()
        assert _rel(got, single[i]) < 1e-6
        # (the split counts of the group differ from the single launches': the same products summed in another split order)


# conv_pp3 (round 4): the PERSISTENT wide-tile kernel -- seamless tile switch, deferred register-direct stores, statistics by DPP row
# sums, eval affine from LDS-staged coefficients.  grid: blocks of the launch (IFCBK_CONV_PP3_GRID test hook): with 2-3 blocks every
# block walks several tiles, so the tile boundary (parity running on, gather switch, pack, deferred stores of tile i inside tile
# i + 1) is what is tested; 0 = the launch's own grid (one tile per block on these sizes: only the exposed last-tile epilogue).
PP3 = [
    ((9, 192, 17, 17, 192, 1, 7, 1, 1, 0, 3), 3, 0, 0),          # 11 tiles on 3 blocks, inception's 1x7, M tail (2601 pixels), SPP 1
    ((6, 160, 17, 17, 200, 7, 1, 1, 1, 3, 0), 2, 16, 24),        # two N tiles (K tail 8 of 192), Kg tail, channel slices, SPP 1
    ((4, 768, 17, 17, 384, 1, 1, 1, 1, 0, 0), 3, 0, 8),          # plain 1x1 GEMM (the sibling-GEMM shape), 5 M tiles x 2 N tiles, SPP 1
    ((3, 96, 23, 19, 136, 3, 3, 2, 2, 0, 0), 2, 8, 0),           # stride-2 forward, 136 of 192 channels, SPP 1 (nk = 14)
    ((5, 40, 15, 13, 72, 3, 3, 1, 1, 1, 1), 2, 0, 0),            # nk = 6: SPP 2
    ((6, 256, 9, 9, 264, 1, 1, 1, 1, 0, 0), 0, 0, 0),            # nk = 4, the launch's own grid
    ((8, 192, 12, 12, 192, 1, 1, 1, 1, 0, 0), 2, 0, 0),          # nk = 3 (the shortest tile the burst allows), 5 tiles on 2 blocks
    ((2, 64, 40, 31, 192, 3, 3, 1, 1, 1, 1), 5, 0, 16),          # 10 tiles on 5 blocks: two tiles each
]


@pytest.mark.parametrize('case,grid,lx,ly', PP3)
def test_persistent_wide_tile_kernel_forced(ctx, forced, case, grid, lx, ly):
    from ifcb_classifier_amd import _lib
    env = dict(IFCBK_CONV_PP3=2, IFCBK_CONV_FLAT=0, IFCBK_CONV_BIG=0, IFCBK_CONV_SLAB=0)
    if grid:
        env['IFCBK_CONV_PP3_GRID'] = grid
    forced(**env)
    N, Cc, H, W, K, R, S, sh, sw, ph, pw = case
    x, w, dy, xb, dyb, wk, wT, P, Q, LDX, LDY = _tensors(case, 31, lx, ly)
    d = _desc(case, P, Q, LDX, LDY)
    st = _lib.cur_stream()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, (sh, sw), (ph, pw))
    yr.backward(dy)
    yref = yr.detach().permute(0, 2, 3, 1)
    xp, _ = _slice(xb, lx // 2, Cc)
    # ---- forward + BatchNorm partial sums: two partial rows per 256-pixel tile
    assert _kname(ctx, d, _lib.OP_CONV_FWD).startswith('conv_pp3<3, 8, 4, '), _kname(ctx, d, _lib.OP_CONV_FWD)
    yb = torch.full((N, P, Q, LDY), float('nan'), dtype=torch.bfloat16, device='cuda')
    yp, yv = _slice(yb, ly // 2, K)
    mb = ctx.lib.ifcbk_conv2d_fwd_mblocks(C.byref(d))
    assert mb == 2 * ((N * P * Q + 255) // 256)
    part = torch.full((mb, 2, K), float('nan'), device='cuda')
    ctx.call('ifcbk_conv2d_fwd', C.byref(d), xp, _lib.ptr(wk), yp, _lib.ptr(part), st)
    torch.cuda.synchronize()
    yh = yv.float().cpu()
    assert _rel(yh, yref) < 3e-3
    assert not torch.isnan(part).any()
    assert torch.allclose(part[:, 0].sum(0).cpu(), yh.sum((0, 1, 2)), rtol=1e-4, atol=1e-3)
    assert torch.allclose(part[:, 1].sum(0).cpu(), (yh * yh).sum((0, 1, 2)), rtol=1e-4, atol=1e-3)
    if ly:
        assert torch.isnan(yb[..., :ly // 2].float()).all() and torch.isnan(yb[..., ly // 2 + K:].float()).all()
    y1 = yb.clone()
    yb.fill_(float('nan'))
    part.fill_(float('nan'))
    ctx.call('ifcbk_conv2d_fwd', C.byref(d), xp, _lib.ptr(wk), yp, _lib.ptr(part), st)
    torch.cuda.synchronize()
    assert torch.equal(yb.view(torch.int16), y1.view(torch.int16))          # bitwise repeatable (NaN padding included)
    # ---- without statistics (a raw store only)
    yb.fill_(float('nan'))
    ctx.call('ifcbk_conv2d_fwd', C.byref(d), xp, _lib.ptr(wk), yp, None, st)
    torch.cuda.synchronize()
    assert torch.equal(yb.view(torch.int16), y1.view(torch.int16))
    # ---- eval epilogue: affine (+ReLU, and without)
    g = torch.Generator().manual_seed(3)
    scale, shift = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.3
    scd, shd = scale.cuda(), shift.cuda()
    for relu in (1, 0):
        assert _kname(ctx, d, _lib.OP_CONV_FWD_AFFINE).startswith('conv_pp3<3, 8, 4, 1, '), _kname(ctx, d, _lib.OP_CONV_FWD_AFFINE)
        y2 = torch.full((N, P, Q, LDY), float('nan'), dtype=torch.bfloat16, device='cuda')
        y2p, y2v = _slice(y2, ly // 2, K)
        ctx.call('ifcbk_conv2d_fwd_affine', C.byref(d), xp, _lib.ptr(wk), y2p, _lib.ptr(scd), _lib.ptr(shd), None, 0, relu, st)
        torch.cuda.synchronize()
        want = _bf(yref) * scale + shift                         # the affine acts on the conv output as stored (rounded), as in conv_pp2
        if relu:
            want = torch.relu(want)
        assert _rel(y2v.float().cpu(), want) < 4e-3
        # ... bit for bit what the LDS-staged epilogue of conv_pp2 / conv_igemm writes: a RUN batch gets the same scores whichever
        # kernel its size selects
        forced(IFCBK_CONV_PP3=0, IFCBK_CONV_FLAT=0, IFCBK_CONV_SLAB=0)
        y3 = torch.full((N, P, Q, LDY), float('nan'), dtype=torch.bfloat16, device='cuda')
        y3p, y3v = _slice(y3, ly // 2, K)
        ctx.call('ifcbk_conv2d_fwd_affine', C.byref(d), xp, _lib.ptr(wk), y3p, _lib.ptr(scd), _lib.ptr(shd), None, 0, relu, st)
        torch.cuda.synchronize()
        assert torch.equal(y3v, y2v)
        forced(**env)
        if ly:
            assert torch.isnan(y2[..., :ly // 2].float()).all() and torch.isnan(y2[..., ly // 2 + K:].float()).all()
    if sh != 1:
        return
    # ---- first-writer input gradient (a conv over dy with the flipped filter)
    assert _kname(ctx, d, _lib.OP_CONV_DGRAD).startswith('conv_pp3<3, 8, 4, '), _kname(ctx, d, _lib.OP_CONV_DGRAD)
    dyp, _ = _slice(dyb, ly // 2, K)
    dxb = torch.full((N, H, W, LDX), float('nan'), dtype=torch.bfloat16, device='cuda')
    dxp, dxv = _slice(dxb, lx // 2, Cc)
    ctx.call('ifcbk_conv2d_dgrad', C.byref(d), dyp, _lib.ptr(wT), dxp, 0, st)
    torch.cuda.synchronize()
    assert _rel(dxv.float().cpu(), xr.grad.permute(0, 2, 3, 1)) < 3e-3
    if lx:
        assert torch.isnan(dxb[..., :lx // 2].float()).all() and torch.isnan(dxb[..., lx // 2 + Cc:].float()).all()


# grouped weight gradients (ifcbk_conv2d_wgrad_group, round 4): members with DIFFERENT filter shapes, maps, strides, tails and slices in
# one split-K grid + one reduce; every member against torch's autograd, bitwise repeatable, accumulate, and the plan's own numbers
WGROUPS = [
    (6, [((3, 192, 17, 17, 192, 1, 7, 1, 1, 0, 3), 0, 0), ((3, 192, 17, 17, 192, 7, 1, 1, 1, 3, 0), 8, 16),
         ((2, 96, 19, 19, 136, 3, 3, 2, 2, 0, 0), 8, 0)]),                                   # 1x7 + 7x1 + stride-2 3x3 with a K tail
    (5, [((2, 160, 17, 17, 160, 1, 7, 1, 1, 0, 3), 0, 0), ((2, 160, 17, 17, 160, 7, 1, 1, 1, 3, 0), 0, 0)]),   # the c7 = 160 pair of Mixed_6c / 6d
    (4, [((2, 768, 9, 9, 328, 1, 1, 1, 1, 0, 0), 0, 8), ((5, 40, 15, 13, 72, 3, 3, 1, 1, 1, 1), 0, 0),
         ((4, 64, 23, 9, 120, 5, 5, 1, 1, 2, 2), 8, 0), ((1, 128, 8, 8, 128, 3, 1, 1, 1, 1, 0), 0, 0)]),   # four members, every kind of tail
]


@pytest.mark.parametrize('kh,members', WGROUPS)
def test_grouped_weight_gradient_forced(ctx, forced, kh, members):
    from ifcb_classifier_amd import _lib
    forced(IFCBK_WGRAD_PP=2, IFCBK_WGRAD_PP_KH=kh)
    n = len(members)
    st = _lib.cur_stream()
    descs = (_lib.ConvDesc * n)()
    xs, dys, dws = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
    keep, refs, outs = [], [], []
    for i, (case, lx, ly) in enumerate(members):
        N, Cc, H, W, K, R, S, sh, sw, ph, pw = case
        x, w, dy, xb, dyb, wk, wT, P, Q, LDX, LDY = _tensors(case, 40 + i, lx, ly)
        descs[i] = _desc(case, P, Q, LDX, LDY)
        assert ctx.lib.ifcbk_conv2d_wgrad_group_member_kh(C.byref(descs[i])) == kh
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        F.conv2d(xr, wr, None, (sh, sw), (ph, pw)).backward(dy)
        refs.append((wr.grad.permute(0, 2, 3, 1), N * P * Q))
        xp, _ = _slice(xb, lx // 2, Cc)
        dyp, _ = _slice(dyb, ly // 2, K)
        dw = torch.full((K, R, S, Cc), float('nan'), device='cuda')
        xs[i], dys[i], dws[i] = xp.value, dyp.value, dw.data_ptr()
        keep += [xb, dyb]
        outs.append(dw)
    need = ctx.lib.ifcbk_conv2d_wgrad_group_workspace(n, descs)
    assert need > 0
    khq, blocks, ns = C.c_int(), C.c_int(), (C.c_int * n)()
    assert ctx.lib.ifcbk_conv2d_wgrad_group_info(n, descs, C.byref(khq), C.byref(blocks), ns) == 0
    assert khq.value == kh and blocks.value >= n and all(v >= 1 for v in ns)
    ctx.reserve(need)
    # the op-table form names the grouped kernel
    items = (_lib.WgradItem * n)()
    for i in range(n):
        items[i].d, items[i].x, items[i].dy, items[i].dw = descs[i], xs[i], dys[i], dws[i]
    op = _lib.Op()
    op.kind, op.p[0], op.i[0] = _lib.OP_CONV_WGRAD_GROUP, C.addressof(items), n
    buf = C.create_string_buffer(64)
    ctx.lib.ifcbk_op_kernel(C.byref(op), buf, 64)
    assert buf.value.decode() == 'conv_wgrad_ppg<%d>' % kh
    ctx.call('ifcbk_conv2d_wgrad_group', n, descs, xs, dys, dws, 0, st)
    torch.cuda.synchronize()
    first = [o.clone() for o in outs]
    for o, (rdw, M) in zip(outs, refs):
        assert _rel(o.cpu(), rdw) < 1e-4
        err = (o.cpu() - rdw).abs().max().item()
        assert err <= 2e-5 * max(1.0, rdw.abs().max().item()) * M ** 0.5, err
    for o in outs:
        o.fill_(float('nan'))
    ctx.run_program((_lib.Op * 1)(op), 1, st)                # the same launch through the program runner
    torch.cuda.synchronize()
    for o, f in zip(outs, first):
        assert torch.equal(o, f)                             # fixed split plan, fixed reduction order
    ctx.call('ifcbk_conv2d_wgrad_group', n, descs, xs, dys, dws, 1, st)      # accumulate
    torch.cuda.synchronize()
    for o, (rdw, M) in zip(outs, refs):
        assert _rel(o.cpu(), 2 * rdw) < 1e-4
    # a member with another channel tile does not join
    bad = (_lib.ConvDesc * 2)(descs[0], _desc((2, 64, 9, 9, 64 if kh != 4 else 192, 3, 3, 1, 1, 1, 1), 9, 9, 64, 64 if kh != 4 else 192))
    if ctx.lib.ifcbk_conv2d_wgrad_group_member_kh(C.byref(bad[1])) != kh:
        assert ctx.lib.ifcbk_conv2d_wgrad_group_workspace(2, bad) == 0


# conv_flat: N, C, H, W, K, R, S, ph, pw | ldx / ldy padding
FLAT = [
    ((3, 96, 17, 13, 96, 3, 3, 1, 1), 16, 8),      # odd map, slices
    ((5, 64, 19, 23, 96, 3, 3, 0, 0), 0, 0),       # unpadded (P = H - 2)
    ((4, 96, 8, 8, 96, 3, 3, 2, 1), 0, 8),         # asymmetric padding, tiny map
    ((7, 48, 12, 12, 64, 5, 5, 1, 2), 8, 0),       # 5x5: only its input gradient (64 -> 48) has a flat form
    ((40, 64, 35, 35, 96, 3, 3, 1, 1), 0, 0),      # 270 segments on 256 blocks: the persistent loop, deferred stores, image double buffer
]


@pytest.mark.parametrize('case,lx,ly', FLAT)
def test_flat_image_kernel_forced(ctx, forced, case, lx, ly):
    from ifcb_classifier_amd import _lib
    forced(IFCBK_CONV_FLAT=2, IFCBK_CONV_BIG=0)
    N, Cc, H, W, K, R, S, ph, pw = case
    full = (N, Cc, H, W, K, R, S, 1, 1, ph, pw)
    x, w, dy, xb, dyb, wk, wT, P, Q, LDX, LDY = _tensors(full, 31, lx, ly)
    d = _desc(full, P, Q, LDX, LDY)
    st = _lib.cur_stream()
    big = N >= 32
    if big:     # the reference of the one large case on the GPU (torch fp32 convolution of the same bf16 operands)
        xg, wg, dyg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), dy.cuda()
        yg = F.conv2d(xg, wg, None, 1, (ph, pw))
        yg.backward(dyg)
        yref, rdx = yg.detach().permute(0, 2, 3, 1).cpu(), xg.grad.permute(0, 2, 3, 1).cpu()
    else:
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        yr = F.conv2d(xr, wr, None, 1, (ph, pw))
        yr.backward(dy)
        yref, rdx = yr.detach().permute(0, 2, 3, 1), xr.grad.permute(0, 2, 3, 1)
    xp, _ = _slice(xb, lx // 2, Cc)
    dyp, _ = _slice(dyb, ly // 2, K)
    g = torch.Generator().manual_seed(4)
    if (Cc, K, R) != (48, 64, 5):
        # ---- forward + statistics, eval affine + ReLU
        assert _kname(ctx, d, _lib.OP_CONV_FWD).startswith('conv_flat<'), _kname(ctx, d, _lib.OP_CONV_FWD)
        yb = torch.full((N, P, Q, LDY), float('nan'), dtype=torch.bfloat16, device='cuda')
        yp, yv = _slice(yb, ly // 2, K)
        mb = ctx.lib.ifcbk_conv2d_fwd_mblocks(C.byref(d))
        part = torch.full((mb, 2, K), float('nan'), device='cuda')
        ctx.call('ifcbk_conv2d_fwd', C.byref(d), xp, _lib.ptr(wk), yp, _lib.ptr(part), st)
        torch.cuda.synchronize()
        yh = yv.float().cpu()
        assert _rel(yh, yref) < 3e-3
        assert torch.allclose(part[:, 0].sum(0).cpu(), yh.sum((0, 1, 2)), rtol=1e-4, atol=1e-3)
        assert torch.allclose(part[:, 1].sum(0).cpu(), (yh * yh).sum((0, 1, 2)), rtol=1e-4, atol=1e-3)
        if ly:
            assert torch.isnan(yb[..., :ly // 2].float()).all() and torch.isnan(yb[..., ly // 2 + K:].float()).all()
        scale, shift = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.3
        y2 = torch.full((N, P, Q, LDY), float('nan'), dtype=torch.bfloat16, device='cuda')
        y2p, y2v = _slice(y2, ly // 2, K)
        scd, shd = scale.cuda(), shift.cuda()
        ctx.call('ifcbk_conv2d_fwd_affine', C.byref(d), xp, _lib.ptr(wk), y2p, _lib.ptr(scd), _lib.ptr(shd), None, 0, 1, st)
        torch.cuda.synchronize()
        assert _rel(y2v.float().cpu(), torch.relu(_bf(yref) * scale + shift)) < 4e-3
    # ---- input gradient (the flat kernel in the swapped role: K -> C channels), plain and with the BN-backward sums (MODE 3)
    names = [_kname(ctx, d, _lib.OP_CONV_DGRAD), _kname(ctx, d, _lib.OP_CONV_DGRAD_BNSTAT)]
    assert (K, Cc, R) in ((64, 48, 5), (96, 64, 3), (96, 96, 3), (64, 96, 3))      # the input-gradient roles conv_flat.hip builds
    assert names[0].startswith('conv_flat<') and names[1].startswith('conv_flat<') and names[1].endswith(', 3>'), names
    dxb = torch.full((N, H, W, LDX), float('nan'), dtype=torch.bfloat16, device='cuda')
    dxp, dxv = _slice(dxb, lx // 2, Cc)
    ctx.call('ifcbk_conv2d_dgrad', C.byref(d), dyp, _lib.ptr(wT), dxp, 0, st)
    torch.cuda.synchronize()
    first = dxv.float().cpu().clone()
    assert _rel(first, rdx) < 3e-3
    if lx:
        assert torch.isnan(dxb[..., :lx // 2].float()).all() and torch.isnan(dxb[..., lx // 2 + Cc:].float()).all()
    raw = _bf(torch.randn(N, H, W, Cc, generator=g) * 1.5)
    mean, invstd = torch.randn(Cc, generator=g) * 0.2, torch.rand(Cc, generator=g) + 0.5
    bsc, bsh = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    nrow = ctx.lib.ifcbk_conv2d_dgrad_bnstat_mblocks(C.byref(d))
    assert nrow > 0
    part2 = torch.full((nrow, 2, Cc), float('nan'), device='cuda')
    dx3 = torch.full((N, H, W, LDX), float('nan'), dtype=torch.bfloat16, device='cuda')
    dx3p, dx3v = _slice(dx3, lx // 2, Cc)
    rawd, dev = raw.to(torch.bfloat16).cuda(), [t.cuda() for t in (mean, invstd, bsc, bsh)]
    ctx.call('ifcbk_conv2d_dgrad_bnstat', C.byref(d), dyp, _lib.ptr(wT), dx3p, _lib.ptr(rawd), Cc, _lib.ptr(dev[0]), _lib.ptr(dev[1]),
             _lib.ptr(dev[2]), _lib.ptr(dev[3]), _lib.ptr(part2), st)
    torch.cuda.synchronize()
    dxs = dx3v.float().cpu()
    assert torch.equal(dxs, first)
    dz = torch.where(raw * bsc + bsh > 0, dxs, torch.zeros(()))
    assert _rel(part2[:, 0].sum(0).cpu(), dz.sum((0, 1, 2))) < 1e-4
    assert _rel(part2[:, 1].sum(0).cpu(), (dz * ((raw - mean) * invstd)).sum((0, 1, 2))) < 1e-4
    # results do not depend on the launch: twice the same bits (deferred stores, per-block statistics in a fixed order)
    part3 = torch.full((nrow, 2, Cc), float('nan'), device='cuda')
    dx4 = torch.full((N, H, W, LDX), float('nan'), dtype=torch.bfloat16, device='cuda')
    dx4p, dx4v = _slice(dx4, lx // 2, Cc)
    ctx.call('ifcbk_conv2d_dgrad_bnstat', C.byref(d), dyp, _lib.ptr(wT), dx4p, _lib.ptr(rawd), Cc, _lib.ptr(dev[0]), _lib.ptr(dev[1]),
             _lib.ptr(dev[2]), _lib.ptr(dev[3]), _lib.ptr(part3), st)
    torch.cuda.synchronize()
    assert torch.equal(dx4v, dx3v) and torch.equal(part3, part2)
