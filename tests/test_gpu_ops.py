"""C-ABI unit parity on the GPU: ROI preprocessing (bit-exact vs Pillow and the oracle restatement),
BatchNorm, pooling, head, loss and Adam vs the node-level CPU oracle (oracle/ops.py)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()


def nchw(y):
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


def _roi_batch(rois, S, ctx, in_ch=1, flips=None, mean=None, std=None, want_u8=True):
    from ifcb_classifier_amd import _lib
    n = len(rois)
    hs = torch.tensor([r.shape[0] for r in rois], dtype=torch.int32)
    ws = torch.tensor([r.shape[1] for r in rois], dtype=torch.int32)
    sizes = hs.long() * ws.long() * in_ch
    offs = torch.zeros(n, dtype=torch.int64)
    offs[1:] = torch.cumsum(sizes, 0)[:-1]
    pix = torch.from_numpy(np.concatenate([np.ascontiguousarray(r).reshape(-1) for r in rois])).cuda()
    d = _lib.RoiDesc()
    d.n_img, d.S, d.in_channels, d.out_channels, d.dtype = n, S, in_ch, 8, 0
    d.flip_bits_valid = 1 if flips is not None else 0
    for k in range(3):
        d.mean[k] = 0.0 if mean is None else mean[k]
        d.std[k] = 1.0 if std is None else std[k]
        d.tin_scale[k], d.tin_shift[k] = 1.0, 0.0
    mh, mw = int(hs.max()), int(ws.max())
    ctx.reserve(ctx.lib.ifcbk_roi_preprocess_workspace(C.byref(d), mh, mw))
    out = torch.zeros(n, S, S, 8, dtype=torch.bfloat16, device='cuda')
    u8 = torch.zeros(n, S, S, in_ch, dtype=torch.uint8, device='cuda') if want_u8 else None
    fl = torch.tensor(flips, dtype=torch.uint8).cuda() if flips is not None else None
    offs_d, hs_d, ws_d = offs.cuda(), hs.cuda(), ws.cuda()         # keep device operands alive across the launch
    ctx.call('ifcbk_roi_preprocess', C.byref(d), _lib.ptr(pix), _lib.ptr(offs_d), _lib.ptr(hs_d),
             _lib.ptr(ws_d), _lib.ptr(fl), mh, mw, _lib.ptr(out), _lib.ptr(u8), _lib.cur_stream())
    torch.cuda.synchronize()
    return out, u8


@pytest.mark.parametrize('S', [299, 224])
def test_roi_resize_bit_exact_vs_pil_golden(ctx, S):
    z = np.load(os.path.join(GOLD, 'pil_resize_cases.npz'))
    meta = [c for c in json.load(open(os.path.join(GOLD, 'pil_resize_cases.json')))['cases'] if c['S'] == S]
    rois = [z['in_%d' % c['case']] for c in meta]
    out, u8 = _roi_batch(rois, S, ctx)
    u8 = u8.cpu().numpy()
    for i, c in enumerate(meta):
        assert hashlib.sha256(u8[i, :, :, 0].tobytes()).hexdigest() == c['sha256'], c
    # float output = u8/255 rounded to bf16, 3 identical channels, 5 zero pad channels
    o = out.float().cpu()
    ref = _bf(torch.from_numpy(u8[..., 0].astype(np.float32)) / 255.0)
    for ch in range(3):
        assert torch.equal(o[..., ch], ref)
    assert o[..., 3:].abs().max().item() == 0


def test_roi_flips_normalize_and_rgb_vs_oracle(ctx):
    from oracle.pil_resize import roi_to_tensor, resize_bilinear_u8
    rng = np.random.default_rng(11)
    rois = [rng.integers(0, 256, (h, w), dtype=np.uint8) for h, w in ((40, 90), (333, 61), (77, 77), (12, 500))]
    flips = [0, 1, 2, 3]
    mean, std = [0.4, 0.5, 0.6], [0.2, 0.3, 0.25]
    out, u8 = _roi_batch(rois, 299, ctx, flips=flips, mean=mean, std=std)
    o = out.float().cpu()
    for i, r in enumerate(rois):
        ref = torch.from_numpy(roi_to_tensor(r, 299, mean, std, flip_v=bool(flips[i] & 1), flip_h=bool(flips[i] & 2)))
        assert torch.equal(o[i, :, :, :3].permute(2, 0, 1), _bf(ref)), i
    # interleaved RGB input (ImageDataset / NeustonDataset path)
    z = np.load(os.path.join(GOLD, 'pil_resize_cases.npz'))
    out, u8 = _roi_batch([z['rgb_in']], 299, ctx, in_ch=3)
    assert np.array_equal(u8.cpu().numpy()[0], z['rgb_out_299'])
    assert np.array_equal(resize_bilinear_u8(z['rgb_in'], 299, 299), z['rgb_out_299'])


def test_roi_empty_batch_is_a_noop(ctx):
    from ifcb_classifier_amd import _lib
    d = _lib.RoiDesc()
    d.n_img, d.S, d.in_channels, d.out_channels = 0, 299, 1, 8
    ctx.call('ifcbk_roi_preprocess', C.byref(d), None, None, None, None, None, 1, 1, None, None, _lib.cur_stream())


def test_nchw_to_nhwc_with_transform_input(ctx):
    from ifcb_classifier_amd import _lib
    x = torch.rand(3, 3, 17, 19)
    sc = (C.c_float * 3)(0.229 / 0.5, 0.224 / 0.5, 0.225 / 0.5)
    sh = (C.c_float * 3)((0.485 - 0.5) / 0.5, (0.456 - 0.5) / 0.5, (0.406 - 0.5) / 0.5)
    y = torch.full((3, 17, 19, 8), 7.0, dtype=torch.bfloat16, device='cuda')
    xd = x.cuda()
    ctx.call('ifcbk_nchw_to_nhwc', _lib.ptr(xd), 3, 3, 17, 19, 8, 0, sc, sh, _lib.ptr(y), _lib.cur_stream())
    torch.cuda.synchronize()
    ref = torch.stack([x[:, c] * sc[c] + sh[c] for c in range(3)], 1)
    # the kernel contracts v*scale+shift into one fma: at most one bf16 ulp from torch's mul-then-add
    assert (nchw(y)[:, :3] - _bf(ref)).abs().max().item() <= 2 ** -8 * ref.abs().max().item()
    assert ((nchw(y)[:, :3] != _bf(ref)).float().mean().item()) < 0.01
    assert nchw(y)[:, 3:].abs().max().item() == 0


@pytest.mark.parametrize('N,Cc,H,W,ld,relu,res', [(4, 32, 9, 9, 32, 1, 0), (3, 96, 7, 5, 160, 1, 0), (2, 64, 6, 6, 64, 1, 1),
                                                 (2, 128, 5, 5, 128, 0, 0), (33, 16, 40, 40, 16, 1, 0)])
def test_bn_fwd_bwd_vs_oracle(ctx, N, Cc, H, W, ld, relu, res):
    from ifcb_classifier_amd import _lib
    from oracle import ops as O
    g = torch.Generator().manual_seed(N * 100 + Cc)
    raw = _bf(torch.randn(N, Cc, H, W, generator=g) * 2 + 0.5)
    gamma = torch.rand(Cc, generator=g) + 0.5
    beta = torch.randn(Cc, generator=g) * 0.2
    resid = _bf(torch.randn(N, Cc, H, W, generator=g)) if res else None
    gy = _bf(torch.randn(N, Cc, H, W, generator=g))
    eps = 1e-3
    y_ref, mean, var = O.bn_act_fwd(raw, gamma, beta, eps, bool(relu), resid)
    d_raw, dg, db, dres = O.bn_act_bwd(raw, gamma, beta, eps, bool(relu), resid, gy)
    M = N * H * W
    # statistics exactly as the conv epilogue hands them over: per-128-row partial (sum, sumsq)
    rows = raw.permute(0, 2, 3, 1).reshape(M, Cc)
    mb = (M + 127) // 128
    part = torch.zeros(mb, 2, Cc)
    for i in range(mb):
        blk = rows[i * 128:(i + 1) * 128]
        part[i, 0], part[i, 1] = blk.sum(0), (blk * blk).sum(0)
    d = _lib.BnDesc(M, Cc, Cc, ld, relu, 0, eps, 0.1)
    st = _lib.cur_stream()
    dev = lambda t: t.cuda() if t is not None else None
    rm, rv = torch.zeros(Cc).cuda(), torch.ones(Cc).cuda()
    stats = [torch.zeros(Cc).cuda() for _ in range(4)]
    part_d, gamma_d, beta_d = part.cuda(), gamma.cuda(), beta.cuda()
    ctx.call('ifcbk_bn_finalize', C.byref(d), _lib.ptr(part_d), mb, _lib.ptr(gamma_d), _lib.ptr(beta_d),
             _lib.ptr(rm), _lib.ptr(rv), *[_lib.ptr(s) for s in stats], st)
    rawd = nhwc(raw)
    y = torch.zeros(N, H, W, ld, dtype=torch.bfloat16, device='cuda')
    resd = nhwc(resid) if res else None
    ctx.call('ifcbk_bn_apply', C.byref(d), _lib.ptr(rawd), _lib.ptr(stats[2]), _lib.ptr(stats[3]), _lib.ptr(resd), Cc,
             _lib.ptr(y), st)
    torch.cuda.synchronize()
    assert torch.allclose(stats[0].cpu(), mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(stats[1].cpu(), 1 / torch.sqrt(var + eps), rtol=1e-5)
    assert torch.allclose(rm.cpu(), 0.1 * mean, rtol=1e-5, atol=1e-7)
    assert torch.allclose(rv.cpu(), 0.9 + 0.1 * var * M / (M - 1), rtol=1e-5)
    yh = nchw(y)[:, :Cc]
    assert (yh - y_ref).abs().max().item() <= 2e-2 * y_ref.abs().max().item()
    assert ((yh - y_ref).abs() > 0).float().mean().item() < 0.02      # only rare 1-ulp bf16 flips
    # backward (teacher-forced with the oracle's y so the ReLU mask is identical)
    yd = torch.zeros(N, H, W, ld, dtype=torch.bfloat16, device='cuda')
    yd[..., :Cc] = nhwc(y_ref)
    gyd = torch.zeros(N, H, W, ld, dtype=torch.bfloat16, device='cuda')
    gyd[..., :Cc] = nhwc(gy)
    dx = torch.zeros(N, H, W, Cc, dtype=torch.bfloat16, device='cuda')
    dresd = torch.zeros(N, H, W, Cc, dtype=torch.bfloat16, device='cuda') if res else None
    dgam, dbet = torch.zeros(Cc).cuda(), torch.zeros(Cc).cuda()
    ctx.call('ifcbk_bn_bwd', C.byref(d), _lib.ptr(rawd), _lib.ptr(yd), _lib.ptr(gyd), ld, _lib.ptr(gamma_d),
             _lib.ptr(stats[0]), _lib.ptr(stats[1]), _lib.ptr(dx), Cc, _lib.ptr(dresd), Cc, 0, _lib.ptr(dgam),
             _lib.ptr(dbet), 0, _lib.ptr(stats[2]), _lib.ptr(stats[3]), st)
    torch.cuda.synchronize()
    assert torch.allclose(dgam.cpu(), dg, rtol=2e-4, atol=2e-4 * dg.abs().max().item())
    assert torch.allclose(dbet.cpu(), db, rtol=2e-4, atol=2e-4 * db.abs().max().item())
    assert (nchw(dx) - d_raw).abs().max().item() <= 1e-2 * d_raw.abs().max().item()
    if res:
        assert (nchw(dresd) - _bf(dres)).abs().max().item() <= 1e-2 * dres.abs().max().item()


@pytest.mark.parametrize('N,Cc,H,W,pad,dtype', [(3, 64, 21, 21, 0, 0), (2, 192, 15, 13, 0, 0), (2, 64, 16, 16, 1, 0),
                                               (2, 32, 9, 12, 1, 1), (2, 32, 11, 14, 0, 1), (1, 64, 147, 147, 0, 0)])
def test_bn_maxpool_fused_equals_unfused(ctx, N, Cc, H, W, pad, dtype):
    """ifcbk_bn_apply_maxpool == bn_apply -> maxpool_fwd bit for bit (values and arg-max), and ifcbk_bn_bwd_maxpool ==
    maxpool_bwd -> bn_bwd: bit for bit in fp32 storage; in bf16 storage the unfused path rounds the (never stored here)
    activation gradient -- a sum of up to four pooled gradients -- to bf16, the fused path keeps it in fp32; [TV] inception.py
    `F.max_pool2d(relu(bn(conv(x))), 3, 2)`, reference call site neuston_models.py:66-68."""
    from ifcb_classifier_amd import _lib
    g = torch.Generator().manual_seed(N * 1000 + Cc + H)
    tdt = torch.float32 if dtype else torch.bfloat16
    P, Q = (H + 2 * pad - 3) // 2 + 1, (W + 2 * pad - 3) // 2 + 1
    raw = (torch.randn(N, H, W, Cc, generator=g) * 2 + 0.3).to(tdt).cuda()
    scale = (torch.randn(Cc, generator=g) * 0.7).cuda()            # negative scales too: affine before the max, always
    shift = (torch.randn(Cc, generator=g) * 0.3).cuda()
    gamma = (torch.rand(Cc, generator=g) + 0.5).cuda()
    mean = torch.randn(Cc, generator=g).cuda() * 0.2
    invstd = (torch.rand(Cc, generator=g) + 0.5).cuda()
    dpool = torch.randn(N, P, Q, Cc, generator=g).to(tdt).cuda()
    st = _lib.cur_stream()
    M = N * H * W
    bd = _lib.BnDesc(M, Cc, Cc, Cc, 1, dtype, 1e-3, 0.1)
    pd = _lib.PoolDesc(N, H, W, Cc, Cc, 3, 3, 2, 2, pad, pad, P, Q, Cc, dtype)
    ctx.reserve(1 << 24)
    # ---- unfused
    y = torch.empty(N, H, W, Cc, dtype=tdt, device='cuda')
    yp = torch.empty(N, P, Q, Cc, dtype=tdt, device='cuda')
    arg = torch.empty(N, P, Q, Cc, dtype=torch.uint8, device='cuda')
    ctx.call('ifcbk_bn_apply', C.byref(bd), _lib.ptr(raw), _lib.ptr(scale), _lib.ptr(shift), None, 0, _lib.ptr(y), st)
    ctx.call('ifcbk_maxpool_fwd', C.byref(pd), _lib.ptr(y), _lib.ptr(yp), _lib.ptr(arg), st)
    dy = torch.empty_like(y)
    ctx.call('ifcbk_maxpool_bwd', C.byref(pd), _lib.ptr(dpool), _lib.ptr(arg), _lib.ptr(dy), 0, st)
    dx = torch.empty_like(y)
    dg, db = torch.zeros(Cc).cuda(), torch.zeros(Cc).cuda()
    ctx.call('ifcbk_bn_bwd', C.byref(bd), _lib.ptr(raw), _lib.ptr(y), _lib.ptr(dy), Cc, _lib.ptr(gamma), _lib.ptr(mean),
             _lib.ptr(invstd), _lib.ptr(dx), Cc, None, 0, 0, _lib.ptr(dg), _lib.ptr(db), 0, _lib.ptr(scale), _lib.ptr(shift), st)
    # ---- fused
    yp2 = torch.empty_like(yp)
    arg2 = torch.empty_like(arg)
    ctx.call('ifcbk_bn_apply_maxpool', C.byref(pd), _lib.ptr(raw), _lib.ptr(scale), _lib.ptr(shift), 1, _lib.ptr(yp2),
             _lib.ptr(arg2), st)
    dx2 = torch.empty_like(dx)
    dg2, db2 = torch.zeros(Cc).cuda(), torch.zeros(Cc).cuda()
    ctx.call('ifcbk_bn_bwd_maxpool', C.byref(pd), _lib.ptr(raw), _lib.ptr(dpool), _lib.ptr(arg2), _lib.ptr(gamma),
             _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(scale), _lib.ptr(shift), 1, _lib.ptr(dx2), Cc, _lib.ptr(dg2),
             _lib.ptr(db2), 0, st)
    torch.cuda.synchronize()
    assert torch.equal(yp, yp2)
    assert torch.equal(arg, arg2)
    if dtype:        # fp32 storage: identical terms, summed in a different order (2x2 pixel blocks) in the unpadded fast path
        for a, b in ((dg, dg2), (db, db2), (dx, dx2)):
            assert (a - b).abs().max().item() <= 1e-5 * a.abs().max().item()
    else:
        for a, b in ((dg, dg2), (db, db2), (dx.float(), dx2.float())):
            assert (a - b).abs().max().item() <= 1e-2 * a.abs().max().item()


def test_bn_eval_scale_shift(ctx):
    from ifcb_classifier_amd import _lib
    Cc = 40
    g = torch.Generator().manual_seed(1)
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    rm, rv = torch.randn(Cc, generator=g), torch.rand(Cc, generator=g) + 0.1
    d = _lib.BnDesc(10, Cc, Cc, Cc, 1, 0, 1e-5, 0.1)
    sc, sh = torch.zeros(Cc).cuda(), torch.zeros(Cc).cuda()
    gd, bd, rmd, rvd = gamma.cuda(), beta.cuda(), rm.cuda(), rv.cuda()
    ctx.call('ifcbk_bn_finalize', C.byref(d), None, 0, _lib.ptr(gd), _lib.ptr(bd), _lib.ptr(rmd),
             _lib.ptr(rvd), None, None, _lib.ptr(sc), _lib.ptr(sh), _lib.cur_stream())
    torch.cuda.synchronize()
    s = gamma / torch.sqrt(rv + 1e-5)
    assert torch.allclose(sc.cpu(), s, rtol=1e-6) and torch.allclose(sh.cpu(), beta - rm * s, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('kind,N,Cc,H,W,k,s,p', [('max', 2, 64, 15, 15, 3, 2, 0), ('max', 2, 64, 14, 14, 3, 2, 1),
                                                 ('avg', 2, 48, 9, 9, 3, 1, 1), ('avg', 3, 768, 17, 17, 5, 3, 0),
                                                 ('max', 1, 8, 5, 5, 3, 2, 0)])
def test_pool_fwd_bwd_vs_oracle(ctx, kind, N, Cc, H, W, k, s, p):
    from ifcb_classifier_amd import _lib
    from oracle import ops as O
    g = torch.Generator().manual_seed(H * 7 + Cc)
    x = _bf(torch.randn(N, Cc, H, W, generator=g))
    if kind == 'max':
        x = F.relu(x)                    # many exact ties (zeros), as after ReLU: first-max tie-breaking matters
    P = (H + 2 * p - k) // s + 1
    gy = _bf(torch.randn(N, Cc, P, P, generator=g))
    y_ref = O.pool_fwd(kind, x, k, s, p)
    dx_ref = O.pool_bwd(kind, x, k, s, p, gy)
    d = _lib.PoolDesc(N, H, W, Cc, Cc, k, k, s, s, p, p, P, P, Cc, 0)
    st = _lib.cur_stream()
    xd, gyd = nhwc(x), nhwc(gy)
    y = torch.zeros(N, P, P, Cc, dtype=torch.bfloat16, device='cuda')
    dx = torch.full((N, H, W, Cc), 1.0, dtype=torch.bfloat16, device='cuda')
    if kind == 'max':
        am = torch.zeros(N, P, P, Cc, dtype=torch.uint8, device='cuda')
        ctx.call('ifcbk_maxpool_fwd', C.byref(d), _lib.ptr(xd), _lib.ptr(y), _lib.ptr(am), st)
        ctx.call('ifcbk_maxpool_bwd', C.byref(d), _lib.ptr(gyd), _lib.ptr(am), _lib.ptr(dx), 0, st)
    else:
        ctx.call('ifcbk_avgpool_fwd', C.byref(d), _lib.ptr(xd), _lib.ptr(y), st)
        ctx.call('ifcbk_avgpool_bwd', C.byref(d), _lib.ptr(gyd), _lib.ptr(dx), 0, st)
    torch.cuda.synchronize()
    if kind == 'max':
        assert torch.equal(nchw(y), y_ref)
    else:
        assert (nchw(y) - y_ref).abs().max().item() <= 8e-3 * y_ref.abs().max().item()
    assert (nchw(dx) - _bf(dx_ref)).abs().max().item() <= 1e-2 * dx_ref.abs().max().item() + 1e-6
    # accumulate
    if kind == 'max':
        ctx.call('ifcbk_maxpool_bwd', C.byref(d), _lib.ptr(gyd), _lib.ptr(am), _lib.ptr(dx), 1, st)
    else:
        ctx.call('ifcbk_avgpool_bwd', C.byref(d), _lib.ptr(gyd), _lib.ptr(dx), 1, st)
    torch.cuda.synchronize()
    assert (nchw(dx) - 2 * dx_ref).abs().max().item() <= 3e-2 * dx_ref.abs().max().item() + 1e-6


@pytest.mark.parametrize('N,HW,Cc,NC,drop', [(6, 64, 2048, 100, True), (5, 1, 768, 100, False), (3, 49, 512, 2, False)])
def test_head_loss_vs_oracle(ctx, N, HW, Cc, NC, drop):
    from ifcb_classifier_amd import _lib
    from oracle import ops as O
    g = torch.Generator().manual_seed(Cc + NC)
    side = int(HW ** 0.5)
    x = _bf(torch.rand(N, Cc, side, side, generator=g))
    W = torch.randn(NC, Cc, generator=g) * 0.05
    b = torch.randn(NC, generator=g) * 0.1
    mask = (torch.rand(N, Cc, generator=g) > 0.5) if drop else None
    tgt = torch.randint(0, NC, (N,), generator=g)
    feat, logits = O.head_fwd(x, W, b, mask)
    loss_ref, dl_ref = O.xent(logits, tgt, 0.4)
    dx_ref, dW_ref, db_ref = O.head_bwd(x, W, b, mask, dl_ref)
    d = _lib.HeadDesc(N, HW, Cc, Cc, NC, 0, 2.0)
    st = _lib.cur_stream()
    xd = nhwc(x)
    md = mask.to(torch.uint8).cuda() if drop else None
    Wd, bd = W.cuda(), b.cuda()
    featd, lg = torch.zeros(N, Cc).cuda(), torch.zeros(N, NC).cuda()
    ctx.call('ifcbk_head_fwd', C.byref(d), _lib.ptr(xd), _lib.ptr(md), _lib.ptr(Wd), _lib.ptr(bd), _lib.ptr(featd),
             _lib.ptr(lg), st)
    loss, dl = torch.full((1,), 5.0).cuda(), torch.zeros(N, NC).cuda()
    tgt_d = tgt.cuda()
    ctx.call('ifcbk_softmax_xent', _lib.ptr(lg), _lib.ptr(tgt_d), N, NC, 0.4, _lib.ptr(loss), 1, _lib.ptr(dl), st)
    probs = torch.zeros(N, NC).cuda()
    ctx.call('ifcbk_softmax', _lib.ptr(lg), N, NC, _lib.ptr(probs), st)
    dW, dbb = torch.zeros(NC, Cc).cuda(), torch.zeros(NC).cuda()
    dx = torch.zeros(N, side, side, Cc, dtype=torch.bfloat16, device='cuda')
    ctx.call('ifcbk_head_bwd', C.byref(d), _lib.ptr(dl), _lib.ptr(featd), _lib.ptr(md), _lib.ptr(Wd), _lib.ptr(dW),
             _lib.ptr(dbb), _lib.ptr(dx), Cc, 0, st)
    torch.cuda.synchronize()
    assert torch.allclose(lg.cpu(), logits, rtol=1e-4, atol=1e-5)
    assert abs(loss.item() - (5.0 + loss_ref.item())) < 1e-5                    # accumulate flag honoured
    assert torch.allclose(dl.cpu(), dl_ref, rtol=1e-4, atol=1e-7)
    assert torch.allclose(probs.cpu(), torch.softmax(logits, 1), rtol=1e-5, atol=1e-7)
    assert torch.allclose(dW.cpu(), dW_ref, rtol=1e-4, atol=1e-7)
    assert torch.allclose(dbb.cpu(), db_ref, rtol=1e-4, atol=1e-7)
    assert (nchw(dx) - dx_ref).abs().max().item() <= 1e-2 * dx_ref.abs().max().item()


def test_dropout_mask_rate_and_determinism(ctx):
    from ifcb_classifier_amd import _lib
    n = 1 << 20
    m1, m2 = torch.zeros(n, dtype=torch.uint8).cuda(), torch.zeros(n, dtype=torch.uint8).cuda()
    ctx.call('ifcbk_dropout_mask', _lib.ptr(m1), n, 0.5, 42, 0, _lib.cur_stream())
    ctx.call('ifcbk_dropout_mask', _lib.ptr(m2), n, 0.5, 42, 0, _lib.cur_stream())
    torch.cuda.synchronize()
    assert torch.equal(m1, m2)
    assert abs(m1.float().mean().item() - 0.5) < 5e-3
    ctx.call('ifcbk_dropout_mask', _lib.ptr(m2), n, 0.5, 42, n, _lib.cur_stream())
    torch.cuda.synchronize()
    assert not torch.equal(m1, m2)


def test_adam_matches_torch_adam_over_steps(ctx):
    from ifcb_classifier_amd import _lib
    from oracle import ops as O
    n = 100003
    g = torch.Generator().manual_seed(0)
    p = torch.randn(n, generator=g)
    pd, m, v = p.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    pr, mr, vr = p.clone(), torch.zeros(n), torch.zeros(n)
    for step in range(1, 4):
        gr = torch.randn(n, generator=g) * 10 ** (step - 2)
        gr_d = gr.cuda()
        ctx.call('ifcbk_adam_flat', _lib.ptr(pd), _lib.ptr(gr_d), _lib.ptr(m), _lib.ptr(v), n, 1e-3, 0.9, 0.999, 1e-8,
                 0.0, step, 1.0, _lib.cur_stream())
        torch.cuda.synchronize()
        pr, mr, vr = O.adam_step(pr, gr, mr, vr, step)
    torch.cuda.synchronize()
    assert (pd.cpu() - pr).abs().max().item() < 2e-6
    assert torch.allclose(m.cpu(), mr, rtol=1e-5, atol=1e-8)
    assert torch.allclose(v.cpu(), vr, rtol=2e-5, atol=1e-8)        # fma contraction vs torch's mul_/addcmul_


@pytest.mark.parametrize('N,Cc,H,W,ld,dtype', [(3, 64, 35, 35, 288, 0), (2, 192, 17, 17, 192, 0), (5, 32, 8, 8, 96, 1), (1, 8, 3, 3, 8, 0)])
def test_bn_stats_of_a_stored_tensor(ctx, N, Cc, H, W, ld, dtype):
    """ifcbk_bn_stats (batch statistics of the pooled conv output of a commuted pool branch): per-1024-row partial sums of
    the STORED values; their finalize gives the mean / biased variance torch computes on the same tensor."""
    from ifcb_classifier_amd import _lib
    tdt = torch.bfloat16 if dtype == 0 else torch.float32
    g = torch.Generator().manual_seed(N + Cc)
    buf = (torch.randn(N, H, W, ld, generator=g) * 1.5 + 0.3).to(tdt).cuda()      # the tensor is a channel slice of a wider one
    x = buf[..., 8:8 + Cc] if ld >= Cc + 8 else buf[..., :Cc]
    M = N * H * W
    rows = ctx.lib.ifcbk_bn_stats_rows(M)
    assert rows == (M + 1023) // 1024
    part = torch.full((rows, 2, Cc), float('nan'), device='cuda')
    d = _lib.BnDesc(M, Cc, ld, ld, 1, dtype, 1e-3, 0.1)
    ctx.call('ifcbk_bn_stats', C.byref(d), _lib.ptr(x), _lib.ptr(part), _lib.cur_stream())
    torch.cuda.synchronize()
    xs = x.float().reshape(M, Cc).double().cpu()
    s = part.double().cpu().sum(0)
    assert torch.isfinite(part).all()
    assert torch.allclose(s[0], xs.sum(0), rtol=2e-6, atol=1e-3)
    assert torch.allclose(s[1], (xs * xs).sum(0), rtol=2e-6, atol=1e-3)
    for r in range(rows):       # every partial row covers exactly its 1024 rows
        blk = xs[r * 1024:(r + 1) * 1024]
        assert torch.allclose(part[r, 0].double().cpu(), blk.sum(0), rtol=1e-5, atol=1e-3)


def test_sgd_matches_torch_sgd_over_steps(ctx):
    """additive --optimizer SGD (north_star "SGD/Adam step"; upstream only has Adam): torch.optim.SGD's update with and
    without momentum, and the fused engine step that reaches it through IFCBK_OP_SGD"""
    from ifcb_classifier_amd import _lib
    n = 70001
    for mu in (0.0, 0.9):
        g = torch.Generator().manual_seed(1)
        p = torch.randn(n, generator=g)
        pd, mom = p.clone().cuda(), torch.zeros(n).cuda()
        pr = p.clone().requires_grad_(True)
        opt = torch.optim.SGD([pr], lr=0.05, momentum=mu)
        for step in range(4):
            gr = torch.randn(n, generator=g)
            ctx.call('ifcbk_sgd_flat', _lib.ptr(pd), _lib.ptr(gr.cuda()), _lib.ptr(mom) if mu else None, n, 0.05, mu, 0.0, 1.0,
                     _lib.cur_stream())
            pr.grad = gr.clone()
            opt.step()
        torch.cuda.synchronize()
        assert (pd.cpu() - pr.detach()).abs().max().item() < 2e-6, mu
    # the engine's fused step with optimizer='sgd' == its own gradients applied by torch.optim.SGD
    from ifcb_classifier_amd import graph
    from ifcb_classifier_amd.engine import Engine
    eng = Engine(graph.build('resnet18', 3), 0, max_batch=4, optimizer='sgd', lr=0.01, momentum=0.9)
    eng.init_weights(seed=2)
    ref = eng.P.clone().requires_grad_(True)
    opt = torch.optim.SGD([ref], lr=0.01, momentum=0.9)
    gg = torch.Generator().manual_seed(3)
    for step in range(2):
        x = torch.rand(4, 3, 224, 224, generator=gg).cuda()
        eng.load_input_nchw(x)
        eng.target[:4].copy_(torch.tensor([0, 1, 2, 1]))
        with torch.no_grad():
            ref.copy_(eng.P)                      # same starting point each step: compares ONE update, not a trajectory
        mom_before = eng.M.clone()
        eng.train_step(4)
        torch.cuda.synchronize()
        if step:
            opt.state[ref]['momentum_buffer'] = mom_before.clone()
        ref.grad = eng.G.clone()
        opt.step()
        assert (eng.P - ref.detach()).abs().max().item() < 1e-6
    names = set()
    import ctypes as CC
    buf = CC.create_string_buffer(64)
    pl = eng.plan(4)
    for k in range(pl.step.n):
        eng.ctx.lib.ifcbk_op_kernel(CC.byref(pl.step.arr[k]), buf, 64)
        names.add(buf.value)
    assert b'sgd_kernel' in names and b'adam_kernel' not in names
