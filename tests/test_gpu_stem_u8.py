"""The stem conv on the resized u8 ROI plane (csrc/conv_stem_u8.hip, ifcbk_stem_u8_{fwd,wgrad}) against torch's fp64
convolution of the three affine copies of that plane -- what PIL convert('RGB') -> Resize -> ToTensor -> Normalize followed by
[TV] Inception3.Conv2d_1a_3x3 computes (neuston_data.py:342-371, 456-464; neuston_models.py:66-68) -- and the engine's u8 input
path against its dense [N,S,S,8] path on the same ROIs."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def _ab(mean, std, tsc=(1, 1, 1), tsh=(0, 0, 0)):
    return [tsc[c] / (255.0 * std[c]) for c in range(3)] + [tsh[c] - tsc[c] * mean[c] / std[c] for c in range(3)]


@pytest.mark.parametrize('N,H,W,dtype', [(3, 299, 299, 'bf16'), (2, 31, 37, 'bf16'), (5, 64, 64, 'fp32'), (1, 33, 9, 'fp32'),
                                         (40, 75, 75, 'bf16'), (2, 69, 67, 'bf16'), (3, 90, 131, 'bf16')])
def test_stem_u8_kernels_vs_fp64_conv_of_the_three_affine_planes(N, H, W, dtype):
    from ifcb_classifier_amd import _lib
    from ifcb_classifier_amd._lib import ConvDesc
    ctx = _lib.Context(0)
    st = _lib.cur_stream()
    tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    cdt = _lib.BF16 if dtype == 'bf16' else _lib.F32
    P, Q, K, LD = (H - 3) // 2 + 1, (W - 3) // 2 + 1, 32, 40
    d = ConvDesc(N, H, W, 8, 8, K, 3, 3, 2, 2, 0, 0, P, Q, LD, 3, cdt)
    rows = ctx.lib.ifcbk_stem_u8_rows(C.byref(d))
    assert rows == ((N * P + 7) // 8 if dtype == 'bf16' and Q >= 32 else (N * P * Q + 2047) // 2048)      # MFMA kernels: a block per 8 output rows
    ctx.reserve(max(1 << 20, ctx.lib.ifcbk_stem_u8_wgrad_workspace(C.byref(d))))
    gen = torch.Generator(device='cuda').manual_seed(3)
    g = torch.randint(0, 256, (N, H, W), device='cuda', generator=gen, dtype=torch.uint8)
    w = torch.randn(K, 3, 3, 3, device='cuda', generator=gen) * 0.2                    # master layout [K][R][S][C]
    ab_host = _ab((0.485, 0.456, 0.406), (0.229, 0.224, 0.225), (0.458, 0.448, 0.45), (-0.03, -0.088, -0.188))
    ab = torch.tensor(ab_host, device='cuda', dtype=torch.float32)
    x = torch.stack([ab[c].double() * g.double() + ab[3 + c].double() for c in range(3)], 1)          # [N,3,H,W] fp64
    x.requires_grad_(False)
    w64 = w.double().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    ref = F.conv2d(x, w64, None, 2)                                                   # [N,32,P,Q]
    ref_y = ref.permute(0, 2, 3, 1).detach()

    # ---- training forward: raw output + BatchNorm partial sums of the rounded outputs; the pad columns stay untouched
    yb = torch.full((N, P, Q, LD), float('nan'), device='cuda', dtype=tdt)
    part = torch.full((rows, 2, K), float('nan'), device='cuda')
    ctx.call('ifcbk_stem_u8_fwd', C.byref(d), _lib.ptr(g), _lib.ptr(w), _lib.ptr(ab), _lib.ptr(yb), _lib.ptr(part), None, None, 0, st)
    torch.cuda.synchronize()
    y = yb[..., :K]
    assert torch.isnan(yb[..., K:].float()).all()
    tol = 2.0 ** -8 if dtype == 'bf16' else 2e-6
    assert ((y.double() - ref_y).abs() <= tol * ref_y.abs() + 1e-4).all(), (y.double() - ref_y).abs().max().item()
    assert _rel(part[:, 0].double().sum(0), y.double().sum((0, 1, 2))) < 2e-6
    assert _rel(part[:, 1].double().sum(0), (y.double() ** 2).sum((0, 1, 2))) < 2e-6
    yb2 = torch.full_like(yb, float('nan'))
    part2 = torch.full_like(part, float('nan'))
    ctx.call('ifcbk_stem_u8_fwd', C.byref(d), _lib.ptr(g), _lib.ptr(w), _lib.ptr(ab), _lib.ptr(yb2), _lib.ptr(part2), None, None, 0, st)
    torch.cuda.synchronize()
    assert torch.equal(yb2[..., :K], y) and torch.equal(part2, part)                   # bitwise repeatable

    # ---- eval forward: folded BatchNorm affine + ReLU in the same pass
    scale = torch.rand(K, device='cuda', generator=gen) + 0.5
    shift = torch.randn(K, device='cuda', generator=gen) * 0.3
    for relu in (1, 0):
        ya = torch.full((N, P, Q, LD), float('nan'), device='cuda', dtype=tdt)
        ctx.call('ifcbk_stem_u8_fwd', C.byref(d), _lib.ptr(g), _lib.ptr(w), _lib.ptr(ab), _lib.ptr(ya), None, _lib.ptr(scale),
                 _lib.ptr(shift), relu, st)
        torch.cuda.synchronize()
        want = ref_y * scale.double() + shift.double()
        if relu:
            want = want.clamp_min(0)
        assert ((ya[..., :K].double() - want).abs() <= tol * want.abs() + 2e-4).all()

    # ---- weight gradient (master layout [K][R][S][3]), plain and accumulating
    d2 = ConvDesc(N, H, W, 8, 8, K, 3, 3, 2, 2, 0, 0, P, Q, K, 3, cdt)
    dy = torch.randn(N, P, Q, K, device='cuda', generator=gen).to(tdt)
    ref.backward(dy.double().permute(0, 3, 1, 2))
    ref_dw = w64.grad.permute(0, 2, 3, 1).contiguous()                                # [K][R][S][C]
    dw = torch.full((K, 3, 3, 3), float('nan'), device='cuda')
    ctx.call('ifcbk_stem_u8_wgrad', C.byref(d2), _lib.ptr(g), _lib.ptr(dy), _lib.ptr(ab), _lib.ptr(dw), 0, st)
    torch.cuda.synchronize()
    assert _rel(dw, ref_dw) < 2e-5, _rel(dw, ref_dw)
    base = torch.randn(K, 3, 3, 3, device='cuda', generator=gen)
    dw2 = base.clone()
    ctx.call('ifcbk_stem_u8_wgrad', C.byref(d2), _lib.ptr(g), _lib.ptr(dy), _lib.ptr(ab), _lib.ptr(dw2), 1, st)
    torch.cuda.synchronize()
    assert torch.equal(dw2, base + dw)


def test_stem_u8_rows_beyond_the_2_gib_offset_of_the_output():
    """1,520 images of 149 x 149 x 32 bf16 are 2.16 GB: the rows of images >= 1,511 start beyond byte 2^31 of the output.  The
    MFMA kernel builds a row's address from two readfirstlane halves; round 4 found the low half sign-extended there (the rows
    were written 4 GiB in front of the tensor: a RUN batch of 1,536 faulted, one of 2,048 returned wrong probabilities).  The
    last images of the big launch must equal the same images run on their own, and the tensor in front of them stays intact."""
    from ifcb_classifier_amd import _lib
    from ifcb_classifier_amd._lib import ConvDesc
    ctx = _lib.Context(0)
    st = _lib.cur_stream()
    N, H, W, K, n_tail = 1520, 299, 299, 32, 24
    P = Q = 149
    gen = torch.Generator(device='cuda').manual_seed(5)
    g = torch.randint(0, 256, (N, H, W), device='cuda', generator=gen, dtype=torch.uint8)
    w = torch.randn(K, 3, 3, 3, device='cuda', generator=gen) * 0.2
    ab = torch.tensor(_ab((0.485, 0.456, 0.406), (0.229, 0.224, 0.225)), device='cuda', dtype=torch.float32)
    scale = torch.rand(K, device='cuda', generator=gen) + 0.5
    shift = torch.randn(K, device='cuda', generator=gen) * 0.3
    assert (N - n_tail) * P * Q * K * 2 < (1 << 31) < N * P * Q * K * 2
    y = torch.zeros(N, P, Q, K, device='cuda', dtype=torch.bfloat16)
    d = ConvDesc(N, H, W, 8, 8, K, 3, 3, 2, 2, 0, 0, P, Q, K, 3, _lib.BF16)
    ctx.call('ifcbk_stem_u8_fwd', C.byref(d), _lib.ptr(g), _lib.ptr(w), _lib.ptr(ab), _lib.ptr(y), None, _lib.ptr(scale), _lib.ptr(shift), 1, st)
    torch.cuda.synchronize()
    i0 = N - n_tail
    y2 = torch.zeros(n_tail, P, Q, K, device='cuda', dtype=torch.bfloat16)
    d2 = ConvDesc(n_tail, H, W, 8, 8, K, 3, 3, 2, 2, 0, 0, P, Q, K, 3, _lib.BF16)
    ctx.call('ifcbk_stem_u8_fwd', C.byref(d2), _lib.ptr(g[i0:]), _lib.ptr(w), _lib.ptr(ab), _lib.ptr(y2), None, _lib.ptr(scale), _lib.ptr(shift), 1, st)
    torch.cuda.synchronize()
    assert y2.float().abs().sum().item() > 0
    assert torch.equal(y[i0:], y2)
    y3 = torch.zeros(8, P, Q, K, device='cuda', dtype=torch.bfloat16)
    d3 = ConvDesc(8, H, W, 8, 8, K, 3, 3, 2, 2, 0, 0, P, Q, K, 3, _lib.BF16)
    ctx.call('ifcbk_stem_u8_fwd', C.byref(d3), _lib.ptr(g), _lib.ptr(w), _lib.ptr(ab), _lib.ptr(y3), None, _lib.ptr(scale), _lib.ptr(shift), 1, st)
    torch.cuda.synchronize()
    assert torch.equal(y[:8], y3)


def test_stem_u8_refuses_what_it_does_not_serve():
    from ifcb_classifier_amd import _lib
    from ifcb_classifier_amd._lib import ConvDesc
    ctx = _lib.Context(0)
    for d in (ConvDesc(2, 32, 32, 8, 8, 64, 3, 3, 2, 2, 0, 0, 15, 15, 64, 3, _lib.BF16),          # 64 output channels
              ConvDesc(2, 32, 32, 8, 8, 32, 3, 3, 1, 1, 0, 0, 30, 30, 32, 3, _lib.BF16),          # stride 1
              ConvDesc(2, 32, 32, 8, 8, 32, 3, 3, 2, 2, 1, 1, 16, 16, 32, 3, _lib.BF16),          # padded
              ConvDesc(2, 32, 32, 8, 8, 32, 7, 7, 2, 2, 0, 0, 13, 13, 32, 3, _lib.BF16)):         # 7x7
        assert ctx.lib.ifcbk_stem_u8_rows(C.byref(d)) == 0
        t = torch.zeros(16, device='cuda')
        rc = ctx.lib.ifcbk_stem_u8_fwd(ctx.h, C.byref(d), _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), None, None, None, 0,
                                       _lib.cur_stream())
        assert rc == -4                                                                   # IFCBK_EUNSUPPORTED


def _rois(B, rng, lo=20, hi=150):
    hs = rng.integers(lo, hi, B).astype(np.int32)
    ws = rng.integers(lo, hi, B).astype(np.int32)
    sizes = hs.astype(np.int64) * ws
    offs = np.zeros(B, np.int64)
    offs[1:] = np.cumsum(sizes)[:-1]
    blob = rng.integers(0, 256, int(sizes.sum()), dtype=np.uint8)
    return dict(pixels=torch.from_numpy(blob).cuda(), offs=torch.from_numpy(offs).cuda(), hs=torch.from_numpy(hs).cuda(),
                ws=torch.from_numpy(ws).cuda(), max_h=int(hs.max()), max_w=int(ws.max()))


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_engine_u8_input_path_equals_the_dense_input_path(dtype, monkeypatch):
    """same ROIs, same weights: ROIs -> u8 plane -> stem_u8 ops (default) against ROIs -> [N,S,S,8] tensor -> GEMM kernels
    (IFCBK_STEM_U8=0).  fp32 parity mode: only the association of the input affine differs (1e-6); bf16: the dense path rounds the
    normalised input and the filter to bf16 first, the u8 path does not (bounded by that rounding)."""
    from ifcb_classifier_amd import _lib, graph
    from ifcb_classifier_amd.engine import Engine
    B, NC = 4, 5
    rng = np.random.default_rng(11)
    kw = _rois(B, rng)
    kw.update(mean=(0.5, 0.4, 0.3), std=(0.2, 0.25, 0.3), flips=torch.tensor([0, 1, 2, 3], dtype=torch.uint8).cuda())
    y = torch.from_numpy(rng.integers(0, NC, B)).cuda()
    out = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('IFCBK_STEM_U8', mode)
        e = Engine(graph.build('inception_v3', NC), 0, max_batch=B, dtype=dtype)
        e.init_weights(seed=2)
        e.external_mask = torch.ones(B, 2048, dtype=torch.uint8, device='cuda')
        assert (e.stem_u8 is not None) == (mode == '1')
        e.load_rois(**kw)
        assert e.in_kind[e.in_slot] == ('u8' if mode == '1' else 'nhwc')
        pl = e.forward_eval(B)
        e.run(pl.softmax)
        torch.cuda.synchronize()
        kinds = [pl.fwd_eval.arr[k].kind for k in range(pl.fwd_eval.n)]
        assert (_lib.OP_STEM_U8_FWD in kinds) == (mode == '1')
        probs = e.probs[:B].clone()
        e.target[:B].copy_(y)
        first = e.convs[0]
        assert first.x.buf.is_input
        e.forward_train(B)
        e.backward(B)
        torch.cuda.synchronize()
        kinds = [pl.step.arr[k].kind for k in range(pl.step.n)]
        assert (_lib.OP_STEM_U8_FWD in kinds and _lib.OP_STEM_U8_WGRAD in kinds) == (mode == '1')
        raw = e.act[first.raw.id][:B].float().clone()
        o, nel = e.poff[first.conv_key + '.weight'][0], first.K * 27
        gw = e.G[o:o + nel].clone()
        out[mode] = (probs, e.loss.clone(), raw, gw)
        del e
    tol_p, tol_l = (2e-4, 2e-4) if dtype == 'fp32' else (5e-2, 5e-2)
    assert (out['1'][0] - out['0'][0]).abs().max().item() < tol_p
    assert abs(out['1'][1].item() - out['0'][1].item()) < tol_l * max(1.0, abs(out['0'][1].item()))
    assert _rel(out['1'][2], out['0'][2]) < (1e-5 if dtype == 'fp32' else 1e-2)
    if dtype == 'fp32':
        assert _rel(out['1'][3], out['0'][3]) < 5e-3


def test_u8_and_dense_inputs_alternate_on_one_engine():
    """load_rois (u8 plane) and load_input_nchw (dense tensor) on the same engine pick their own programs; going back and forth
    reproduces each path's first result bit for bit"""
    from ifcb_classifier_amd import graph
    from ifcb_classifier_amd.engine import Engine
    B = 3
    rng = np.random.default_rng(5)
    kw = _rois(B, rng)
    e = Engine(graph.build('inception_v3', 4), 0, max_batch=B)
    e.init_weights(seed=1)
    x = torch.rand(B, 3, 299, 299, device='cuda')

    def ev(load):
        load()
        pl = e.forward_eval(B)
        e.run(pl.softmax)
        torch.cuda.synchronize()
        return [h for h in e.heads if not h.aux][0].logits[:B].clone()       # (the softmax saturates under the initial running statistics)
    a1 = ev(lambda: e.load_rois(**kw))
    b1 = ev(lambda: e.load_input_nchw(x))
    a2 = ev(lambda: e.load_rois(**kw))
    b2 = ev(lambda: e.load_input_nchw(x))
    assert torch.equal(a1, a2) and torch.equal(b1, b2) and not torch.equal(a1, b1)
    assert {k[2] for k in e._plans} == {'u8', 'nhwc'}
