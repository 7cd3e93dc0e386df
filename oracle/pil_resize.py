"""numpy restatement of Pillow's 8-bit bilinear ``Image.resize`` -- the arithmetic behind
``transforms.Resize([S,S])`` at ``/root/reference/neuston_data.py:345,460`` (PIL is a third-party
dependency of the reference, pinned ``pillow==8.4.0`` in ``requirements/pkgs.hpc.txt:77``; its C source
``libImaging/Resample.c`` is not under /root/reference).  Published algorithm restated here:

  precompute_coeffs : per output index, support = max(scale,1), window [center-support, center+support],
                      triangle weights evaluated in double, normalised by their sum
  normalize_coeffs_8bpc : weights -> int, round half away from zero, PRECISION_BITS = 32-8-2 = 22
  horizontal pass then vertical pass, each: acc = 1<<21 + sum(pixel*k); out = clip8(acc >> 22), with an
  8-bit intermediate image between the passes.

Pinned against Pillow itself (installed, 12.2.0; bit-identical to 8.4.0 on this chain, SURVEY.md Appendix B
probe 6) by ``tests/test_oracle_cpu.py`` and the committed ``tests/golden/pil_resize_*.npz`` vectors.
TEST INFRASTRUCTURE ONLY.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _coeffs(in_size, out_size):
    scale = float(np.float32(in_size)) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int64)
    kk = np.zeros((out_size, ksize), np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(ksize, np.float64)
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            if a < 0.0:
                a = -a
            w[x] = 1.0 - a if a < 1.0 else 0.0
            ww += w[x]
        for x in range(xmax):
            if ww != 0.0:
                w[x] /= ww
        for x in range(ksize):
            v = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(a):
    return np.clip(a >> PRECISION_BITS, 0, 255)


def resize_bilinear_u8(img, out_h, out_w):
    """img: uint8 [H,W] or [H,W,C] -> uint8 [out_h,out_w(,C)], bit-exact to PIL ``resize(..., BILINEAR)``."""
    img = np.asarray(img, np.uint8)
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    H, W, Cn = img.shape
    src = img.astype(np.int64)
    # horizontal
    if out_w != W:
        bounds, kk = _coeffs(W, out_w)
        tmp = np.zeros((H, out_w, Cn), np.int64)
        for xx in range(out_w):
            x0, n = bounds[xx]
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(src[:, x0:x0 + n, :], kk[xx, :n], axes=([1], [0]))
            tmp[:, xx, :] = _clip8(acc)
        src = tmp
    # vertical
    if out_h != H:
        bounds, kk = _coeffs(H, out_h)
        out = np.zeros((out_h, src.shape[1], Cn), np.int64)
        for yy in range(out_h):
            y0, n = bounds[yy]
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[yy, :n], src[y0:y0 + n, :, :], axes=([0], [0]))
            out[yy] = _clip8(acc)
        src = out
    res = src.astype(np.uint8)
    return res[:, :, 0] if squeeze else res


def roi_to_tensor(roi, S, mean=None, std=None, flip_v=False, flip_h=False):
    """``IfcbBinDataset.__getitem__`` / train transforms restated (neuston_data.py:342-371,456-464):
    [flips] -> L->RGB -> Resize([S,S]) -> ToTensor -> [Normalize].  Returns float32 [3,S,S]."""
    a = np.asarray(roi, np.uint8)
    if flip_v:
        a = a[::-1]
    if flip_h:
        a = a[:, ::-1]
    if a.ndim == 2:
        a = np.repeat(a[:, :, None], 3, 2)
    r = resize_bilinear_u8(np.ascontiguousarray(a), S, S)
    t = r.astype(np.float32).transpose(2, 0, 1) / np.float32(255)
    if mean is not None:
        m = np.asarray(mean, np.float32)[:, None, None]
        s = np.asarray(std, np.float32)[:, None, None]
        t = (t - m) / s
    return t
