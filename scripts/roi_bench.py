"""dev tool: the ROI preprocess kernel alone at batch 256 (synthetic ROIs as in bench.py), microseconds per call.
IFCBK_LIB=<other libifcbk.so> times another build on the same box."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ifcb_classifier_amd import _lib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_rois

ctx = _lib.Context(0)
for S in (299, 224):
    rois, _ = synth_rois(256, 1234, torch.device('cuda'))
    d = _lib.RoiDesc()
    d.n_img, d.S, d.in_channels, d.out_channels, d.dtype, d.flip_bits_valid = 256, S, 1, 8, 0, 0
    for k in range(3):
        d.mean[k], d.std[k], d.tin_scale[k], d.tin_shift[k] = 0.0, 1.0, 1.0, 0.0
    ctx.reserve(ctx.lib.ifcbk_roi_preprocess_workspace(C.byref(d), rois['max_h'], rois['max_w']))
    out = torch.empty(256, S, S, 8, dtype=torch.bfloat16, device='cuda')
    fn = lambda: ctx.call('ifcbk_roi_preprocess', C.byref(d), _lib.ptr(rois['pixels']), _lib.ptr(rois['offs']), _lib.ptr(rois['hs']),
                          _lib.ptr(rois['ws']), None, rois['max_h'], rois['max_w'], _lib.ptr(out), None, _lib.cur_stream())
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print('S=%d  %.1f us per batch of 256 (coeffs + resize), checksum %d' % (S, best * 1e3, int(out.float().sum().item())), flush=True)
