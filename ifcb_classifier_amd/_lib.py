"""ctypes binding of ``libifcbk.so`` (the C-ABI declared in ``include/ifcbk.h``).

There is deliberately no fallback: if the shared library is missing or a call fails, a ``RuntimeError``
is raised.  Nothing here imports the CPU oracle.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libifcbk.so')
if os.environ.get('IFCBK_LIB'):
    # another build of the library for a same-box A/B of kernel variants: an explicit ABSOLUTE path to an existing file, nothing
    # is searched for; the product never sets it
    _alt = os.environ['IFCBK_LIB']
    if not (os.path.isabs(_alt) and os.path.isfile(_alt)):
        raise RuntimeError('IFCBK_LIB must be the absolute path of an existing libifcbk build, got %r' % _alt)
    import sys as _sys
    print('[ifcbk] IFCBK_LIB: loading %s instead of the in-tree library' % _alt, file=_sys.stderr)
    LIB_PATH = _alt

BF16, F32 = 0, 1
OK, EINVAL, EHIP, ENOMEM, EUNSUPPORTED = 0, -1, -2, -3, -4        # include/ifcbk.h

(OP_CONV_FWD, OP_CONV_DGRAD, OP_CONV_WGRAD, OP_WEIGHT_PACK, OP_BN_FINALIZE, OP_BN_APPLY, OP_BN_BWD,
 OP_MAXPOOL_FWD, OP_MAXPOOL_BWD, OP_AVGPOOL_FWD, OP_AVGPOOL_BWD, OP_HEAD_FWD, OP_HEAD_BWD,
 OP_SOFTMAX_XENT, OP_SOFTMAX, OP_ADAM, OP_MEMSET, OP_COPY2D, OP_DROPOUT_MASK, OP_CONV_FWD_AFFINE,
 OP_WEIGHT_PACK_MULTI, OP_CONV_WGRAD_SEG, OP_BN_APPLY_MAXPOOL, OP_BN_BWD_MAXPOOL, OP_CONV_DGRAD_BNSTAT,
 OP_BN_BWD_PARTIALS, OP_BN_STATS, OP_AVGPOOL_AFFINE, OP_CONV_FWD_AFFINE_SEG, OP_SGD, OP_CONV_DGRAD_BNSTAT_TAB,
 OP_BIAS_RELU_BWD, OP_DROPOUT, OP_FLATTEN_CHW, OP_STEM_U8_FWD, OP_STEM_U8_WGRAD, OP_CONV_FWD_AFFINE_MAXPOOL,
 OP_STEP_COUNTERS, OP_CONV_WGRAD_GROUP) = range(1, 40)

OP_NAMES = {1: 'conv_fwd', 2: 'conv_dgrad', 3: 'conv_wgrad', 4: 'weight_pack', 5: 'bn_finalize', 6: 'bn_apply',
            7: 'bn_bwd', 8: 'maxpool_fwd', 9: 'maxpool_bwd', 10: 'avgpool_fwd', 11: 'avgpool_bwd', 12: 'head_fwd',
            13: 'head_bwd', 14: 'softmax_xent', 15: 'softmax', 16: 'adam', 17: 'memset', 18: 'copy2d',
            19: 'dropout_mask', 20: 'conv_fwd_affine', 21: 'weight_pack_multi', 22: 'conv_wgrad', 23: 'bn_apply_maxpool',
            24: 'bn_bwd_maxpool', 25: 'conv_dgrad', 26: 'bn_bwd', 27: 'bn_stats', 28: 'avgpool_fwd', 29: 'conv_fwd_affine', 30: 'sgd', 31: 'conv_dgrad',
            32: 'bias_relu_bwd', 33: 'dropout', 34: 'flatten_chw', 35: 'conv_fwd', 36: 'conv_wgrad', 37: 'conv_fwd_affine', 38: 'step_counters', 39: 'conv_wgrad'}


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ('N', 'H', 'W', 'C', 'ldx', 'K', 'R', 'S', 'stride_h', 'stride_w', 'pad_h', 'pad_w', 'P', 'Q',
                 'ldy', 'Cw', 'dtype')]


class BnDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('M', 'C', 'ldx', 'ldy', 'relu', 'dtype')] + \
               [('eps', C.c_float), ('momentum', C.c_float)]


class PoolDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ('N', 'H', 'W', 'C', 'ldx', 'R', 'S', 'stride_h', 'stride_w', 'pad_h', 'pad_w', 'P', 'Q', 'ldy',
                 'dtype')]


class HeadDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('N', 'HW', 'C', 'ldx', 'NC', 'dtype')] + [('keep_scale', C.c_float)]


class RoiDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('n_img', 'S', 'in_channels', 'out_channels', 'flip_bits_valid', 'dtype')] + \
               [('mean', C.c_float * 3), ('std', C.c_float * 3), ('tin_scale', C.c_float * 3),
                ('tin_shift', C.c_float * 3)]


class BsChunk(C.Structure):
    _fields_ = [('raw', C.c_void_p), ('stat', C.c_void_p), ('raw_ld', C.c_int32), ('stat_ld', C.c_int32)]


class WgradItem(C.Structure):
    _fields_ = [('d', ConvDesc), ('x', C.c_void_p), ('dy', C.c_void_p), ('dw', C.c_void_p)]


class PackItem(C.Structure):
    _fields_ = [('w_master', C.c_void_p), ('w', C.c_void_p), ('wT', C.c_void_p), ('K', C.c_int32), ('RS', C.c_int32),
                ('C', C.c_int32), ('Cw', C.c_int32), ('first_block', C.c_int64), ('wT_ld', C.c_int32), ('pad_', C.c_int32)]


class _OpU(C.Union):
    _fields_ = [('conv', ConvDesc), ('bn', BnDesc), ('pool', PoolDesc), ('head', HeadDesc)]


class Op(C.Structure):
    _fields_ = [('kind', C.c_int32), ('flags', C.c_int32), ('p', C.c_void_p * 12), ('i', C.c_int64 * 4),
                ('f', C.c_float * 8), ('u', _OpU)]


_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_PROTOS = {
    'ifcbk_version': (C.c_char_p, []),
    'ifcbk_ctx_create': (_i, [_i, C.POINTER(_vp)]),
    'ifcbk_ctx_destroy': (_i, [_vp]),
    'ifcbk_ctx_reserve': (_i, [_vp, _sz]),
    'ifcbk_ctx_workspace_bytes': (_sz, [_vp]),
    'ifcbk_ctx_set_lanes': (_i, [_vp, _i]),
    'ifcbk_ctx_live_graphs': (_i, [_vp]),
    'ifcbk_last_error': (C.c_char_p, [_vp]),
    'ifcbk_conv2d_fwd': (_i, [_vp, C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp]),
    'ifcbk_conv2d_fwd_affine': (_i, [_vp, C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    'ifcbk_conv2d_fwd_affine_maxpool_ok': (_i, [C.POINTER(ConvDesc)]),
    'ifcbk_conv2d_fwd_affine_maxpool': (_i, [_vp, C.POINTER(ConvDesc), _vp, _vp, _vp, _i, _vp, _vp, _i, _vp]),
    'ifcbk_conv2d_dgrad': (_i, [_vp, C.POINTER(ConvDesc), _vp, _vp, _vp, _i, _vp]),
    'ifcbk_conv2d_wgrad': (_i, [_vp, C.POINTER(ConvDesc), _vp, _vp, _vp, _i, _vp]),
    'ifcbk_conv2d_wgrad_segments': (_i, [_vp, C.POINTER(ConvDesc), _vp, _vp, _i, C.POINTER(_vp), C.POINTER(C.c_int32), _i, _vp]),
    'ifcbk_bn_finalize_ld': (_i, [_vp, C.POINTER(BnDesc), _vp, _i, _i] + [_vp] * 9),
    'ifcbk_conv2d_wgrad_workspace': (_sz, [C.POINTER(ConvDesc)]),
    'ifcbk_conv2d_wgrad_group': (_i, [_vp, _i, C.POINTER(ConvDesc), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _i, _vp]),
    'ifcbk_conv2d_wgrad_group_workspace': (_sz, [_i, C.POINTER(ConvDesc)]),
    'ifcbk_conv2d_wgrad_group_member_kh': (_i, [C.POINTER(ConvDesc)]),
    'ifcbk_conv2d_wgrad_group_info': (_i, [_i, C.POINTER(ConvDesc), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    'ifcbk_conv2d_fwd_mblocks': (_i, [C.POINTER(ConvDesc)]),
    'ifcbk_weight_pack': (_i, [_vp, C.POINTER(ConvDesc), _vp, _vp, _vp, _vp]),
    'ifcbk_weight_pack_multi': (_i, [_vp, _vp, _i, C.c_int64, _i, _vp]),
    'ifcbk_bn_finalize': (_i, [_vp, C.POINTER(BnDesc), _vp, _i] + [_vp] * 9),
    'ifcbk_bn_apply': (_i, [_vp, C.POINTER(BnDesc), _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    'ifcbk_bn_bwd': (_i, [_vp, C.POINTER(BnDesc), _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _vp, _vp,
                          _i, _vp, _vp, _vp]),
    'ifcbk_conv2d_fwd_affine_segments': (_i, [_vp, C.POINTER(ConvDesc), _vp, _vp, _i, C.POINTER(_vp), C.POINTER(C.c_int32),
                                          C.POINTER(C.c_int32), C.POINTER(C.c_int32), _vp, _vp, _vp]),
    'ifcbk_avgpool3x3_affine': (_i, [_vp, C.POINTER(PoolDesc), _vp, _vp, _vp, _i, _vp, _vp]),
    'ifcbk_bn_stats_rows': (_i, [C.c_int64]),
    'ifcbk_bn_stats': (_i, [_vp, C.POINTER(BnDesc), _vp, _vp, _vp]),
    'ifcbk_conv2d_dgrad_bnstat_mblocks': (_i, [C.POINTER(ConvDesc)]),
    'ifcbk_conv2d_dgrad_bnstat': (_i, [_vp, C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'ifcbk_bn_bwd_partials': (_i, [_vp, C.POINTER(BnDesc), _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _i,
                                   _vp]),
    'ifcbk_bn_bwd_partials_ld': (_i, [_vp, C.POINTER(BnDesc), _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp,
                                      _i, _vp]),
    'ifcbk_conv2d_dgrad_bnstat_table': (_i, [_vp, C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp]),
    'ifcbk_bn_apply_maxpool': (_i, [_vp, C.POINTER(PoolDesc), _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    'ifcbk_bn_bwd_maxpool': (_i, [_vp, C.POINTER(PoolDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _i,
                                  _vp]),
    'ifcbk_maxpool_fwd': (_i, [_vp, C.POINTER(PoolDesc), _vp, _vp, _vp, _vp]),
    'ifcbk_maxpool_bwd': (_i, [_vp, C.POINTER(PoolDesc), _vp, _vp, _vp, _i, _vp]),
    'ifcbk_avgpool_fwd': (_i, [_vp, C.POINTER(PoolDesc), _vp, _vp, _vp]),
    'ifcbk_avgpool_bwd': (_i, [_vp, C.POINTER(PoolDesc), _vp, _vp, _i, _vp]),
    'ifcbk_head_fwd': (_i, [_vp, C.POINTER(HeadDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'ifcbk_head_bwd': (_i, [_vp, C.POINTER(HeadDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    'ifcbk_dropout_mask': (_i, [_vp, _vp, C.c_int64, _f, C.c_uint64, C.c_uint64, _vp]),
    'ifcbk_bias_relu_bwd': (_i, [_vp, C.c_int64, _i, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp]),
    'ifcbk_bias_relu_bwd_workspace': (_sz, [C.c_int64, _i]),
    'ifcbk_bias_relu_bwd_rows': (_i, [C.c_int64]),
    'ifcbk_dropout_apply': (_i, [_vp, C.c_int64, _i, _vp, _vp, _f, _vp, _i, _vp]),
    'ifcbk_flatten_chw': (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp]),
    'ifcbk_softmax_xent': (_i, [_vp, _vp, _vp, _i, _i, _f, _vp, _i, _vp, _vp]),
    'ifcbk_softmax': (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    'ifcbk_step_counters': (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    'ifcbk_adam_flat': (_i, [_vp, _vp, _vp, _vp, _vp, C.c_int64, _f, _f, _f, _f, _f, _i, _f, _vp]),
    'ifcbk_sgd_flat': (_i, [_vp, _vp, _vp, _vp, C.c_int64, _f, _f, _f, _f, _vp]),
    'ifcbk_roi_preprocess': (_i, [_vp, C.POINTER(RoiDesc), _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    'ifcbk_roi_preprocess_workspace': (_sz, [C.POINTER(RoiDesc), _i, _i]),
    'ifcbk_stem_u8_rows': (_i, [C.POINTER(ConvDesc)]),
    'ifcbk_stem_u8_fwd': (_i, [_vp, C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    'ifcbk_stem_u8_wgrad_workspace': (_sz, [C.POINTER(ConvDesc)]),
    'ifcbk_stem_u8_wgrad': (_i, [_vp, C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _i, _vp]),
    'ifcbk_nchw_to_nhwc': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, C.POINTER(_f), C.POINTER(_f), _vp, _vp]),
    'ifcbk_nhwc_to_nchw_f32': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'ifcbk_run_program': (_i, [_vp, C.POINTER(Op), _i, _vp, C.POINTER(_f)]),
    'ifcbk_run_program_ev': (_i, [_vp, C.POINTER(Op), _i, _vp, _i]),
    'ifcbk_program_times': (_i, [_vp, _i, _i, C.POINTER(_f)]),
    'ifcbk_program_capture': (_i, [_vp, C.POINTER(Op), _i, C.POINTER(_vp)]),
    'ifcbk_graph_launch': (_i, [_vp, _vp, _vp]),
    'ifcbk_graph_destroy': (_i, [_vp, _vp]),
    'ifcbk_op_kernel': (_i, [C.POINTER(Op), C.c_char_p, _sz]),
    'ifcbk_op_cost': (_i, [C.POINTER(Op), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}
EXPORTS = tuple(_PROTOS)

_lib = None


def load():
    """Load libifcbk.so (built by ``__graft_entry__.build()`` / ``make -C ifcb_classifier_amd/csrc``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('libifcbk.so not found at %s -- build it with `python -c "import __graft_entry__ as g; '
                               'g.build()"` (hipcc --offload-arch=gfx950); there is no CPU fallback' % LIB_PATH)
        # torch first: it brings its own libamdhip64 and the process must have ONE HIP runtime.  Loaded the other way round
        # (libifcbk.so, then torch) the library's hipGetDeviceCount saw "no ROCm-capable device" while torch.cuda worked --
        # build() followed by smoke() in one process hit exactly that
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class Context:
    """One ifcbk_ctx per (process, GPU).  ``call`` turns non-zero returns into RuntimeError."""

    def __init__(self, device=0):
        self.lib = load()
        h = _vp()
        rc = self.lib.ifcbk_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise RuntimeError('ifcbk_ctx_create(device=%d) failed with %d: %s' % (device, rc, self.lib.ifcbk_last_error(None).decode()))
        self.h = h
        self.device = device

    def call(self, name, *args):
        rc = getattr(self.lib, name)(self.h, *args)
        if rc != 0:
            raise RuntimeError('%s failed (%d): %s' % (name, rc, self.lib.ifcbk_last_error(self.h).decode()))

    def reserve(self, nbytes):
        self.call('ifcbk_ctx_reserve', int(nbytes))

    def run_program(self, ops, n, stream, op_ms=None):
        self.call('ifcbk_run_program', ops, int(n), stream, op_ms)

    def capture(self, ops, n):
        """record a program into a hipGraph (nothing runs); returns the graph handle for ``graph_launch``."""
        g = _vp()
        self.call('ifcbk_program_capture', ops, int(n), C.byref(g))
        return g

    def graph_launch(self, g, stream):
        self.call('ifcbk_graph_launch', g, stream)

    def live_graphs(self):
        return int(self.lib.ifcbk_ctx_live_graphs(self.h)) if self.h else 0

    def close(self):
        """destroy the context: its graphs first (the library's lifetime rule), then arenas, lane streams and events"""
        if getattr(self, 'h', None):
            self.lib.ifcbk_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PlanOnlyContext:
    """``Engine(plan_only=True)``: the library's pure host-side planning helpers (tile counts, workspace sizes, kernel names)
    without a device context; anything that would launch raises."""

    def __init__(self):
        self.lib = load()
        self.h = None
        self.device = None

    def reserve(self, nbytes):
        pass

    def close(self):
        pass

    def call(self, name, *args):
        raise RuntimeError('%s: this engine was built with plan_only=True (no HIP context)' % name)

    run_program = capture = graph_launch = call


def ptr(t):
    """raw device pointer of a torch tensor (or None)."""
    return None if t is None else _vp(t.data_ptr())


def cur_stream():
    import torch
    return _vp(torch.cuda.current_stream().cuda_stream)
