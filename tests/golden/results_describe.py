"""Shared by the fixture generator (runs the REFERENCE's writers under /opt/conda/bin/python3.9, which has h5py) and by the
tests (run OUR writers): deterministic inputs, and a plain-JSON description of a .json / .mat / .h5 results file --
names, dtypes, shapes, values, attributes -- so that schemas can be compared without shipping binary files."""
import json
import os

import numpy as np

CLASSES = ['Akashiwo', 'Bacillaria', 'Ceratium', 'detritus']
TIMESTAMP = '2021-05-04T03:02:01+00:00'
MODEL_ID = 'golden_model'


def scores(n, seed):
    rs = np.random.RandomState(seed)
    s = rs.rand(n, len(CLASSES)).astype('float32')
    return (s / s.sum(1, keepdims=True)).astype('float32')


def val_inputs():
    sc = scores(9, 11)
    in_cls = np.array([0, 1, 2, 3, 0, 1, 1, 3, 3])
    srcs = ['/data/%s/IFCB_%03d.png' % (CLASSES[c], i) for i, c in enumerate(in_cls)]
    train_images = ['/data/%s/T_%03d.png' % (CLASSES[c % 4], c) for c in range(14)]
    train_targets = [c % 4 for c in range(14)]
    return dict(outputs=sc, input_classes=in_cls, input_srcs=srcs, epoch=2, best=True, train_images=train_images,
                train_targets=train_targets, train_counts=[4, 4, 3, 3], val_counts=[2, 3, 1, 3])


VAL_SERIES = ('training_image_basenames training_classes image_basenames image_fullpaths input_classes output_scores '
              'output_winscores confusion_matrix counts_perclass val_counts_perclass train_counts_perclass f1_perclass '
              'recall_perclass f1_weighted f1_macro precision_macro classes_by_recall classes_by_count').split()


def _plain(v):
    if isinstance(v, bytes):
        return v.decode()
    if isinstance(v, np.ndarray):
        if v.dtype.kind in 'OSU':
            return [_plain(x) for x in v.ravel().tolist()]
        return np.round(v.astype('float64'), 4).tolist()
    if isinstance(v, (np.floating, float)):
        return round(float(v), 4)
    if isinstance(v, (np.integer, int)):
        return int(v)
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    return v


def describe(path):
    ext = os.path.splitext(path)[1]
    if ext == '.json':
        def rnd(o):
            if isinstance(o, float):
                return round(o, 4)
            if isinstance(o, list):
                return [rnd(x) for x in o]
            if isinstance(o, dict):
                return {k: rnd(v) for k, v in o.items()}
            return o
        return dict(kind='json', content=rnd(json.load(open(path))))
    if ext == '.mat':
        from scipy.io import loadmat
        m = loadmat(path)
        out = {}
        for k in sorted(m):
            if k.startswith('__'):
                continue
            v = m[k]
            flat = [(_plain(x.ravel()[0]) if isinstance(x, np.ndarray) and x.size == 1 else _plain(x)) for x in v.ravel()] \
                if v.dtype.kind == 'O' else _plain(v)
            out[k] = dict(dtype=str(v.dtype), shape=list(v.shape), values=flat)
        return dict(kind='mat', vars=out)
    if ext == '.h5':
        import h5py
        out = {}
        with h5py.File(path, 'r') as f:
            for k in sorted(f):
                d = f[k]
                ent = dict(dtype=str(d.dtype), shape=(None if d.shape is None else list(d.shape)), compression=d.compression,
                           attrs={a: _plain(d.attrs[a]) for a in sorted(d.attrs)})
                if d.shape is not None:
                    ent['values'] = _plain(d[()])
                out[k] = ent
        return dict(kind='h5', datasets=out)
    raise ValueError(path)
