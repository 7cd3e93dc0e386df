"""Generates the committed golden vectors (run in the build container only; PIL is the generator).

  python tests/golden/make_golden.py

pil_resize_cases.npz : ragged u8 ROIs and Pillow's own ``convert('RGB').resize((S,S), BILINEAR)`` outputs --
                       the exact call chain of neuston_data.py:456-464 -- full arrays for small cases,
                       sha256 digests for the rest.
model_keys.json      : state_dict key -> shape of the oracle graphs (pins [TV] naming/registration order).
"""
import hashlib
import json
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

CASES = [(1, 1), (2, 1000), (600, 80), (37, 53), (299, 299), (300, 298), (45, 212), (150, 150), (1024, 33),
         (64, 64), (224, 224), (500, 400)]
FULL = {(1, 1), (37, 53), (2, 1000), (64, 64)}


def pil_chain(a, S):
    im = Image.fromarray(a, 'L').convert('RGB').resize((S, S), Image.BILINEAR)
    return np.asarray(im)


def main():
    rng = np.random.default_rng(1234)
    out = {}
    meta = []
    for k, (h, w) in enumerate(CASES):
        a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        if (h, w) == (64, 64):
            a[:] = np.add.outer(np.arange(64), np.arange(64)).astype(np.uint8) * 2   # smooth ramp
        out['in_%d' % k] = a
        for S in (299, 224):
            r = pil_chain(a, S)
            assert (r[..., 0] == r[..., 1]).all() and (r[..., 0] == r[..., 2]).all()
            dig = hashlib.sha256(r[..., 0].tobytes()).hexdigest()
            meta.append(dict(case=k, h=h, w=w, S=S, sha256=dig, full=(h, w) in FULL))
            if (h, w) in FULL:
                out['out_%d_%d' % (k, S)] = r[..., 0].copy()
    # an RGB case (NeustonDataset / ImageDataset path: default_loader -> RGB)
    rgb = rng.integers(0, 256, (41, 67, 3), dtype=np.uint8)
    out['rgb_in'] = rgb
    out['rgb_out_299'] = np.asarray(Image.fromarray(rgb, 'RGB').resize((299, 299), Image.BILINEAR))
    np.savez_compressed(os.path.join(HERE, 'pil_resize_cases.npz'), **out)
    import PIL
    json.dump(dict(pillow=PIL.__version__, cases=meta), open(os.path.join(HERE, 'pil_resize_cases.json'), 'w'), indent=1)

    from oracle import tv_models
    keys = {}
    for name, nc in (('inception_v3', 100), ('resnet18', 2), ('resnet50', 1000)):
        m = tv_models.get_namebrand_model(name, nc)
        keys[name + ':%d' % nc] = dict(params=sum(p.numel() for p in m.parameters()),
                                       state_dict=[[k, list(v.shape)] for k, v in m.state_dict().items()])
    json.dump(keys, open(os.path.join(HERE, 'model_keys.json'), 'w'))
    print('golden written:', os.listdir(HERE))


if __name__ == '__main__':
    main()
