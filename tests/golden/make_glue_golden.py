"""Generates tests/golden/dataset_glue.json by running the REFERENCE's own pure-Python dataset logic
(/root/reference/neuston_data.py: NeustonDataset, .split, .from_csv, parse_imgnorm) in this container, with stub
modules for its missing third-party imports (torchvision, ifcb).  Only inputs/outputs are stored.

  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_glue_golden.py
"""
import json
import os
import random
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))


def stub_modules():
    tv = types.ModuleType('torchvision')
    tr = types.ModuleType('torchvision.transforms')
    ds = types.ModuleType('torchvision.datasets')
    fo = types.ModuleType('torchvision.datasets.folder')
    fo.IMG_EXTENSIONS = ('.jpg', '.jpeg', '.png', '.ppm', '.bmp', '.pgm', '.tif', '.tiff', '.webp')
    ds.folder = fo

    class ImageFolder:
        pass
    ds.ImageFolder = ImageFolder
    tv.transforms, tv.datasets = tr, ds
    ifcb = types.ModuleType('ifcb')
    data = types.ModuleType('ifcb.data')
    adc = types.ModuleType('ifcb.data.adc')
    adc.SCHEMA_VERSION_1 = 'v1'
    st = types.ModuleType('ifcb.data.stitching')
    st.InfilledImages = object
    for name, m in (('torchvision', tv), ('torchvision.transforms', tr), ('torchvision.datasets', ds),
                    ('torchvision.datasets.folder', fo), ('ifcb', ifcb), ('ifcb.data', data), ('ifcb.data.adc', adc),
                    ('ifcb.data.stitching', st)):
        sys.modules[name] = m


LAYOUT = {'Akashiwo': 17, 'Bacillaria': 5, 'Ceratium': 1, 'Ditylum': 40, 'Euglena': 2, 'detritus': 23}


def make_tree(root):
    for cls, n in LAYOUT.items():
        os.makedirs(os.path.join(root, cls))
        for i in range(n):
            open(os.path.join(root, cls, 'IFCB_%s_%03d.png' % (cls[:3], i)), 'w').close()
        open(os.path.join(root, cls, 'notes.txt'), 'w').close()        # non-image file must be ignored


def main():
    stub_modules()
    sys.path.insert(0, '/root/reference')
    import neuston_data as nd                                           # the reference itself
    out = {'layout': LAYOUT, 'imgnorm': [], 'datasets': []}
    for arg in (['0.667', '0.161'], ['0.056,0.058,0.051', '0.067,0.071,0.057'], ['1', '2,3,4']):
        out['imgnorm'].append(dict(arg=arg, result=list(nd.parse_imgnorm(arg))))
    with tempfile.TemporaryDirectory() as root:
        make_tree(root)
        rel = lambda p: os.path.relpath(p, root)
        csvf = os.path.join(root, 'cfg.csv')
        with open(csvf, 'w') as f:
            f.write('class,v1\nAkashiwo,1\nBacillaria,0\nCeratium,GROUP\nDitylum,1\nEuglena,GROUP\nmissing_cls,1\n')
        for cmin, cmax, seed, split, use_csv in ((2, None, 7, (80, 20), False), (1, None, 3, (50, 50), False),
                                                 (2, 10, 11, (80, 20), False), (6, None, 1, (90, 10), False),
                                                 (2, None, 5, (80, 20), True), (2, None, 0, (100, 0), False)):
            random.seed(1000 + (seed or 0))
            if use_csv:
                ds = nd.NeustonDataset.from_csv(root, csvf, 'v1', minimum_images_per_class=cmin, maximum_images_per_class=cmax)
            else:
                ds = nd.NeustonDataset(root, minimum_images_per_class=cmin, maximum_images_per_class=cmax)
            random.seed(2000 + (seed or 0))
            try:
                d1, d2 = ds.split(split[0], split[1], seed=seed)
            except AssertionError as e:                                 # the reference's own edge cases
                out['datasets'].append(dict(class_min=cmin, class_max=cmax, seed=seed, split=list(split), csv=use_csv,
                                            classes=ds.classes, error='AssertionError'))
                continue
            out['datasets'].append(dict(
                class_min=cmin, class_max=cmax, seed=seed, split=list(split), csv=use_csv, classes=ds.classes,
                ignored=[list(t) for t in ds.classes_ignored_from_too_few_samples],
                images=[rel(p) for p in ds.images], targets=list(ds.targets), count_perclass=ds.count_perclass,
                train=[rel(p) for p in d1.images], train_targets=list(d1.targets),
                val=[rel(p) for p in d2.images], val_targets=list(d2.targets)))
    json.dump(out, open(os.path.join(HERE, 'dataset_glue.json'), 'w'))
    print('written', len(out['datasets']), 'dataset cases')


if __name__ == '__main__':
    main()
