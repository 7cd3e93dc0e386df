"""CPU suite (-m "not gpu"): pins the oracle against golden vectors and live PIL, checks host logic and
that the C-ABI library loads and exports every symbol include/ifcbk.h declares (no compute without a GPU)."""
import ctypes
import hashlib
import json
import os
import re

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, 'golden')


def test_pil_resize_restatement_matches_golden_and_live_pil():
    from oracle.pil_resize import resize_bilinear_u8
    z = np.load(os.path.join(GOLD, 'pil_resize_cases.npz'))
    meta = json.load(open(os.path.join(GOLD, 'pil_resize_cases.json')))
    for c in meta['cases']:
        if c['h'] * c['w'] > 120000 and c['S'] == 224:
            continue                                        # keep the CPU suite short
        a = z['in_%d' % c['case']]
        r = resize_bilinear_u8(a, c['S'], c['S'])
        assert hashlib.sha256(r.tobytes()).hexdigest() == c['sha256'], c
        if c['full']:
            assert np.array_equal(r, z['out_%d_%d' % (c['case'], c['S'])])
    r = resize_bilinear_u8(z['rgb_in'], 299, 299)
    assert np.array_equal(r, z['rgb_out_299'])
    # live PIL on fresh random ROIs (upscale, downscale, mixed)
    from PIL import Image
    rng = np.random.default_rng(7)
    for h, w in ((33, 290), (310, 40), (299, 17), (5, 5)):
        a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(a, 'L').resize((299, 299), Image.BILINEAR))
        assert np.array_equal(resize_bilinear_u8(a, 299, 299), ref)


def test_roi_to_tensor_matches_reference_chain():
    """neuston_data.py:456-464: ToPILImage('L') -> convert('RGB') -> Resize -> ToTensor -> Normalize."""
    from oracle.pil_resize import roi_to_tensor
    from PIL import Image
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (57, 131), dtype=np.uint8)
    im = Image.fromarray(a, 'L').convert('RGB').resize((299, 299), Image.BILINEAR)
    ref = torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).float().div(255)
    mean, std = [0.5, 0.4, 0.3], [0.2, 0.25, 0.3]
    refn = (ref - torch.tensor(mean)[:, None, None]) / torch.tensor(std)[:, None, None]
    assert np.array_equal(roi_to_tensor(a, 299), ref.numpy())
    assert np.allclose(roi_to_tensor(a, 299, mean, std), refn.numpy(), rtol=0, atol=1e-6)
    # flips are applied to the source image before the resize ('x' = vertical, 'y' = horizontal)
    imf = Image.fromarray(a[::-1].copy(), 'L').convert('RGB').resize((299, 299), Image.BILINEAR)
    assert np.array_equal(roi_to_tensor(a, 299, flip_v=True), np.asarray(imf).transpose(2, 0, 1).astype(np.float32) / np.float32(255))


def test_oracle_graphs_match_published_counts_and_golden_keys():
    from oracle import tv_models
    assert sum(p.numel() for p in tv_models.Inception3(1000).parameters()) == 27161264
    assert sum(p.numel() for p in tv_models.ResNet(*tv_models._RESNETS['resnet18'], 1000).parameters()) == 11689512
    gold = json.load(open(os.path.join(GOLD, 'model_keys.json')))
    for tag, g in gold.items():
        name, nc = tag.split(':')
        m = tv_models.get_namebrand_model(name, int(nc))
        assert sum(p.numel() for p in m.parameters()) == g['params']
        assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == g['state_dict']
    assert gold['inception_v3:100']['params'] == 24625064
    assert gold['resnet18:2']['params'] == 11177538
    with pytest.raises(KeyError, match='model unknown'):
        tv_models.get_namebrand_model('efficientnet_b4', 3)


def test_hip_graph_matches_oracle_keys_shapes_and_order():
    """the HIP plan's parameter table == oracle state_dict (keys, OIHW shapes, registration order)."""
    from ifcb_classifier_amd import graph
    gold = json.load(open(os.path.join(GOLD, 'model_keys.json')))
    for tag, g in gold.items():
        name, nc = tag.split(':')
        net = graph.build(name, int(nc))
        pk = [[k, list(s)] for k, s, _, _ in net.params]
        ok = [kv for kv in g['state_dict'] if not re.search(r'running_|num_batches', kv[0])]
        assert pk == ok
        bk = [k for k, _, _ in net.buffers]
        assert bk == [kv[0] for kv in g['state_dict'] if 'running_' in kv[0]]
    with pytest.raises(KeyError, match='model unknown'):
        graph.build('efficientnet_b4', 3)


def test_graph_mac_totals_match_survey():
    """5.7164 GMAC train-forward / image at NC=100; 1.8136 GMAC resnet18 NC=2."""
    from ifcb_classifier_amd import graph

    def macs(net, train=True):
        t = 0
        for n in net.nodes:
            if n.kind == 'conv' and (train or not n.aux):
                t += n.P * n.Q * n.K * n.R * n.S * n.Cw
            elif n.kind == 'head' and (train or not n.aux):
                t += n.C * n.NC
        return t
    net = graph.build('inception_v3', 100)
    assert abs(macs(net) / 1e9 - 5.7164) < 5e-4
    assert abs(macs(net, False) / 1e9 - 5.7114) < 5e-4
    assert abs(macs(graph.build('resnet18', 2)) / 1e9 - 1.8136) < 5e-4


def test_inception_concat_slices_cover_block_outputs_exactly():
    from ifcb_classifier_amd import graph
    net = graph.build('inception_v3', 10)
    cover = {}
    for n in net.nodes:
        if n.kind in ('conv', 'max', 'avg'):
            cover.setdefault(n.y.buf.id, []).append((n.y.coff, n.y.coff + n.y.C))
    for bid, spans in cover.items():
        spans.sort()
        assert spans[0][0] == 0 and spans[-1][1] == net.bufs[bid].C
        for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
            assert a1 == b0, (net.bufs[bid].name, spans)


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    from ifcb_classifier_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'ifcbk.h')).read()
    declared = set(re.findall(r'\b(ifcbk_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'ifcbk_ctx', 'ifcbk_op'}
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for sym in sorted(declared):
        assert hasattr(lib, sym), 'missing export ' + sym
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert _lib.load().ifcbk_version().startswith(b'ifcbk')


def test_c_abi_struct_sizes_match_header_layout():
    from ifcb_classifier_amd import _lib
    assert ctypes.sizeof(_lib.ConvDesc) == 17 * 4
    assert ctypes.sizeof(_lib.BnDesc) == 8 * 4
    assert ctypes.sizeof(_lib.PoolDesc) == 15 * 4
    assert ctypes.sizeof(_lib.HeadDesc) == 7 * 4
    assert ctypes.sizeof(_lib.RoiDesc) == 6 * 4 + 12 * 4
    assert ctypes.sizeof(_lib.Op) == 8 + 12 * 8 + 4 * 8 + 8 * 4 + 17 * 4 + 4   # union padded to 8-byte alignment


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, 'ifcb_classifier_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), f


def test_engine_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from ifcb_classifier_amd.neuston_models import get_namebrand_model
    with pytest.raises(RuntimeError, match='HIP device'):
        get_namebrand_model('resnet18', 2)


# ---------------------------------------------------------------------------- dataset glue vs the reference
def _make_tree(root, layout):
    for cls, n in layout.items():
        os.makedirs(os.path.join(root, cls))
        for i in range(n):
            open(os.path.join(root, cls, 'IFCB_%s_%03d.png' % (cls[:3], i)), 'w').close()
        open(os.path.join(root, cls, 'notes.txt'), 'w').close()


def test_dataset_glue_matches_reference_goldens(tmp_path):
    """classes / class-min / class-max / class-config csv / split lists equal the outputs of the REFERENCE's own
    neuston_data.py (captured by tests/golden/make_glue_golden.py) for the same tree, args and seeds."""
    import random
    from ifcb_classifier_amd import neuston_data as nd
    gold = json.load(open(os.path.join(GOLD, 'dataset_glue.json')))
    for g in gold['imgnorm']:
        assert [list(x) for x in nd.parse_imgnorm(g['arg'])] == g['result']
    with pytest.raises(AssertionError, match='img-norm invalid'):
        nd.parse_imgnorm(['1,2', '3'])
    root = str(tmp_path / 'tree')
    os.makedirs(root)
    _make_tree(root, gold['layout'])
    rel = lambda p: os.path.relpath(p, root)
    csvf = os.path.join(root, 'cfg.csv')
    with open(csvf, 'w') as f:
        f.write('class,v1\nAkashiwo,1\nBacillaria,0\nCeratium,GROUP\nDitylum,1\nEuglena,GROUP\nmissing_cls,1\n')
    assert len(gold['datasets']) >= 5
    for g in gold['datasets']:
        seed = g['seed']
        random.seed(1000 + (seed or 0))
        if g['csv']:
            ds = nd.NeustonDataset.from_csv(root, csvf, 'v1', minimum_images_per_class=g['class_min'],
                                            maximum_images_per_class=g['class_max'])
        else:
            ds = nd.NeustonDataset(root, minimum_images_per_class=g['class_min'], maximum_images_per_class=g['class_max'])
        assert ds.classes == g['classes']
        random.seed(2000 + (seed or 0))
        if g.get('error'):
            with pytest.raises(AssertionError):
                ds.split(g['split'][0], g['split'][1], seed=seed)
            continue
        d1, d2 = ds.split(g['split'][0], g['split'][1], seed=seed)
        assert [list(t) for t in ds.classes_ignored_from_too_few_samples] == g['ignored']
        assert [rel(p) for p in ds.images] == g['images'] and list(ds.targets) == g['targets']
        assert ds.count_perclass == g['count_perclass']
        assert [rel(p) for p in d1.images] == g['train'] and list(d1.targets) == g['train_targets']
        assert [rel(p) for p in d2.images] == g['val'] and list(d2.targets) == g['val_targets']


def test_transform_spec_and_cli_surface():
    import argparse
    from ifcb_classifier_amd import neuston_data as nd
    from ifcb_classifier_amd import neuston_net as nn_
    a = argparse.Namespace(MODEL='inception_v3', img_norm=['0.5', '0.25'], flip='x+V')
    tr, va = nd.get_trainval_transforms(a)
    assert a.resize == 299 and tr.vflip and not tr.hflip and va.vflip and tr.img_norm == ([0.5] * 3, [0.25] * 3)
    a = argparse.Namespace(MODEL='inception_v3_foo', img_norm=None, flip='xy')
    tr, va = nd.get_trainval_transforms(a)
    assert a.resize == 224 and tr.vflip and tr.hflip and not va.vflip and not va.hflip and tr.img_norm is None
    p = nn_.argparse_nn()
    t = p.parse_args(['--batch', '32', 'TRAIN', 'src', 'inception_v3', 'id1'])
    assert (t.batch_size, t.loaders, t.pretrained, t.split, t.class_min, t.emax, t.emin, t.estop, t.seed) == \
        (32, 4, True, '80:20', 2, 60, 10, 10, 0)
    assert t.outdir == 'training-output/{TRAIN_ID}' and t.model_id == '{TRAIN_ID}' and t.epochs_log == 'epochs.csv'
    r = p.parse_args(['RUN', 'src', 'm.ptl', 'rid', '--type', 'img', '--filter', 'IN', 'abc'])
    assert (r.batch_size, r.src_type, r.outdir, r.clobber, r.filter) == (108, 'img', 'run-output/{RUN_ID}/v3/{MODEL_ID}', False, ['IN', 'abc'])


def test_collate_rois_layout():
    from ifcb_classifier_amd.neuston_data import collate_rois
    a = np.arange(6, dtype=np.uint8).reshape(2, 3)
    b = np.arange(20, dtype=np.uint8).reshape(5, 4)
    batch, tg, paths = collate_rois([((a, 0), 1, 'p0'), ((b, 3), 0, 'p1')])
    assert batch['in_channels'] == 1 and batch['offs'].tolist() == [0, 6] and batch['hs'].tolist() == [2, 5]
    assert batch['ws'].tolist() == [3, 4] and batch['flips'].tolist() == [0, 3] and (batch['max_h'], batch['max_w']) == (5, 4)
    assert batch['pixels'].tolist() == list(range(6)) + list(range(20)) and tg.tolist() == [1, 0] and paths == ['p0', 'p1']
    c = np.zeros((2, 2, 3), np.uint8)
    batch, ids = collate_rois([((a, 0), 'x'), ((c, 0), 'y')])
    assert batch['in_channels'] == 3 and batch['offs'].tolist() == [0, 18] and batch['pixels'].numel() == 18 + 12


def test_ifcb_bin_reader_and_pid(tmp_path):
    from ifcb_classifier_amd.ifcb_bins import DataDirectory, Pid
    d = tmp_path / 'D2013' / 'D20130526'
    d.mkdir(parents=True)
    lid = 'D20130526T092352_IFCB013'
    rois = [np.full((3, 4), 7, np.uint8), np.zeros((0, 0), np.uint8), np.arange(10, dtype=np.uint8).reshape(2, 5)]
    blob, lines, off = b'', [], 0
    for r in rois:
        h, w = r.shape
        cols = ['0'] * 24
        cols[13], cols[14], cols[15], cols[16], cols[17] = '1', '2', str(w), str(h), str(off)
        lines.append(','.join(cols))
        blob += r.tobytes()
        off += h * w
    (d / (lid + '.adc')).write_text('\n'.join(lines) + '\n')
    (d / (lid + '.roi')).write_bytes(blob)
    bins = list(DataDirectory(str(tmp_path)))
    assert len(bins) == 1 and bins[0].pid.pid == lid and bins[0].pid.year == '2013' and bins[0].pid.yearday == '20130526'
    imgs = bins[0].images
    assert sorted(imgs) == [1, 3] and np.array_equal(imgs[3], rois[2]) and imgs[1].shape == (3, 4)
    assert bins[0].pid.with_target(3).pid == lid + '_00003' and bins[0].pid.with_target(3).target == 3
    old = Pid('IFCB1_2010_025_134132')
    assert (old.year, old.yearday, old.schema) == ('2010', '2010_025', 'v1')
    assert list(DataDirectory(str(tmp_path), blacklist=['IFCB013'])) == []
