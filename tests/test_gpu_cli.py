"""End-to-end drop-in surface on the GPU: ``neuston_net TRAIN`` (BASELINE config 0 shape: resnet18, 2 classes,
synthetic ROIs as PNG files) then ``RUN --type img`` and ``RUN --type bin`` with the trained .ptl."""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make_dataset(root, per_class=128):
    """BASELINE.json configs[0] / SURVEY 8(d) config 1: 256 PNGs = 2 class folders x 128 u8 grey ROIs, h, w ~ U{32..256} (seed 1234),
    pixels = class mean (96 / 160) + N(0, 32) clipped"""
    from PIL import Image
    rng = np.random.default_rng(1234)
    for cls, mean in (('cls_dark', 96), ('cls_bright', 160)):
        os.makedirs(os.path.join(root, cls))
        for i in range(per_class):
            h, w = rng.integers(32, 257, 2)
            a = np.clip(rng.normal(mean, 32, (h, w)), 0, 255).astype(np.uint8)
            Image.fromarray(a, 'L').save(os.path.join(root, cls, 'roi_%s_%03d.png' % (cls, i)))


def _cli(argv):
    from ifcb_classifier_amd import neuston_net as nn_
    args = nn_.argparse_nn().parse_args(argv)
    nn_.argparse_nn_runtimeparams(args)
    nn_.main(args)
    return args


def test_train_then_run_img_and_bin(tmp_path, capsys):
    src = str(tmp_path / 'training-data')
    os.makedirs(src)
    _make_dataset(src)
    outdir = str(tmp_path / 'training-output' / 'smoke')
    # (the configs[0] command line of SURVEY 8(d): --batch 32 --loaders 0 TRAIN <dir> resnet18 smoke --untrain --seed 1 --emax 2 --emin 1 --estop 0)
    args = _cli(['--batch', '32', '--loaders', '0', 'TRAIN', src, 'resnet18', 'smoke', '--untrain', '--seed', '1',
                 '--emax', '2', '--emin', '1', '--estop', '0', '--outdir', outdir, '--flip', 'xy',
                 '--results', 'results.json', 'image_basenames', 'output_scores',
                 'confusion_matrix', 'f1_macro', '--results', 'results.mat', 'counts_perclass', 'f1_perclass'])
    for f in ('smoke.ptl', 'epochs.csv', 'args.yml', 'training_images.list', 'validation_images.list', 'results.json',
              'results.mat'):
        assert os.path.isfile(os.path.join(outdir, f)), f
    rows = open(os.path.join(outdir, 'epochs.csv')).read().strip().splitlines()
    assert rows[0].split(',')[:4] == ['epoch', 'best', 'train_loss', 'val_loss'] and len(rows) == 3
    tl = [float(r.split(',')[2]) for r in rows[1:]]
    assert tl[-1] < tl[0], tl                                          # it learns the two brightness classes
    assert len(open(os.path.join(outdir, 'training_images.list')).read().splitlines()) == 204     # 80:20 of 2 x 128
    assert len(open(os.path.join(outdir, 'validation_images.list')).read().splitlines()) == 52
    res = json.load(open(os.path.join(outdir, 'results.json')))
    assert res['class_labels'] == ['cls_bright', 'cls_dark'] and len(res['output_scores']) == 52
    assert np.allclose(np.sum(res['output_scores'], 1), 1.0, atol=1e-4)
    ck = torch.load(os.path.join(outdir, 'smoke.ptl'), map_location='cpu', weights_only=False)
    assert ck['hyper_parameters']['model_id'] == 'smoke' and ck['hyper_parameters']['resize'] == 224
    assert ck['hyper_parameters']['classes'] == ['cls_bright', 'cls_dark']
    assert list(ck['state_dict'])[0] == 'model.conv1.weight' and tuple(ck['state_dict']['model.conv1.weight'].shape) == (64, 3, 7, 7)

    # ---- RUN --type img
    run_out = str(tmp_path / 'run-output')
    _cli(['--batch', '32', '--loaders', '0', 'RUN', src, os.path.join(outdir, 'smoke.ptl'), 'r1', '--type', 'img',
          '--outdir', run_out + '/{RUN_ID}/v3/{MODEL_ID}', '--outfile', 'img_results.json', '--outfile', 'img_results.mat'])
    rj = json.load(open(os.path.join(run_out, 'r1', 'v3', 'smoke', 'img_results.json')))
    assert rj['version'] == 'v3' and rj['model_id'] == 'smoke' and len(rj['input_images']) == 256
    scores = np.array(rj['output_scores'])
    assert scores.shape == (256, 2) and np.allclose(scores.sum(1), 1, atol=1e-4)
    # RUN must reproduce, image by image, the validation scores TRAIN saved for the checkpointed (best) epoch:
    # same weights + same eval-mode BatchNorm => same probabilities (eval BN makes samples independent of batching)
    by_name = {os.path.splitext(os.path.basename(p))[0]: s for p, s in zip(rj['input_images'], rj['output_scores'])}
    for name, sc in zip(res['image_basenames'], res['output_scores']):
        assert np.allclose(by_name[name], sc, atol=2e-3), (name, by_name[name], sc)
    assert (np.array(rj['output_classes']) == scores.argmax(1)).all()
    assert os.path.isfile(os.path.join(run_out, 'r1', 'v3', 'smoke', 'img_results.mat'))

    # ---- RUN --type bin on a synthetic raw bin; clobber-skip on the second pass
    bdir = tmp_path / 'run-data' / 'D2013' / 'D20130526'
    bdir.mkdir(parents=True)
    lid = 'D20130526T092352_IFCB013'
    rng = np.random.default_rng(5)
    blob, lines, off = b'', [], 0
    for k in range(21):
        h, w = (int(v) for v in rng.integers(20, 90, 2))
        if k == 4:
            h = w = 0
        a = np.clip(rng.normal(96 if k % 2 else 160, 30, (h, w)), 0, 255).astype(np.uint8)
        cols = ['0'] * 24
        cols[15], cols[16], cols[17] = str(w), str(h), str(off)
        lines.append(','.join(cols))
        blob += a.tobytes()
        off += h * w
    (bdir / (lid + '.adc')).write_text('\n'.join(lines) + '\n')
    (bdir / (lid + '.roi')).write_bytes(blob)
    argv = ['--batch', '8', '--loaders', '0', 'RUN', str(tmp_path / 'run-data'), os.path.join(outdir, 'smoke.ptl'), 'r2',
            '--outdir', run_out + '/{RUN_ID}/v3/{MODEL_ID}', '--outfile', 'D{BIN_YEAR}/D{BIN_DATE}/{BIN_ID}_class.json']
    _cli(argv)
    bj = json.load(open(os.path.join(run_out, 'r2', 'v3', 'smoke', 'D2013', 'D20130526', lid + '_class.json')))
    assert bj['bin_id'] == lid and bj['roi_numbers'] == [n for n in range(1, 22) if n != 5]
    assert np.array(bj['output_scores']).shape == (20, 2)
    assert np.allclose(np.array(bj['output_scores']).sum(1), 1, atol=1e-4)
    capsys.readouterr()
    _cli(argv)
    assert 'already exist - skipping this bin' in capsys.readouterr().out
    # ---- f-3: the default bin path above uploaded the .roi blob ONCE and cut the batches on the device from the ADC table
    # (Trainer.test_resident); the per-ROI DataLoader path (the reference's shape, neuston_data.py:433-467) writes the same bits.
    # pyifcb is absent: what an .adc column means is this build's reading of the file layout -- PARITY UNPINNED (ifcb_bins.py)
    import ifcb_classifier_amd.neuston_net as nn_mod
    calls = []
    orig = nn_mod.Trainer.test_resident
    nn_mod.Trainer.test_resident = lambda self, *a, **k: (calls.append(1), orig(self, *a, **k))[1]
    try:
        os.environ['IFCBK_BIN_RESIDENT'] = '0'
        argv0 = list(argv)
        argv0[argv0.index('r2')] = 'r3'
        _cli(argv0)
        assert not calls
        os.environ.pop('IFCBK_BIN_RESIDENT')
        argv1 = list(argv)
        argv1[argv1.index('r2')] = 'r4'
        _cli(argv1)
        assert calls
    finally:
        nn_mod.Trainer.test_resident = orig
        os.environ.pop('IFCBK_BIN_RESIDENT', None)
    b3 = json.load(open(os.path.join(run_out, 'r3', 'v3', 'smoke', 'D2013', 'D20130526', lid + '_class.json')))
    b4 = json.load(open(os.path.join(run_out, 'r4', 'v3', 'smoke', 'D2013', 'D20130526', lid + '_class.json')))
    assert b3['roi_numbers'] == b4['roi_numbers'] == bj['roi_numbers']
    assert b3['output_scores'] == b4['output_scores'] == bj['output_scores'] and b3['output_classes'] == b4['output_classes']


def test_ddp_step_world1_equals_fused_step():
    """train_step_ddp with a no-op all-reduce and world=1 must produce bitwise the same update as train_step."""
    from ifcb_classifier_amd import graph
    from ifcb_classifier_amd.engine import Engine
    net = graph.build('resnet18', 3)
    eng = Engine(net, 0, max_batch=4)
    eng.init_weights(seed=5)
    x = torch.rand(4, 3, 224, 224, device='cuda')
    eng.target[:4].copy_(torch.tensor([0, 1, 2, 1]))
    p0 = eng.P.clone()
    eng.load_input_nchw(x)
    eng.train_step(4)
    torch.cuda.synchronize()
    p1, l1 = eng.P.clone(), eng.loss.item()
    eng.P.copy_(p0); eng.M.zero_(); eng.V.zero_(); eng.step_count = 0; eng.params_changed()
    eng.RB.zero_()
    calls = []
    eng.load_input_nchw(x)
    eng.train_step_ddp(4, 1, lambda t: calls.append(t.numel()))
    torch.cuda.synchronize()
    assert len(calls) >= 2 and sum(calls) == eng.nparam_padded
    assert eng.loss.item() == l1 and torch.equal(eng.P, p1)


def _make_bin(bdir, lid, n, seed):
    rng = np.random.default_rng(seed)
    blob, lines, off = b'', [], 0
    for k in range(n):
        h, w = (int(v) for v in rng.integers(20, 90, 2))
        a = np.clip(rng.normal(96 if k % 2 else 160, 30, (h, w)), 0, 255).astype(np.uint8)
        cols = ['0'] * 24
        cols[15], cols[16], cols[17] = str(w), str(h), str(off)
        lines.append(','.join(cols))
        blob += a.tobytes()
        off += h * w
    (bdir / (lid + '.adc')).write_text('\n'.join(lines) + '\n')
    (bdir / (lid + '.roi')).write_bytes(blob)


def test_checkpoint_layout_gobig_and_h5_preflight(tmp_path, capsys):
    """f-2: the .ptl carries the [PL] 1.3.8 keys (optimizer_states in parameters() order, hparams_name/type) and restores the
    optimizer; RUN --gobig (one stream of full batches across bins, reference neuston_net.py:261-263,271) writes the same
    per-bin files as the bin-at-a-time run; the default .h5 outfile fails fast when h5py is missing (ADVICE)."""
    import argparse
    from ifcb_classifier_amd.neuston_models import NeustonModel
    hp = argparse.Namespace(MODEL='resnet18', classes=['a', 'b', 'c'], pretrained=False, batch_size=8, model_id='ck', resize=224,
                            img_norm=None, seed=3, cmd_timestamp='2021-01-01T00:00:00+00:00')
    torch.manual_seed(5)
    m = NeustonModel(hp)
    x = torch.rand(8, 3, 224, 224).cuda()
    y = torch.randint(0, 3, (8,)).cuda()
    for _ in range(2):
        m.fit_batch(x, y)
    torch.cuda.synchronize()
    ck = m.checkpoint_dict(epoch=1, global_step=2)
    for key in ('state_dict', 'hyper_parameters', 'epoch', 'global_step', 'optimizer_states', 'callbacks', 'lr_schedulers',
                'pytorch-lightning_version', 'hparams_name', 'hparams_type'):
        assert key in ck, key
    st = ck['optimizer_states'][0]
    names = [k for k, _ in m.named_parameters()]
    assert st['param_groups'][0]['params'] == list(range(len(names))) and st['param_groups'][0]['lr'] == 0.001
    assert tuple(st['state'][0]['exp_avg'].shape) == (64, 3, 7, 7) and st['state'][0]['step'] == 2
    # a torch Adam built over the module's parameters accepts it (what Lightning does on resume)
    opt = torch.optim.Adam(m.parameters(), lr=0.001)
    opt.load_state_dict(st)
    path = str(tmp_path / 'ck.ptl')
    torch.save(ck, path)
    m2 = NeustonModel.load_from_checkpoint(path, max_batch=8)
    e1, e2 = m.model.engine, m2.model.engine
    assert torch.equal(e1.P, e2.P) and torch.equal(e1.M, e2.M) and torch.equal(e1.V, e2.V) and e2.step_count == 2
    assert torch.equal(e1.RB, e2.RB)
    m.fit_batch(x, y); m2.fit_batch(x, y)
    torch.cuda.synchronize()
    assert torch.equal(e1.P, e2.P)                     # training resumes bit-identically from the file

    # ---- RUN on two bins: bin-at-a-time vs --gobig
    bdir = tmp_path / 'run-data' / 'D2013' / 'D20130526'
    bdir.mkdir(parents=True)
    lids = ['D20130526T092352_IFCB013', 'D20130526T101500_IFCB013']
    _make_bin(bdir, lids[0], 11, 1)
    _make_bin(bdir, lids[1], 7, 2)
    outs = {}
    for tag, extra in (('one', []), ('big', ['--gobig'])):
        out = str(tmp_path / ('run_' + tag))
        _cli(['--batch', '8', '--loaders', '0', 'RUN', str(tmp_path / 'run-data'), path, tag, '--outdir', out,
              '--outfile', '{BIN_ID}_class.json'] + extra)
        outs[tag] = [json.load(open(os.path.join(out, lid + '_class.json'))) for lid in lids]
    for a, b in zip(outs['one'], outs['big']):
        assert a['bin_id'] == b['bin_id'] and a['roi_numbers'] == b['roi_numbers']
        assert np.allclose(np.array(a['output_scores']), np.array(b['output_scores']), atol=2e-3)
    assert [len(o['roi_numbers']) for o in outs['big']] == [11, 7]
    # ---- default outfile is .h5: without h5py the run refuses up front instead of failing bin by bin
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(SystemExit, match='h5py'):
            _cli(['--batch', '8', '--loaders', '0', 'RUN', str(tmp_path / 'run-data'), path, 'h5', '--outdir', str(tmp_path / 'run_h5')])


@pytest.mark.parametrize('model,first_key,first_shape,head_key,head_shape',
                         [('squeezenet', 'model.features.0.weight', (64, 3, 3, 3), 'model.classifier.1.weight', (2, 512, 1, 1)),
                          ('alexnet', 'model.features.0.weight', (64, 3, 11, 11), 'model.classifier.6.weight', (2, 4096))])
def test_train_then_run_other_backbone_families(tmp_path, model, first_key, first_shape, head_key, head_shape):
    """the same CLI round trip on backbones without BatchNorm (neuston_models.py:27-33): the checkpoint carries torchvision's
    state_dict layout with the head the reference rebuilds for the dataset's classes, and RUN reloads it"""
    src = str(tmp_path / 'training-data')
    os.makedirs(src)
    _make_dataset(src, per_class=16)
    outdir = str(tmp_path / 'training-output' / 'fam')
    _cli(['--batch', '8', '--loaders', '0', 'TRAIN', src, model, 'fam', '--untrain', '--seed', '2', '--emax', '3', '--emin', '1',
          '--estop', '0', '--outdir', outdir, '--learning-rate', '0.0001', '--results', 'results.json', 'output_scores'])
    ck = torch.load(os.path.join(outdir, 'fam.ptl'), map_location='cpu', weights_only=False)
    sd = ck['state_dict']
    assert list(sd)[0] == first_key and tuple(sd[first_key].shape) == first_shape
    assert tuple(sd[head_key].shape) == head_shape and ck['hyper_parameters']['resize'] == 224
    rows = open(os.path.join(outdir, 'epochs.csv')).read().strip().splitlines()
    assert len(rows) == 4 and all(float(r.split(',')[2]) == float(r.split(',')[2]) for r in rows[1:])       # finite losses
    run_out = str(tmp_path / 'run-output')
    _cli(['--batch', '8', '--loaders', '0', 'RUN', src, os.path.join(outdir, 'fam.ptl'), 'r1', '--type', 'img',
          '--outdir', run_out + '/{RUN_ID}', '--outfile', 'img_results.json'])
    res = json.load(open(os.path.join(run_out, 'r1', 'img_results.json')))
    assert len(res['output_scores']) == 32 and np.allclose(np.sum(res['output_scores'], 1), 1.0, atol=1e-4)
