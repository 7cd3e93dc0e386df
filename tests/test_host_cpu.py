"""Host-side logic that needs no GPU: the tolerant ``.ptl`` reader (f-2), the additive CLI flags, the RUN-mode pre-flight
checks and the C-ABI surface (every symbol of include/ifcbk.h resolves; no compute call is made)."""
import argparse
import os
import re
import sys
import types

import pytest
import torch


def test_reference_style_checkpoint_loads_without_lightning(tmp_path):
    """a [PL] 1.3.8 checkpoint keys ``callbacks`` by callback CLASS objects and stores ``hyper_parameters`` as an
    AttributeDict: reading it must not need pytorch_lightning (SURVEY §5 hazard, reference neuston_net.py:173,443)"""
    from ifcb_classifier_amd.neuston_models import load_checkpoint_file
    pl = types.ModuleType('pytorch_lightning')
    cb = types.ModuleType('pytorch_lightning.callbacks')
    es = types.ModuleType('pytorch_lightning.callbacks.early_stopping')
    ut = types.ModuleType('pytorch_lightning.utilities')
    pa = types.ModuleType('pytorch_lightning.utilities.parsing')

    class EarlyStopping:
        pass
    EarlyStopping.__module__ = es.__name__
    EarlyStopping.__qualname__ = 'EarlyStopping'

    class AttributeDict(dict):
        pass
    AttributeDict.__module__ = pa.__name__
    AttributeDict.__qualname__ = 'AttributeDict'
    es.EarlyStopping, pa.AttributeDict = EarlyStopping, AttributeDict
    mods = {m.__name__: m for m in (pl, cb, es, ut, pa)}
    sys.modules.update(mods)
    try:
        hp = AttributeDict(MODEL='resnet18', classes=['a', 'b'], pretrained=True, model_id='m1', resize=224, img_norm=None, seed=7)
        ck = {'epoch': 3, 'global_step': 40, 'pytorch-lightning_version': '1.3.8',
              'callbacks': {EarlyStopping: {'wait_count': 2, 'best_score': torch.tensor(0.5)}},
              'optimizer_states': [{'state': {0: {'step': 40, 'exp_avg': torch.ones(2)}}, 'param_groups': [{'lr': 1e-3}]}],
              'lr_schedulers': [], 'state_dict': {'model.fc.weight': torch.arange(6.).view(2, 3)}, 'hparams_name': 'hparams',
              'hyper_parameters': hp}
        path = str(tmp_path / 'ref.ptl')
        torch.save(ck, path)
    finally:
        for k in mods:
            del sys.modules[k]
    with pytest.raises(Exception):
        torch.load(path, map_location='cpu', weights_only=False)          # the plain reader needs Lightning
    got = load_checkpoint_file(path)
    assert dict(got['hyper_parameters']) == dict(hp) and got['epoch'] == 3
    assert torch.equal(got['state_dict']['model.fc.weight'], ck['state_dict']['model.fc.weight'])
    (cls, state), = got['callbacks'].items()
    assert cls.__name__ == 'EarlyStopping' and state['wait_count'] == 2
    assert got['optimizer_states'][0]['state'][0]['step'] == 40


def test_additive_cli_flags_default_to_the_reference_behaviour():
    from ifcb_classifier_amd import neuston_net as nn_
    p = nn_.argparse_nn()
    t = p.parse_args(['TRAIN', 'src', 'inception_v3', 'id1'])
    assert (t.optimizer, t.learning_rate, t.momentum, t.precision) == ('Adam', 0.001, 0.0, 'bf16')
    assert t.pretrained is True and t.weights == os.environ.get('IFCBK_PRETRAINED_WEIGHTS')
    t = p.parse_args(['TRAIN', 'src', 'resnet18', 'id1', '--untrain', '--optimizer', 'SGD', '--learning-rate', '0.1', '--momentum', '0.9',
                      '--weights', 'w.pth'])
    assert (t.optimizer, t.learning_rate, t.momentum, t.pretrained, t.weights) == ('SGD', 0.1, 0.9, False, 'w.pth')
    r = p.parse_args(['RUN', 'src', 'm.ptl', 'rid', '--gobig'])
    assert r.gobig is True


def test_train_refuses_pretrained_without_weights(tmp_path, monkeypatch):
    """upstream's default (no --untrain) fine-tunes downloaded ImageNet weights; silently training from random init would
    change the experiment (reference neuston_net.py:340, neuston_models.py:23)"""
    from PIL import Image
    import numpy as np
    from ifcb_classifier_amd import neuston_net as nn_
    monkeypatch.delenv('IFCBK_PRETRAINED_WEIGHTS', raising=False)
    for cls in ('a', 'b'):
        os.makedirs(tmp_path / 'src' / cls)
        for i in range(4):
            Image.fromarray(np.full((20, 30), 40 * i, np.uint8)).save(str(tmp_path / 'src' / cls / ('%s%d.png' % (cls, i))))
    p = nn_.argparse_nn()
    a = p.parse_args(['--loaders', '0', 'TRAIN', str(tmp_path / 'src'), 'resnet18', 'x', '--outdir', str(tmp_path / 'out')])
    a.weights = None
    nn_.argparse_nn_runtimeparams(a)
    with pytest.raises(SystemExit, match='--weights'):
        nn_.do_training(a)


def test_header_and_binding_export_the_same_symbols():
    from ifcb_classifier_amd import _lib
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'ifcbk.h')).read()
    declared = set(re.findall(r'\b(ifcbk_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = _lib.load()                          # every declared symbol resolves in the built library
    assert lib.ifcbk_version().startswith(b'ifcbk')
    assert 'IFCBK_OP_SGD' in hdr and _lib.OP_SGD == 30


def test_the_dynamic_symbol_table_holds_the_c_abi_and_nothing_else():
    """built with -fvisibility=hidden + a linker version script, the library exports exactly the entry points
    include/ifcbk.h declares -- no C++-mangled internal launch helper, no compiler-generated object"""
    import subprocess
    from ifcb_classifier_amd import _lib
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'ifcb_classifier_amd', 'libifcbk.so')
    out = subprocess.run(['nm', '-D', '--defined-only', so], stdout=subprocess.PIPE, text=True, check=True).stdout
    syms = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert syms == set(_lib.EXPORTS), syms ^ set(_lib.EXPORTS)


def test_schema_v1_bins_are_refused_not_silently_unstitched(monkeypatch):
    """upstream classifies old-style bins through pyifcb's InfilledImages (neuston_data.py:446-449); raw halves would be a
    different ROI set, so the dataset refuses unless told otherwise"""
    import numpy as np
    from ifcb_classifier_amd.ifcb_bins import Pid
    from ifcb_classifier_amd.neuston_data import IfcbBinDataset
    monkeypatch.delenv('IFCBK_ALLOW_UNSTITCHED_V1', raising=False)
    b = types.SimpleNamespace(pid=Pid('IFCB1_2010_025_134132'), schema='v1', images={1: np.zeros((3, 4), np.uint8)})
    with pytest.raises(NotImplementedError, match='stitching'):
        IfcbBinDataset(b, 299)
    monkeypatch.setenv('IFCBK_ALLOW_UNSTITCHED_V1', '1')
    assert len(IfcbBinDataset(b, 299)) == 1
    b2 = types.SimpleNamespace(pid=Pid('D20130526T092352_IFCB013'), schema='v2', images={2: np.zeros((3, 4), np.uint8)})
    monkeypatch.delenv('IFCBK_ALLOW_UNSTITCHED_V1')
    assert len(IfcbBinDataset(b2, 299)) == 1


def test_bin_table_is_one_pass_over_the_adc_and_views_into_one_blob(tmp_path):
    """f-3 (reference neuston_data.py:433-454 through pyifcb -- absent, so the column meaning is PARITY UNPINNED): the .adc is
    read once into (targets, offs, hs, ws), the .roi once into one u8 blob; ``images`` are views into that blob -- the form
    ifcbk_roi_preprocess takes (absolute byte offsets), so a bin is ONE upload."""
    import numpy as np
    from ifcb_classifier_amd.ifcb_bins import Bin, DataDirectory
    lid = 'D20130526T092352_IFCB013'
    rng = np.random.default_rng(3)
    blob, lines, off, want = b'', [], 0, {}
    for k in range(9):
        h, w = (int(v) for v in rng.integers(5, 40, 2))
        if k in (2, 7):
            h = w = 0                                   # triggers without an image
        a = rng.integers(0, 256, (h, w)).astype(np.uint8)
        cols = ['0'] * 24
        cols[15], cols[16], cols[17] = str(w), str(h), str(off)
        lines.append(','.join(cols))
        blob += a.tobytes()
        off += h * w
        if h * w:
            want[k + 1] = a
    (tmp_path / (lid + '.adc')).write_text('\n'.join(lines) + '\n')
    (tmp_path / (lid + '.roi')).write_bytes(blob)
    b = Bin(str(tmp_path / lid))
    t = b.table
    assert list(t['targets']) == sorted(want) and t['offs'].dtype == np.int64 and t['hs'].dtype == np.int32
    assert b.blob.nbytes == len(blob)
    for i, n in enumerate(t['targets']):
        img = b.images[int(n)]
        assert img.base is not None and np.shares_memory(img, b.blob)          # a view, not a copy
        assert np.array_equal(img, want[int(n)])
        assert np.array_equal(b.blob[t['offs'][i]:t['offs'][i] + t['hs'][i] * t['ws'][i]].reshape(t['hs'][i], t['ws'][i]), img)
    assert [x.pid.lid for x in DataDirectory(str(tmp_path))] == [lid]
    # an ADC row past the end of the .roi file is an error, not a silent short read
    (tmp_path / (lid + '.roi')).write_bytes(blob[:-10])
    with pytest.raises(ValueError, match='past the end'):
        Bin(str(tmp_path / lid)).table


def test_weights_file_that_needs_unpickling_is_refused_unless_trusted(tmp_path, monkeypatch):
    """--weights reads plain state dicts with weights_only=True; a file that needs arbitrary unpickling is a full
    checkpoint and is loaded only when the caller says it is trusted; other errors propagate (ref neuston_models.py:23-42: upstream
    downloads torchvision's own file here)"""
    import torch
    from ifcb_classifier_amd import neuston_models as nm

    import argparse
    Odd = argparse.Namespace         # a global the safe unpickler does not allow (what Lightning's hyper_parameters hold)
    good, bad = str(tmp_path / 'sd.pth'), str(tmp_path / 'ckpt.pth')
    torch.save({'fc.weight': torch.zeros(2, 3)}, good)
    torch.save({'state_dict': {'fc.weight': torch.zeros(2, 3)}, 'extra': Odd()}, bad)

    class Backbone(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc = torch.nn.Linear(3, 2, bias=False)
    monkeypatch.delenv('IFCBK_TRUST_WEIGHTS', raising=False)
    loaded, skipped = nm.load_pretrained_weights(Backbone(), good)
    assert loaded == ['fc.weight']
    with pytest.raises(RuntimeError, match='IFCBK_TRUST_WEIGHTS'):
        nm.load_pretrained_weights(Backbone(), bad)
    with pytest.raises(FileNotFoundError):
        nm.load_pretrained_weights(Backbone(), str(tmp_path / 'missing.pth'))


def test_bin_table_rejects_negative_sizes_and_offsets(tmp_path):
    from ifcb_classifier_amd.ifcb_bins import Bin
    lid = 'D20130526T092352_IFCB013'
    cols = ['0'] * 24
    cols[15], cols[16], cols[17] = '4', '5', '-20'
    (tmp_path / (lid + '.adc')).write_text(','.join(cols) + '\n')
    (tmp_path / (lid + '.roi')).write_bytes(bytes(64))
    with pytest.raises(ValueError, match='negative'):
        Bin(str(tmp_path / lid)).table
