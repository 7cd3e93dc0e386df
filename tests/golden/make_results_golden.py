"""Generates tests/golden/results_golden.json by running the REFERENCE's own result writers
(/root/reference/neuston_callbacks.py: save_run_results :160-272, SaveValidationResults :20-156) on fixed inputs, with stub
modules for its missing imports (pytorch_lightning, ifcb, neuston_data).  Needs h5py: run it with the side interpreter

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/make_results_golden.py

Only descriptions of the files (names, dtypes, shapes, values, attributes) are stored -- no reference source."""
import json
import os
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import results_describe as RD          # noqa: E402


class Pid:
    """the fields of pyifcb's Pid that the reference's writers touch"""

    def __init__(self, pid):
        self.pid = pid
        self.target = int(pid.rsplit('_', 1)[1]) if pid.count('_') >= 2 else None
        self.namespace, self.year, self.yearday = 'D2013/D20130526/', '2013', '20130526'


def stub_modules():
    ptl = types.ModuleType('pytorch_lightning')
    cbs = types.ModuleType('pytorch_lightning.callbacks')
    base = types.ModuleType('pytorch_lightning.callbacks.base')

    class Callback:
        pass
    base.Callback = Callback
    cbs.base = base
    ptl.callbacks = cbs
    ifcb = types.ModuleType('ifcb')
    ifcb.Pid = Pid
    nd = types.ModuleType('neuston_data')
    nd.IfcbBinDataset = object
    for name, m in (('pytorch_lightning', ptl), ('pytorch_lightning.callbacks', cbs), ('pytorch_lightning.callbacks.base', base),
                    ('ifcb', ifcb), ('neuston_data', nd)):
        sys.modules[name] = m


def main():
    stub_modules()
    sys.path.insert(0, '/root/reference')
    import neuston_callbacks as ref                                     # the reference itself
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        # ---- run results, bin mode (all three formats) and image mode (.h5 raises KeyError('bin_id') upstream: recorded)
        lid = 'D20130526T092352_IFCB013'
        bin_pid = Pid(lid)
        rois = ['%s_%05d' % (lid, n) for n in (1, 2, 4, 7, 300)]
        sc = RD.scores(5, 5)
        for ext in ('json', 'mat', 'h5'):
            ref.save_run_results(rois, sc, RD.CLASSES, RD.TIMESTAMP, tmp, 'D{BIN_YEAR}/D{BIN_DATE}/{BIN_ID}_class.' + ext,
                                 RD.MODEL_ID, bin_pid)
            out['run_bin_' + ext] = RD.describe(os.path.join(tmp, 'D2013', 'D20130526', lid + '_class.' + ext))
        imgs = [os.path.join(tmp, 'src', 'a', 'x1.png'), os.path.join(tmp, 'src', 'a', 'x2.png'), os.path.join(tmp, 'src', 'b', 'y1.png')]
        sc3 = RD.scores(3, 6)
        for ext in ('json', 'mat'):
            ref.save_run_results([os.path.relpath(p, tmp) for p in imgs], sc3, RD.CLASSES, RD.TIMESTAMP, tmp, 'out/img_results.' + ext,
                                 RD.MODEL_ID, 'src')
            out['run_img_' + ext] = RD.describe(os.path.join(tmp, 'out', 'img_results.' + ext))
        try:
            ref.save_run_results(['a/x1.png'], sc3[:1], RD.CLASSES, RD.TIMESTAMP, tmp, 'out/img_results.h5', RD.MODEL_ID, 'src')
            out['run_img_h5_error'] = None
        except Exception as e:
            out['run_img_h5_error'] = type(e).__name__
        # ---- validation results through the callback (fake trainer / module objects carrying what it reads)
        vi = RD.val_inputs()

        class DS:
            def __init__(self, images, targets, counts):
                self.images, self.targets, self.count_perclass = images, targets, counts

        class Loader:
            def __init__(self, ds):
                self.dataset = ds

        class Module:
            current_epoch = vi['epoch']
            hparams = types.SimpleNamespace(classes=RD.CLASSES, model_id=RD.MODEL_ID, cmd_timestamp=RD.TIMESTAMP)

            def val_dataloader(self):
                return Loader(DS(vi['input_srcs'], list(vi['input_classes']), vi['val_counts']))

            def train_dataloader(self):
                return Loader(DS(vi['train_images'], vi['train_targets'], vi['train_counts']))

        for ext in ('json', 'mat', 'h5'):
            cb = ref.SaveValidationResults(tmp, 'val/results_{epoch}.' + ext, list(RD.VAL_SERIES))
            trainer = types.SimpleNamespace(callback_metrics=dict(outputs=vi['outputs'].copy(), input_classes=vi['input_classes'].copy(),
                                                                  input_srcs=list(vi['input_srcs']), best=True, epoch=vi['epoch']))
            cb.on_validation_end(trainer, Module())
            out['val_' + ext] = RD.describe(os.path.join(tmp, 'val', 'results_%d.%s' % (vi['epoch'], ext)))
    with open(os.path.join(HERE, 'results_golden.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print('wrote results_golden.json:', sorted(out))


if __name__ == '__main__':
    main()
