"""Result writers of the TRAIN / RUN drivers -- host-side twin of ``/root/reference/neuston_callbacks.py``
(validation results :20-156, run results "v3" :160-272).  Plain host code on small arrays ([N, NC] scores);
formats: .json, .mat (scipy) and .h5 when h5py is importable (it is not in the build image: a clear
RuntimeError is raised instead of writing a different format silently)."""
import json
import os

import numpy as np


def _h5():
    try:
        import h5py
        return h5py
    except ImportError:
        raise RuntimeError('h5py is not installed in this environment: use a .json or .mat --outfile/--results name')


class SaveValidationResults:
    """neuston_callbacks.py:20-156; called by the trainer after every validation epoch."""

    def __init__(self, outdir, outfile, series, best_only=True):
        self.outdir, self.outfile, self.series, self.best_only = outdir, outfile, series, best_only

    def on_validation_end(self, log, model, train_dataset, val_dataset):
        from sklearn import metrics
        if not (log['best'] or not self.best_only):
            return None
        labels = model.hparams.classes
        idxs = list(range(len(labels)))
        val_counts = val_dataset.count_perclass
        train_counts = train_dataset.count_perclass
        counts = [v + t for v, t in zip(val_counts, train_counts)]
        base = lambda p: os.path.splitext(os.path.basename(p))[0]
        scores = log['outputs']
        out_cls = np.argmax(scores, axis=1)
        in_cls = log['input_classes']
        paths = log['input_srcs']
        assert scores.shape[0] == len(in_cls), 'wrong number inputs-to-outputs'
        assert scores.shape[1] == len(labels), 'wrong number of class labels'
        stats = {}
        for mode in ('weighted', 'macro', None):
            for stat in ('f1', 'recall', 'precision'):
                fn = getattr(metrics, stat + '_score')
                stats['{}_{}'.format(stat, mode if mode else 'perclass')] = fn(in_cls, out_cls, labels=idxs, average=mode,
                                                                               zero_division=0)
        by = {'count': sorted(idxs, key=lambda i: counts[i], reverse=True)}
        for stat in ('f1', 'recall', 'precision'):
            by[stat] = sorted(idxs, key=lambda i: stats[stat + '_perclass'][i], reverse=True)
        cm = metrics.confusion_matrix(in_cls, out_cls, labels=idxs, normalize=None)
        res = dict(model_id=model.hparams.model_id, timestamp=model.hparams.cmd_timestamp, class_labels=labels,
                   input_classes=in_cls, output_classes=out_cls)
        opt = dict(image_fullpaths=paths, image_basenames=[base(p) for p in paths],
                   training_image_fullpaths=list(train_dataset.images),
                   training_image_basenames=[base(p) for p in train_dataset.images],
                   training_classes=list(train_dataset.targets), output_winscores=np.max(scores, axis=1),
                   output_scores=scores, confusion_matrix=cm, counts_perclass=counts, val_counts_perclass=val_counts)
        for k, v in opt.items():
            if k in self.series:
                res[k] = v
        if 'train_counts_perclass' in self.series:          # upstream writes the VAL counts under this request (:98)
            res['val_counts_perclass'] = val_counts
        for k, v in stats.items():
            if k in self.series:
                res[k] = v
        for k, v in by.items():
            if 'classes_by_' + k in self.series:
                res['classes_by_' + k] = v
        outfile = os.path.join(self.outdir, self.outfile).format(epoch=log['epoch'])
        os.makedirs(os.path.dirname(outfile) or '.', exist_ok=True)
        self.save_validation_results(outfile, res)
        return outfile

    def save_validation_results(self, outfile, results):
        if outfile.endswith('.json'):
            out = {k: (v.tolist() if isinstance(v, np.ndarray) else (float(v) if isinstance(v, np.floating) else v))
                   for k, v in results.items()}
            with open(outfile, 'w') as f:
                json.dump(out, f)
        if outfile.endswith('.mat'):
            from scipy.io import savemat
            idx = ['input_classes', 'output_classes', 'training_classes'] + ['classes_by_' + s for s in
                                                                             'f1 recall precision count'.split()]
            strs = ['class_labels', 'image_fullpaths', 'image_basenames', 'training_image_fullpaths',
                    'training_image_basenames']
            out = {}
            for k, v in results.items():
                # upstream's order of tests (:128-134): ndarrays go to float32 FIRST, so input_classes / output_classes (numpy
                # arrays) are stored as zero-based float32 and only the list-valued index series get the 1-based uint32 form
                if isinstance(v, np.ndarray):
                    out[k] = v.astype('f4')
                elif isinstance(v, (np.floating, float)):       # (sklearn >= 1.x returns python floats where 0.24 gave np.float64)
                    out[k] = np.asarray(v).astype('f4')
                elif k in strs:
                    out[k] = np.asarray(v, dtype='object')
                elif k in idx:
                    out[k] = np.asarray(v).astype('u4') + 1            # matlab indices are 1-based
                else:
                    out[k] = v
            savemat(outfile, out, do_compression=True)
        if outfile.endswith('.h5'):
            h5 = _h5()
            attrib = ['model_id', 'timestamp'] + 'f1_weighted recall_weighted precision_weighted f1_macro recall_macro precision_macro'.split()
            ints = ['input_classes', 'output_classes', 'training_classes', 'counts_perclass', 'val_counts_perclass',
                    'train_counts_perclass'] + ['classes_by_' + s for s in 'f1 recall precision count'.split()]
            strs = ['class_labels', 'image_fullpaths', 'image_basenames', 'training_image_fullpaths',
                    'training_image_basenames']
            with h5.File(outfile, 'w') as f:
                meta = f.create_dataset('metadata', data=h5.Empty('f'))
                for k, v in results.items():
                    if k in attrib:
                        meta.attrs[k] = v
                    elif k in strs:
                        f.create_dataset(k, data=np.array(v, dtype='S'), compression='gzip', dtype=h5.string_dtype())
                    elif k in ints:
                        f.create_dataset(k, data=v, compression='gzip', dtype='int16')
                    elif isinstance(v, np.ndarray):
                        f.create_dataset(k, data=v, compression='gzip', dtype='float16')
                    else:
                        raise UserWarning('hdf results: WE MISSED THIS ONE: {}'.format(k))


def save_run_results(input_images, output_scores, class_labels, timestamp, outdir, outfile, model_id=None,
                     input_obj=None):
    """neuston_callbacks.py:160-272: the "v3" class file.  ``input_obj`` is a bin pid object (``.pid``,
    ``.namespace``, ``.year``, ``.yearday``; ROI ids expose ``.target``) or the image-source path string."""
    output_classes = np.argmax(output_scores, axis=1)
    assert output_scores.shape[0] == len(output_classes), 'wrong number inputs-to-outputs'
    assert output_scores.shape[1] == len(class_labels), 'wrong number of class labels'
    results = dict(version='v3', model_id=model_id, timestamp=timestamp, class_labels=class_labels,
                   input_images=input_images, output_classes=output_classes, output_scores=output_scores)
    outfile = os.path.join(outdir, outfile)
    if hasattr(input_obj, 'pid') and hasattr(input_obj, 'yearday'):
        results['bin_id'] = input_obj.pid
        results['roi_numbers'] = [getattr(img, 'target', img) for img in input_images]
        outfile = outfile.format(BIN_ID=input_obj.pid, INPUT_SUBDIRS=input_obj.namespace, BIN_YEAR=input_obj.year,
                                 BIN_DATE=input_obj.yearday).replace(2 * os.sep, os.sep)
        os.makedirs(os.path.dirname(outfile) or '.', exist_ok=True)
        _save_run_results(outfile, results)
        return [outfile]
    if '{INPUT_SUBDIRS}' in outfile:
        groups = {}
        src = input_obj if (input_obj and os.path.isdir(input_obj)) else ''
        for path, cls, sc in zip(input_images, output_classes, output_scores):
            parent = os.path.dirname(path.replace(src, ''))
            g = groups.setdefault(parent, {k: (v if k not in ('input_images', 'output_classes', 'output_scores') else [])
                                           for k, v in results.items()})
            g['input_images'].append(os.path.basename(path))
            g['output_classes'].append(cls)
            g['output_scores'].append(sc)
        written = []
        for parent, sub in groups.items():
            sub_out = outfile.format(INPUT_SUBDIRS=parent)
            os.makedirs(os.path.dirname(sub_out) or '.', exist_ok=True)
            sub['output_classes'] = np.asarray(sub['output_classes'], dtype=output_classes.dtype)
            sub['output_scores'] = np.asarray(sub['output_scores'], dtype=output_scores.dtype)
            _save_run_results(sub_out, sub)
            written.append(sub_out)
        return written
    os.makedirs(os.path.dirname(outfile) or '.', exist_ok=True)
    _save_run_results(outfile, results)
    return [outfile]


def _save_run_results(outfile, results):
    ext = os.path.splitext(outfile)[-1]
    assert ext in ['.json', '.mat', '.h5'], 'output fileformat "{}" not valid'.format(ext)
    if ext == '.json':
        out = dict(version=results['version'], model_id=results['model_id'], timestamp=results['timestamp'],
                   class_labels=results['class_labels'], output_scores=np.asarray(results['output_scores']).tolist(),
                   output_classes=np.asarray(results['output_classes']).tolist())
        if 'bin_id' in results:
            out['bin_id'] = results['bin_id']
            out['roi_numbers'] = [int(r) if isinstance(r, (int, np.integer)) else r for r in results['roi_numbers']]
        else:
            out['input_images'] = [str(p) for p in results['input_images']]
        with open(outfile, 'w') as f:
            json.dump(out, f)
    elif ext == '.mat':
        from scipy.io import savemat
        out = dict(output_classes=np.asarray(results['output_classes']).astype('u4') + 1, version=results['version'],
                   model_id=results['model_id'], timestamp=results['timestamp'],
                   output_scores=np.asarray(results['output_scores']).astype('f4'),
                   class_labels=np.asarray(results['class_labels'], dtype='object'))
        if 'bin_id' in results:
            out['bin_id'] = results['bin_id']
            out['roi_numbers'] = results['roi_numbers']
        else:
            out['input_images'] = np.asarray([str(p) for p in results['input_images']], dtype='object')
        savemat(outfile, out, do_compression=True)
    else:
        h5 = _h5()
        with h5.File(outfile, 'w') as f:
            meta = f.create_dataset('metadata', data=h5.Empty('f'))
            meta.attrs['version'] = results['version']
            meta.attrs['model_id'] = results['model_id']
            meta.attrs['timestamp'] = results['timestamp']
            f.create_dataset('output_classes', data=results['output_classes'], compression='gzip', dtype='float16')
            f.create_dataset('output_scores', data=results['output_scores'], compression='gzip', dtype='float16')
            f.create_dataset('class_labels', data=np.array(results['class_labels'], dtype='S'), compression='gzip',
                             dtype=h5.string_dtype())
            if results.get('bin_id'):                      # upstream indexes results['bin_id'] and KeyErrors in img mode
                meta.attrs['bin_id'] = results['bin_id']
                f.create_dataset('roi_numbers', data=results['roi_numbers'], compression='gzip', dtype='uint16')
            else:
                f.create_dataset('input_images', data=np.array([str(p) for p in results['input_images']], dtype='S'),
                                 compression='gzip', dtype=h5.string_dtype())


class SaveTestResults:
    """neuston_callbacks.py:275-296"""

    def __init__(self, outdir, outfile, timestamp):
        self.outdir, self.outfile, self.timestamp = outdir, outfile, timestamp

    def on_test_end(self, run_results, model):
        written = []
        for rr in (run_results if isinstance(run_results, list) else [run_results]):
            written += save_run_results(rr.inputs, rr.outputs, model.hparams.classes, self.timestamp, self.outdir,
                                        self.outfile, model.hparams.model_id, rr.input_obj)
        return written
