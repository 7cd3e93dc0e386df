#!/usr/bin/env python
"""``neuston_net.py TRAIN|RUN`` on the MI355X path -- same command line as ``/root/reference/neuston_net.py``
(:311-413: global ``--batch/--loaders`` before the sub-command, every TRAIN / RUN flag with its default), same
outputs ({model_id}.ptl, epochs.csv, args.yml, training/validation image lists, results files, per-bin class
files), with Lightning's Trainer replaced by the small explicit loop below (fit: :101-115, test: :192-308).

Data parallel: launch one process per GPU with ``python -m torch.distributed.run --nproc-per-node N -m
ifcb_classifier_amd.neuston_net ... TRAIN ...``; ``--batch`` stays per GPU (as under the reference's
ddp_spawn), gradients are averaged over RCCL, BatchNorm statistics stay per rank, rank 0 writes the files.
"""
import argparse
import csv
import datetime as dt
import os
import random
from shutil import copyfile

import numpy as np
import torch
from torch.utils.data import DataLoader

from .neuston_callbacks import SaveValidationResults, SaveTestResults
from .neuston_data import (get_trainval_datasets, IfcbBinDataset, ImageDataset, IMG_EXTENSIONS, collate_rois,
                           rois_to_device)
from .neuston_models import NeustonModel, load_checkpoint_file, load_pretrained_weights


def seed_everything(seed=None):
    """[PL] seed_everything: None -> draw a seed, record it (neuston_net.py:62)."""
    if seed is None:
        seed = random.SystemRandom().randint(0, 2 ** 32 - 1)
    seed = int(seed)
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)
    os.environ['PL_GLOBAL_SEED'] = str(seed)
    return seed


def _dist():
    import torch.distributed as dist
    if int(os.environ.get('WORLD_SIZE', 1)) > 1:
        if not dist.is_initialized():
            local = int(os.environ.get('LOCAL_RANK', 0))
            if torch.cuda.is_available():
                torch.cuda.set_device(local)
                dist.init_process_group('nccl', device_id=torch.device('cuda', local))
            else:
                dist.init_process_group('gloo')
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


class ShardedLoader:
    """per-rank view of a dataset order (what [PL]'s auto DistributedSampler does): a seeded permutation per epoch
    (train) or the natural order (val), padded by wrap-around to a multiple of world, strided by rank."""

    def __init__(self, dataset, batch_size, shuffle, num_workers, rank, world, seed):
        self.dataset, self.bs, self.shuffle, self.nw = dataset, batch_size, shuffle, num_workers
        self.rank, self.world, self.seed, self.epoch = rank, world, seed, 0

    def set_epoch(self, e):
        self.epoch = e

    def indices(self):
        n = len(self.dataset)
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            idx = torch.randperm(n, generator=g).tolist()
        else:
            idx = list(range(n))
        if self.world > 1:
            total = (n + self.world - 1) // self.world * self.world
            idx = (idx + idx[:total - n])[self.rank:total:self.world]
        return idx

    def __iter__(self):
        sub = torch.utils.data.Subset(self.dataset, self.indices())
        return iter(DataLoader(sub, batch_size=self.bs, shuffle=False, pin_memory=True, num_workers=self.nw,
                               collate_fn=collate_rois))

    def __len__(self):
        return (len(self.indices()) + self.bs - 1) // self.bs


class Trainer:
    """fit/test loops with [PL] 1.3.8's observable semantics for this driver: num_sanity_val_steps=0; per epoch
    train -> validate -> callbacks; ModelCheckpoint(monitor=val_loss, top-1); EarlyStopping(val_loss, patience)
    honoured only after min_epochs; CSV log of scalar metrics."""

    def __init__(self, max_epochs, min_epochs, estop, outdir, callbacks=()):
        self.max_epochs, self.min_epochs, self.estop = max_epochs, min_epochs, estop
        self.outdir, self.callbacks = outdir, list(callbacks)
        self.metrics = []
        self.best_model_path = None
        self.dist, self.rank, self.world = _dist()

    @staticmethod
    def _lookahead(loader):
        """(batch, next batch or None) pairs: the next batch is staged on the GPU's side stream while the current one runs"""
        it = iter(loader)
        cur = next(it, None)
        while cur is not None:
            nxt = next(it, None)
            yield cur, nxt
            cur = nxt

    def _allreduce(self, t):
        # the bucket exchange: one all_reduce, or reduce-scatter + all-gather (IFCBK_DP_EXCHANGE=allreduce|rsag, dp.make_exchange)
        if getattr(self, '_exchange', None) is None:
            from .dp import make_exchange
            self._exchange, self.exchange_mode = make_exchange(self.dist)
        return self._exchange(t)

    def fit(self, model, train_loader, val_loader):
        dev = model.model.engine.dev
        eng = model.model.engine
        if self.world > 1:
            self.dist.broadcast(eng.P, 0)
            self.dist.broadcast(eng.RB, 0)
            eng.params_changed()
        chk = os.path.join(self.outdir, 'chkpts')
        best, wait, global_step = np.inf, 0, 0
        for epoch in range(self.max_epochs):
            model.current_epoch = epoch
            train_loader.set_epoch(epoch)
            ttf, vtf = train_loader.dataset.transforms, val_loader.dataset.transforms
            # one batch of look-ahead: upload + on-GPU preprocessing of batch k+1 run on a side stream beside step k
            n_next = None
            for (rois, targets, paths), nxt in self._lookahead(train_loader):
                n = model.stage_batch(rois, ttf, targets) if n_next is None else n_next
                model.use_staged()
                n_next = model.stage_batch(nxt[0], ttf, nxt[1]) if nxt is not None else None
                model.fit_current(n, self.world, self._allreduce if self.world > 1 else None)
                global_step += 1
            model.agg_train_loss = model.epoch_train_loss()
            steps = []
            n_next = None
            for (rois, targets, paths), nxt in self._lookahead(val_loader):
                n = model.stage_batch(rois, vtf, targets) if n_next is None else n_next
                model.use_staged()
                n_next = model.stage_batch(nxt[0], vtf, nxt[1]) if nxt is not None else None
                probs, loss = model.eval_current(n, with_loss=True)
                tg = targets.to(dev, non_blocking=True)
                steps.append(dict(val_batch_loss=loss, val_outputs=probs, val_input_classes=tg, val_input_srcs=list(paths)))
            steps = self._gather_val(steps, len(val_loader.dataset))
            stop = False
            if self.rank == 0:
                model.validation_epoch_end(steps)
                log = model.logged
                self.metrics.append({k: (float(v) if not isinstance(v, bool) else v) for k, v in log.items()
                                     if k not in ('input_classes', 'output_classes', 'input_srcs', 'outputs')})
                for cb in self.callbacks:
                    cb.on_validation_end(log, model, train_loader.dataset, val_loader.dataset)
                if log['val_loss'] < best:
                    best, wait = log['val_loss'], 0
                    if self.world > 1:
                        pass                                  # rank-0 BN buffers win (DDP broadcast_buffers)
                    os.makedirs(chk, exist_ok=True)
                    path = os.path.join(chk, 'epoch={}-step={}.ckpt'.format(epoch, global_step - 1))
                    torch.save(model.checkpoint_dict(epoch, global_step), path)
                    if self.best_model_path and os.path.exists(self.best_model_path) and self.best_model_path != path:
                        os.remove(self.best_model_path)
                    self.best_model_path = path
                else:
                    wait += 1
                stop = bool(self.estop and wait >= self.estop and epoch + 1 >= self.min_epochs)
            if self.world > 1:
                flag = torch.tensor([1 if stop else 0], device=dev)
                self.dist.broadcast(flag, 0)
                stop = bool(flag.item())
            if stop:
                break

    def _gather_val(self, steps, n_total):
        """rank 0 receives every rank's validation outputs (the reference's epoch-end aggregation sees only the
        local shard under ddp: SURVEY.md §5); wrap-around padding is dropped."""
        if self.world == 1:
            return steps
        payload = [dict(val_batch_loss=s['val_batch_loss'].item(), val_outputs=s['val_outputs'].cpu(),
                        val_input_classes=s['val_input_classes'].cpu(), val_input_srcs=s['val_input_srcs']) for s in steps]
        gathered = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object(payload, gathered, dst=0)
        if self.rank != 0:
            return []
        out, seen = [], set()
        for part in gathered:
            for s in part:
                keep = [i for i, p in enumerate(s['val_input_srcs']) if p not in seen]
                seen.update(s['val_input_srcs'])
                if not keep:
                    continue
                out.append(dict(val_batch_loss=torch.tensor(s['val_batch_loss']), val_outputs=s['val_outputs'][keep],
                                val_input_classes=s['val_input_classes'][keep],
                                val_input_srcs=[s['val_input_srcs'][i] for i in keep]))
        return out

    def test(self, model, loader, input_obj, transform, callbacks):
        dev = model.model.engine.dev
        steps = []
        n_next = None
        for (rois, ids), nxt in self._lookahead(loader):
            n = model.stage_batch(rois, transform) if n_next is None else n_next
            model.use_staged()
            n_next = model.stage_batch(nxt[0], transform) if nxt is not None else None
            probs, _ = model.eval_current(n)
            steps.append(dict(test_outputs=probs, test_srcs=list(ids)))
        rr = model.test_epoch_end(steps, input_obj)
        rr.type = 'Bin' if hasattr(input_obj, 'yearday') else 'ImgDir'
        written = []
        for cb in callbacks:
            written += cb.on_test_end(rr, model)
        return rr, written

    def test_resident(self, model, bin_fileset, ds, input_obj, callbacks, batch_size):
        """RUN on one raw bin the way SURVEY 8 f-3 means it (reference data path: neuston_data.py:433-467, one PIL image per
        ROI through DataLoader workers): the .adc was read ONCE into an offset / size table, the .roi blob is uploaded ONCE, and
        every batch of the bin is preprocessed on the GPU straight from that resident blob (ifcbk_roi_preprocess takes absolute
        byte offsets) -- no per-ROI slicing, no per-batch concatenation or upload.  Same scores as ``test`` over a DataLoader of
        the bin, bit for bit."""
        t = bin_fileset.table
        res = model.upload_bin(bin_fileset.blob, t['offs'], t['hs'], t['ws'])
        n_total = len(t['targets'])
        bounds = [(i, min(i + batch_size, n_total)) for i in range(0, n_total, batch_size)]
        steps, n_next = [], None
        for k, (i0, i1) in enumerate(bounds):
            n = model.stage_resident(res, i0, i1, ds.transform) if n_next is None else n_next
            model.use_staged()
            n_next = model.stage_resident(res, bounds[k + 1][0], bounds[k + 1][1], ds.transform) if k + 1 < len(bounds) else None
            probs, _ = model.eval_current(n)
            steps.append(dict(test_outputs=probs, test_srcs=list(ds.pids[i0:i1])))
        rr = model.test_epoch_end(steps, input_obj)
        rr.type = 'Bin'
        written = []
        for cb in callbacks:
            written += cb.on_test_end(rr, model)
        return rr, written

    def test_many(self, model, bins, batch_size, num_workers, callbacks):
        """--gobig: classify the ROIs of all ``bins`` [(pid, IfcbBinDataset)] as one stream of full batches, then write one
        result set per bin (the per-bin files are the same as without --gobig: an image's scores do not depend on its batch)."""
        dev = model.model.engine.dev
        cat = torch.utils.data.ConcatDataset([ds for _, ds in bins])
        loader = DataLoader(cat, batch_size=batch_size, pin_memory=True, num_workers=num_workers, collate_fn=collate_rois)
        probs, ids = [], []
        tf, n_next = bins[0][1].transform, None
        for (rois, pids), nxt in self._lookahead(loader):
            n = model.stage_batch(rois, tf) if n_next is None else n_next
            model.use_staged()
            n_next = model.stage_batch(nxt[0], tf) if nxt is not None else None
            p, _ = model.eval_current(n)
            probs.append(p)
            ids.extend(pids)
        probs = torch.cat(probs, 0).detach().cpu().numpy()
        written, lo = [], 0
        for bin_obj, ds in bins:
            hi = lo + len(ds)
            rr = model.RunResults(inputs=ids[lo:hi], outputs=probs[lo:hi], input_obj=bin_obj)
            rr.type = 'Bin'
            model.logged = dict(RunResults=[rr])
            for cb in callbacks:
                written += cb.on_test_end(rr, model)
            lo = hi
        return written

    def write_metrics_csv(self, path):
        keys = []
        for m in self.metrics:
            for k in m:
                if k not in keys:
                    keys.append(k)
        with open(path, 'w', newline='') as f:
            w = csv.DictWriter(f, fieldnames=keys)
            w.writeheader()
            w.writerows(self.metrics)


def main(args):
    if args.cmd_mode == 'TRAIN':
        do_training(args)
    else:
        do_run(args)
    print('\nDONE!')


def do_training(args):
    date_str = args.cmd_timestamp.split('T')[0]
    args.model_id = args.model_id.format(TRAIN_DATE=date_str, TRAIN_ID=args.TRAIN_ID)
    os.makedirs(args.outdir, exist_ok=True)
    if not args.result_files:
        args.result_files = ['results.mat training_image_basenames training_classes image_basenames input_classes '
                             'output_scores confusion_matrix counts_perclass f1_perclass f1_weighted f1_macro'.split()]
    callbacks = [SaveValidationResults(outdir=args.outdir, outfile=rf[0], series=rf[1:]) for rf in args.result_files]
    args.seed = seed_everything(args.seed or None)

    training_dataset, validation_dataset = get_trainval_datasets(args)
    assert training_dataset.classes == validation_dataset.classes
    args.classes = training_dataset.classes
    dist, rank, world = _dist()
    if rank == 0:
        with open(os.path.join(args.outdir, 'training_images.list'), 'w') as f:
            f.write('\n'.join(sorted(training_dataset.images)))
        with open(os.path.join(args.outdir, 'validation_images.list'), 'w') as f:
            f.write('\n'.join(sorted(validation_dataset.images)))
    print('Loading Training Dataloader...')
    training_loader = ShardedLoader(training_dataset, args.batch_size, True, args.loaders, rank, world, args.seed)
    print('Loading Validation Dataloader...')
    validation_loader = ShardedLoader(validation_dataset, args.batch_size, False, args.loaders, rank, world, args.seed)

    if args.pretrained and not getattr(args, 'weights', None):
        # upstream: pretrained=True downloads torchvision's ImageNet weights (neuston_models.py:23-42).  Training from random
        # initialisation instead would silently change the experiment, so refuse.
        raise SystemExit('TRAIN without --untrain fine-tunes ImageNet weights upstream (torchvision download); this path has no '
                         'network and no torchvision: pass --weights <torchvision state_dict .pth> (or set '
                         'IFCBK_PRETRAINED_WEIGHTS), or add --untrain to train from random initialisation')
    trainer = Trainer(args.emax, args.emin, args.estop, args.outdir, callbacks)
    hp = argparse.Namespace(**{k: v for k, v in vars(args).items()})
    classifier = NeustonModel(hp, device=int(os.environ.get('LOCAL_RANK', 0)), max_batch=args.batch_size)
    if args.pretrained:
        loaded, skipped = load_pretrained_weights(classifier.model, args.weights)
        print('Loaded {} pretrained tensors from {}; freshly initialised: {}'.format(
            len(loaded), args.weights, [k for k in skipped if not k.endswith('num_batches_tracked')]))
    eng = classifier.model.engine
    if eng.window_batch < args.batch_size:
        # --batch is per GPU as upstream (neuston_net.py:102); BatchNorm statistics are per step, so a step cannot be chunked
        raise ValueError('--batch %d: one launch addresses each tensor through a 2 GiB buffer descriptor, which holds %d images '
                         'of %s; use --batch <= %d per GPU (and more GPUs for a larger global batch)'
                         % (args.batch_size, eng.window_batch, args.MODEL, eng.window_batch))
    trainer.fit(classifier, training_loader, validation_loader)
    if rank != 0:
        return
    output_path = os.path.join(args.outdir, args.model_id + '.ptl')
    copyfile(trainer.best_model_path, output_path)
    if args.epochs_log:
        trainer.write_metrics_csv(os.path.join(args.outdir, args.epochs_log))
    if args.args_log:
        import yaml
        with open(os.path.join(args.outdir, args.args_log), 'w') as f:
            yaml.safe_dump({k: (v if isinstance(v, (int, float, str, bool, list, type(None))) else str(v))
                            for k, v in vars(args).items()}, f)
    if args.onnx:
        raise NotImplementedError('--onnx export is outside the MI355X hot path (SURVEY.md §2 row 8)')


class _ImgSource:
    def __init__(self, src):
        self.src = src


def do_run(args):
    if args.filter:
        if args.filter[0] not in ['IN', 'OUT']:
            raise argparse.ArgumentTypeError('IN|OUT must be either "IN" or "OUT"')
        if len(args.filter) < 2:
            raise argparse.ArgumentTypeError('Must be at least one KEYWORD')
    classifier = NeustonModel.load_from_checkpoint(args.MODEL, device=int(os.environ.get('LOCAL_RANK', 0)),
                                                   max_batch=args.batch_size, inference=True)
    if args.batch_size > classifier.model.engine.max_batch:
        args.batch_size = classifier.model.engine.max_batch          # per-image results: a smaller program batch changes nothing
    seed_everything(classifier.hparams.seed)
    # (a RUN batch beyond the 2 GiB buffer-descriptor window needs nothing here: results are per image, and the library cuts the
    #  convolutions of such a batch into launches over image groups)
    if os.path.isdir(args.SRC) and not args.SRC.endswith(os.sep):
        args.SRC = args.SRC + os.sep
    if not args.outfile:
        args.outfile = ['D{BIN_YEAR}/D{BIN_DATE}/{BIN_ID}_class.h5'] if args.src_type == 'bin' else ['img_results.json']
    if any(o.endswith('.h5') for o in args.outfile):
        try:
            import h5py  # noqa: F401
        except ImportError:
            raise SystemExit('RUN: outfile(s) {} need h5py, which is not installed in this environment (the default class file is '
                             '.h5, neuston_net.py:181): install h5py or pass --outfile with a .mat / .json name'.format(
                                 [o for o in args.outfile if o.endswith('.h5')]))
    callbacks = [SaveTestResults(outdir=args.outdir, outfile=o, timestamp=args.cmd_timestamp) for o in args.outfile]
    trainer = Trainer(0, 0, 0, args.outdir)
    filter_mode, filter_keywords = None, []
    if args.filter:
        filter_mode = args.filter[0]
        for kw in args.filter[1:]:
            if os.path.isfile(kw):
                with open(kw) as f:
                    filter_keywords.extend(f.read().splitlines())
            else:
                filter_keywords.append(kw)

    if args.src_type == 'bin':
        from .ifcb_bins import DataDirectory
        if os.path.isdir(args.SRC):
            dd = DataDirectory(args.SRC, whitelist=filter_keywords if filter_mode == 'IN' else None,
                               blacklist=filter_keywords if filter_mode == 'OUT' else None)
        elif os.path.isfile(args.SRC) and args.SRC.endswith('.txt'):
            with open(args.SRC) as f:
                bins = f.read().splitlines()
            dd = DataDirectory(os.path.commonpath(bins), whitelist=bins)
        else:
            dd = DataDirectory(os.path.dirname(args.SRC), whitelist=[os.path.basename(args.SRC)])
        error_bins = []
        big = []          # --gobig: (bin pid, dataset) of every bin, classified together afterwards
        n_bins = 0
        dist, rank, world = _dist()
        if args.gobig:
            print('Loading Bins', end=' ')
        for i, bin_fileset in enumerate(dd):
            if world > 1 and i % world != rank:        # RUN shards bins over ranks; no collective (replicas only)
                continue
            bin_fileset.pid.namespace = os.path.dirname(bin_fileset.basepath.replace(args.SRC, '')) + os.sep
            bin_obj = bin_fileset.pid
            if args.filter:
                if filter_mode == 'IN' and not any(k in str(bin_obj) for k in filter_keywords):
                    continue
                if filter_mode == 'OUT' and any(k in str(bin_obj) for k in filter_keywords):
                    continue
            if not args.clobber:
                fmt = dict(BIN_ID=bin_obj.pid, BIN_YEAR=bin_obj.year, BIN_DATE=bin_obj.yearday, INPUT_SUBDIRS=bin_obj.namespace)
                outs = [os.path.join(args.outdir, o).format(**fmt).replace(2 * os.sep, os.sep) for o in args.outfile]
                if all(os.path.isfile(o) for o in outs):
                    print('{} result-file(s) already exist - skipping this bin'.format(bin_obj))
                    continue
            n_bins += 1
            try:
                ds = IfcbBinDataset(bin_fileset, classifier.hparams.resize, classifier.hparams.img_norm)
                if len(ds) == 0:
                    error_bins.append((bin_obj, AssertionError('Bin is Empty')))
                    continue
                if args.gobig:
                    print('.', end='', flush=True)
                    big.append((bin_obj, ds))
                    continue
                if hasattr(bin_fileset, 'table') and hasattr(bin_fileset, 'blob') and os.environ.get('IFCBK_BIN_RESIDENT', '1') != '0':
                    # one upload per bin, batches cut on the device (f-3); IFCBK_BIN_RESIDENT=0: the per-ROI DataLoader path below
                    trainer.test_resident(classifier, bin_fileset, ds, bin_obj, callbacks, args.batch_size)
                    continue
                loader = DataLoader(ds, batch_size=args.batch_size, pin_memory=True, num_workers=args.loaders,
                                    collate_fn=collate_rois)
                trainer.test(classifier, loader, bin_obj, ds.transform, callbacks)
            except Exception as e:                     # noqa: per-bin isolation as upstream (:266-268)
                error_bins.append((bin_obj, e))
        if args.gobig and big:
            # upstream hands Lightning the list of all bin loaders at once (:261-263,271).  Here "big" means what it buys on the
            # GPU: ONE stream of full fixed-size batches across bin boundaries (a bin's tail no longer runs a short batch, the
            # captured hipGraph of the batch size is replayed throughout); the scores are cut back per bin before the writers
            print()
            try:
                trainer.test_many(classifier, big, args.batch_size, args.loaders, callbacks)
            except Exception as e:                     # noqa
                error_bins.extend((b, e) for b, _ in big)
        print('RUN IS DONE')
        if error_bins:
            print('The following bins failed; they were not processed:')
            for bin_obj, err in error_bins:
                print(bin_obj, type(err), err)
            if n_bins and len(error_bins) >= n_bins:
                raise SystemExit('RUN: every one of the {} bins failed'.format(n_bins))
    else:
        img_paths = []
        if os.path.isdir(args.SRC):
            for pardir, _, imgs in os.walk(args.SRC):
                img_paths.extend(os.path.join(pardir, im) for im in imgs if im.endswith(IMG_EXTENSIONS))
        elif os.path.isfile(args.SRC) and args.SRC.endswith('.txt'):
            with open(args.SRC) as f:
                img_paths = [ln.strip() for ln in f.read().splitlines()]
            img_paths = [p for p in img_paths if p.endswith(IMG_EXTENSIONS)]
        elif args.SRC.endswith(IMG_EXTENSIONS):
            img_paths.append(args.SRC)
        if args.filter:
            for img in img_paths[:]:
                if filter_mode == 'IN' and not any(k in img for k in filter_keywords):
                    img_paths.remove(img)
                elif filter_mode == 'OUT' and any(k in img for k in filter_keywords):
                    img_paths.remove(img)
        assert len(img_paths) > 0, 'No images to process'
        ds = ImageDataset(img_paths, resize=classifier.hparams.resize, input_src=args.SRC)
        loader = DataLoader(ds, batch_size=args.batch_size, pin_memory=True, num_workers=args.loaders,
                            collate_fn=collate_rois)
        trainer.test(classifier, loader, args.SRC, ds.transform, callbacks)


def argparse_nn(parser=None):
    if parser is None:
        parser = argparse.ArgumentParser(description='Train, Run, and perform other tasks related to ifcb and general image classification!')
    subparsers = parser.add_subparsers(dest='cmd_mode', help='These sub-commands are mutually exclusive. Note: optional arguments (below) must be specified before "TRAIN" or "RUN"')
    train = subparsers.add_parser('TRAIN', help='Train a new model')
    run = subparsers.add_parser('RUN', help='Run a previously trained model')
    common = parser.add_argument_group(title='NN Common Args', description=None)
    common.add_argument('--batch', dest='batch_size', metavar='SIZE', default=108, type=int, help='Number of images per batch. Defaults is 108')
    common.add_argument('--loaders', metavar='N', default=4, type=int, help='Number of data-loading threads. 4 per GPU is typical. Default is 4')
    common.add_argument('--precision', choices=['bf16', 'fp32'], default='bf16', help='(MI355X path, additive) activation storage / MFMA type: bf16 = performance mode (default), fp32 = parity mode matching the reference CPU arithmetic to 1e-3')
    argparse_nn_train(train)
    argparse_nn_run(run)
    return parser


def argparse_nn_train(train_subparser):
    t = train_subparser
    t.add_argument('SRC', help='Directory with class-label subfolders and images. May also be a dataset-configuration csv.')
    t.add_argument('MODEL', help='Select a base model. Eg: "inception_v3"')
    t.add_argument('TRAIN_ID', help='Training ID. This value is the default value used by --outdir and --model-id.')
    model = t.add_argument_group(title='Model Adjustments', description=None)
    model.add_argument('--untrain', dest='pretrained', default=True, action='store_false', help='If set, initializes MODEL ~without~ pretrained neurons. Default (unset) is pretrained')
    model.add_argument('--weights', metavar='PATH', default=os.environ.get('IFCBK_PRETRAINED_WEIGHTS'), help='(MI355X path, additive) torchvision state_dict (.pth) standing in for the ImageNet weights the reference downloads when --untrain is not set; there is no network / torchvision here. Default: $IFCBK_PRETRAINED_WEIGHTS')
    model.add_argument('--img-norm', nargs=2, metavar=('MEAN', 'STD'), help='Normalize images by MEAN and STD. eg1: "0.667 0.161", eg2: "0.056,0.058,0.051 0.067,0.071,0.057"')
    data = t.add_argument_group(title='Dataset Adjustments', description=None)
    data.add_argument('--seed', default=0, type=int, help='Set a specific seed for deterministic output & dataset-splitting reproducability.')
    data.add_argument('--split', metavar='T:V', default='80:20', help='Ratio of images per-class to split randomly into Training and Validation datasets. Default is "80:20"')
    data.add_argument('--class-config', metavar=('CSV', 'COL'), nargs=2, help='Skip and combine classes as defined by column COL of a special CSV configuration file')
    data.add_argument('--class-min', metavar='MIN', default=2, type=int, help='Exclude classes with fewer than MIN instances. Default is 2')
    data.add_argument('--class-max', metavar='MAX', default=None, type=int, help='Limit classes to a MAX number of instances.')
    data.add_argument('--swap', default=False, action='store_true', help=argparse.SUPPRESS)
    epochs = t.add_argument_group(title='Epoch Parameters', description=None)
    epochs.add_argument('--emax', metavar='MAX', default=60, type=int, help='Maximum number of training epochs. Default is 60')
    epochs.add_argument('--emin', metavar='MIN', default=10, type=int, help='Minimum number of training epochs. Default is 10')
    epochs.add_argument('--estop', metavar='STOP', default=10, type=int, help='Early Stopping: Number of epochs following a best-epoch after-which to stop training. Set STOP=0 to disable. Default is 10')
    augs = t.add_argument_group(title='Augmentation Options')
    augs.add_argument('--flip', choices=['x', 'y', 'xy', 'x+V', 'y+V', 'xy+V'], help='Training images have 50%% chance of being flipped along the designated axis: (x) vertically, (y) horizontally, (xy) either/both. "+V" includes the Validation dataset')
    out = t.add_argument_group(title='Output Options')
    out.add_argument('--outdir', default='training-output/{TRAIN_ID}', help='Default is "training-output/{TRAIN_ID}"')
    out.add_argument('--model-id', default='{TRAIN_ID}', help='Set a specific model id. Patterns {TRAIN_DATE} and {TRAIN_ID} are recognized. Default is "{TRAIN_ID}"')
    out.add_argument('--epochs-log', metavar='ELOG', default='epochs.csv', help='Specify a csv filename. Default is epochs.csv')
    out.add_argument('--args-log', metavar='ALOG', default='args.yml', help='Specify a human-readable yaml filename. Default is args.yml')
    out.add_argument('--onnx', action='store_true', help='Additionally output an onnx version of the model')
    out.add_argument('--results', dest='result_files', metavar=('FNAME', 'SERIES'), nargs='+', action='append', help='FNAME: validation-results filename or pattern ("{epoch}"); .json .h5 .mat.  SERIES: data to include.')
    optim = t.add_argument_group(title='Optimization', description='(MI355X path, additive: upstream keeps these flags commented out, neuston_net.py:385-390; the defaults are its only behaviour, Adam(lr=0.001))')
    optim.add_argument('--optimizer', default='Adam', choices=['Adam', 'SGD'], help='Select an optimizer. Default is Adam')
    optim.add_argument('--learning-rate', default=0.001, type=float, help='Set a learning rate. Default is 0.001')
    optim.add_argument('--momentum', default=0.0, type=float, help='SGD momentum. Default is 0')
    meta = t.add_argument_group(title='Metadata and Annotations')
    meta.add_argument('--dataset-id', help='Associate a dataset id label with this model')
    meta.add_argument('--notes', help='Add any kind of note to the trained model.')


def argparse_nn_run(run_subparser):
    r = run_subparser
    r.add_argument('SRC', help='Resource(s) to be classified. Accepts a bin, an image, a text-file, or a directory. Directories are accessed recursively')
    r.add_argument('MODEL', help='Path to a previously-trained model file')
    r.add_argument('RUN_ID', help='Run ID. Used by --outdir')
    r.add_argument('--type', dest='src_type', default='bin', choices=['bin', 'img'], help='File type to perform classification on. Defaults is "bin"')
    r.add_argument('--outdir', default='run-output/{RUN_ID}/v3/{MODEL_ID}', help='Default is "run-output/{RUN_ID}/v3/{MODEL_ID}"')
    r.add_argument('--outfile', action='append', help='Name/pattern of the output classification file ({BIN_ID}, {BIN_YEAR}, {BIN_DATE}, {INPUT_SUBDIRS}; .json .mat .h5)')
    r.add_argument('--filter', nargs='+', metavar=('IN|OUT', 'KEYWORD'), help='Explicitly include (IN) or exclude (OUT) bins or image-files by KEYWORDs.')
    r.add_argument('--clobber', action='store_true', help='If set, already processed bins in OUTDIR are reprocessed.')
    r.add_argument('--gobig', action='store_true', help=argparse.SUPPRESS)


def argparse_nn_runtimeparams(args):
    args.cmd_timestamp = dt.datetime.now(dt.timezone.utc).isoformat(timespec='seconds')
    try:
        with open('version') as f:
            args.version = f.read().strip()
    except FileNotFoundError:
        args.version = None
    if torch.cuda.is_available():
        vis = os.environ.get('CUDA_VISIBLE_DEVICES') or os.environ.get('HIP_VISIBLE_DEVICES') or \
            os.environ.get('ROCR_VISIBLE_DEVICES')
        args.gpus = [int(g) for g in vis.split(',')] if vis else list(range(torch.cuda.device_count()))
    else:
        args.gpus = None
    proc_outdir(args)


def proc_outdir(args):
    run_date_str, _ = args.cmd_timestamp.split('T')
    if args.cmd_mode == 'TRAIN':
        args.outdir = args.outdir.format(TRAIN_DATE=run_date_str, TRAIN_ID=args.TRAIN_ID)
    elif args.cmd_mode == 'RUN':
        hp = load_checkpoint_file(args.MODEL)['hyper_parameters']   # no model build
        args.outdir = args.outdir.format(RUN_DATE=run_date_str, RUN_ID=args.RUN_ID, MODEL_ID=hp['model_id'])


if __name__ == '__main__':
    parser = argparse_nn()
    input_args = parser.parse_args()
    argparse_nn_runtimeparams(input_args)
    main(input_args)
