"""Host-side twin of ``/root/reference/neuston_data.py``: dataset selection / splitting and the per-item
input contract, re-cut for the MI355X path.

What changes versus the reference: items are NOT resized/normalised on the CPU.  ``__getitem__`` returns the
decoded u8 image (HxW for grayscale ROIs, HxWx3 otherwise) plus a per-item flip code; ``collate_rois`` packs a
batch into one ragged u8 blob + offset table (pinned), and the resize / ToTensor / Normalize chain of
``get_trainval_transforms`` (:342-371) runs on the GPU in ``ifcbk_roi_preprocess`` (bit-exact to PIL).
What does not change: class-folder scanning, class-min/max, class-config CSV, dataset-config CSV,
``split`` (incl. its re-seeding quirk), ``parse_imgnorm``, resize = 299 iff MODEL == 'inception_v3'.
"""
import os
import random

import numpy as np
import torch
from torch.utils.data.dataset import Dataset

# torchvision.datasets.folder.IMG_EXTENSIONS (the reference filters files with it, neuston_data.py:69)
IMG_EXTENSIONS = ('.jpg', '.jpeg', '.png', '.ppm', '.bmp', '.pgm', '.tif', '.tiff', '.webp')


def default_loader(path):
    """[TV] datasets.folder.default_loader = PIL open -> convert('RGB').  A grayscale file stays 2-D here
    (L -> RGB only replicates the channel; the GPU kernel replicates instead) -- same pixels, 1/3 the bytes."""
    from PIL import Image
    with open(path, 'rb') as f:
        img = Image.open(f)
        if img.mode == 'L':
            return np.asarray(img).copy()
        return np.asarray(img.convert('RGB')).copy()


class RoiTransform:
    """What ``transforms.Compose([flips] + [Resize, ToTensor, Normalize?])`` means on the GPU path."""

    def __init__(self, resize, img_norm=None, vflip=False, hflip=False):
        self.resize = resize
        self.img_norm = img_norm          # (mean[3], std[3]) or None
        self.vflip, self.hflip = vflip, hflip

    def flip_code(self):
        """bit0 = vertical flip ('x'), bit1 = horizontal flip ('y'); each with p = 0.5 (RandomVertical/HorizontalFlip)"""
        code = 0
        if self.vflip and random.random() < 0.5:
            code |= 1
        if self.hflip and random.random() < 0.5:
            code |= 2
        return code


class NeustonDataset(Dataset):
    """neuston_data.py:21-270.  Folders of ``src`` are the classes."""

    def __init__(self, src, minimum_images_per_class=1, maximum_images_per_class=None, transforms=None,
                 images_perclass=None):
        self.src = src
        if not images_perclass:
            images_perclass = self.fetch_images_perclass(src)
        self.minimum_images_per_class = max(1, minimum_images_per_class)
        kept = {c: imgs for c, imgs in images_perclass.items() if len(imgs) >= self.minimum_images_per_class}
        dropped = sorted(set(images_perclass) - set(kept))
        self.classes_ignored_from_too_few_samples = [(c, len(images_perclass[c])) for c in dropped]
        self.classes = sorted(kept)
        self.maximum_images_per_class = maximum_images_per_class
        if maximum_images_per_class:
            assert maximum_images_per_class > self.minimum_images_per_class
            limited = {}
            for c, imgs in kept.items():
                limited[c] = sorted(random.sample(imgs, maximum_images_per_class)) \
                    if maximum_images_per_class < len(imgs) else imgs
            self.classes_limited_from_too_many_samples = [c for c in self.classes if len(limited[c]) < len(kept[c])]
            kept = limited
        else:
            self.classes_limited_from_too_many_samples = None
        kept = {c: sorted(imgs) for c, imgs in kept.items()}
        pairs = [(self.classes.index(c), i) for c in kept for i in kept[c]]
        self.targets, self.images = zip(*pairs)
        self.transforms = transforms

    @classmethod
    def fetch_images_perclass(cls, src, include_exclude_rename=None):
        if os.path.isdir(src) and include_exclude_rename is None:
            classes = sorted(d.name for d in os.scandir(src) if d.is_dir())
            out = {}
            for sub in classes:
                files = sorted(f for f in os.listdir(os.path.join(src, sub)) if os.path.splitext(f)[1] in IMG_EXTENSIONS)
                out[sub] = [os.path.join(src, sub, f) for f in files]
            return out
        if os.path.isdir(src):
            out = cls.fetch_images_perclass(src)
            for key, mode in include_exclude_rename:
                if mode == 1 or mode == '1':
                    continue
                if (mode == 0 or mode == '0') and key in out:
                    del out[key]
                else:                                   # rename / merge
                    if key not in out:
                        continue
                    if mode in out:
                        out[mode].extend(out[key])
                    else:
                        out[mode] = out[key]
                    del out[key]
            return out
        # dataset-configuration csv: columns "[priority:]dataset_dir", rows = classes  (neuston_data.py:91-140)
        import pandas as pd
        df = pd.read_csv(src, header=0, index_col=0)
        entries = []
        for col in df.columns.to_list():
            parts = col.split(':', 1)
            priority, dataset = (int(parts[0]), parts[1]) if len(parts) == 2 else (0, parts[0])
            ipc = cls.fetch_images_perclass(dataset, include_exclude_rename=zip(df.index, df[col].to_list()))
            entries.append((priority, dataset, ipc))
        prios = [p for p, _, _ in entries]
        prios = set(max(prios) + 1 if p == 0 else p for p in prios)
        entries = [((max(prios) if p == 0 else p), d, i) for p, d, i in entries]

        def extend(d1, d2):
            for k in d2:
                if k in d1:
                    d1[k].extend(d2[k])
                else:
                    d1[k] = d2[k]
        out = {}
        for level in sorted(prios):
            merged = {}
            for p, _, ipc in entries:
                if p == level:
                    extend(merged, ipc)
            for k in merged:
                random.shuffle(merged[k])
            extend(out, merged)
        return out

    @property
    def images_perclass(self):
        ipc = {c: [] for c in self.classes}
        for img, trg in zip(self.images, self.targets):
            ipc[self.classes[trg]].append(img)
        return ipc

    @property
    def count_perclass(self):
        cpc = [0] * len(self.classes)
        for t in self.targets:
            cpc[t] += 1
        return cpc

    def split(self, ratio1, ratio2, seed=None, minimum_images_per_class='scale'):
        assert ratio1 + ratio2 == 100, 'ratio1:ratio2 must sum to 100, instead got {}:{} (total: {})'.format(
            ratio1, ratio2, ratio1 + ratio2)
        d1, d2 = {}, {}
        for label, images in self.images_perclass.items():
            n1 = int(ratio1 * len(images) / 100 + 0.5)
            if n1 == len(images) and self.minimum_images_per_class > 1:
                n1 -= 1                                   # keep one for the second set
            if seed:
                random.seed(seed)                          # re-seeded for every class, as upstream (:169-171)
            pick = random.sample(images, n1)
            rest = sorted(set(images) - set(pick))
            assert len(pick) + len(rest) == len(images)
            d1[label], d2[label] = pick, rest
        ds1 = NeustonDataset(src=self.src, images_perclass=d1, transforms=self.transforms)
        ds2 = NeustonDataset(src=self.src, images_perclass=d2, transforms=self.transforms)
        assert ds1.classes == ds2.classes, 'd1-d2_classes:{}, d2-d1_classes:{}'.format(
            set(ds1.classes) - set(ds2.classes), set(ds2.classes) - set(ds1.classes))
        assert len(ds1) + len(ds2) == len(self), 'd1_len:{}, d2_len:{}'.format(len(ds1), len(ds2))
        return ds1, ds2

    @classmethod
    def from_csv(cls, src, csv_file, column_to_run, transforms=None, minimum_images_per_class=1,
                 maximum_images_per_class=None):
        import pandas as pd
        df = pd.read_csv(csv_file, header=0)
        base_list = df.iloc[:, 0].tolist()
        mod_list = df[column_to_run].tolist()
        found = cls.fetch_images_perclass(src)
        missing_src = [c for c in found if c not in base_list]
        new, missing_csv, skipped, grouped = {}, [], [], {}
        for base, mod in zip(base_list, mod_list):
            if base not in found:
                missing_csv.append(base)
                continue
            if str(mod) == '0':
                skipped.append(base)
                continue
            if str(mod) == '1':
                label = base
            else:
                label = mod
                grouped.setdefault(mod, []).append(base)
            if label not in new:
                new[label] = found[base]
            else:
                new[label].extend(found[base])
        name = os.path.basename(csv_file)
        if missing_src:
            print('\n    '.join(['\n{} of {} classes from src dir {} were NOT FOUND in {}'.format(
                len(missing_src), len(found), src, name)] + missing_src))
        if missing_csv:
            print('\n    '.join(['\n{} of {} classes from {} were NOT FOUND in src dir {}'.format(
                len(missing_csv), len(base_list), name, src)] + missing_csv))
        if grouped:
            print('\n{} GROUPED classes were created, as per {}'.format(len(grouped), name))
            for mod, bases in grouped.items():
                print('  {}'.format(mod))
                print('\n'.join('     <-- {}'.format(c) for c in bases))
        if skipped:
            print('\n    '.join(['\n{} classes were SKIPPED, as per {}'.format(len(skipped), name)] + skipped))
        return cls(src=src, images_perclass=new, transforms=transforms,
                   minimum_images_per_class=minimum_images_per_class,
                   maximum_images_per_class=maximum_images_per_class)

    def __getitem__(self, index):
        path = self.images[index]
        data = default_loader(path)
        flip = self.transforms.flip_code() if self.transforms is not None else 0
        return (data, flip), self.targets[index], path

    def __len__(self):
        return len(self.images)

    @property
    def imgs(self):
        return self.images


def get_trainval_datasets(args):
    """neuston_data.py:292-329"""
    print('Initializing Data...')
    if not args.class_config:
        nd = NeustonDataset(src=args.SRC, minimum_images_per_class=args.class_min,
                            maximum_images_per_class=args.class_max)
    else:
        nd = NeustonDataset.from_csv(src=args.SRC, csv_file=args.class_config[0], column_to_run=args.class_config[1],
                                     minimum_images_per_class=args.class_min, maximum_images_per_class=args.class_max)
    r1, r2 = map(int, args.split.split(':'))
    pair = nd.split(r1, r2, seed=args.seed)
    training, validation = pair if not args.swap else pair[::-1]
    ci_nd = nd.classes_ignored_from_too_few_samples
    ci_train = training.classes_ignored_from_too_few_samples
    ci_eval = validation.classes_ignored_from_too_few_samples
    assert ci_eval == ci_train
    if ci_nd:
        msg = '\n{} out of {} classes ignored from --class-minimum {}, PRE-SPLIT'.format(
            len(ci_nd), len(nd.classes + ci_nd), args.class_min)
        print('\n    '.join([msg] + ['({:2}) {}'.format(l, c) for c, l in ci_nd]))
    if ci_eval:
        msg = '\n{} out of {} classes ignored from --class-minimum {}, POST-SPLIT'.format(
            len(ci_eval), len(validation.classes + ci_eval), args.class_min)
        print('\n    '.join([msg] + ['({:2}) {}'.format(l, c) for c, l in ci_eval]))
    training.transforms, validation.transforms = get_trainval_transforms(args)
    return training, validation


def parse_imgnorm(img_norm_arg):
    """neuston_data.py:331-339: "m" / "m1,m2,m3" strings -> two 3-lists of floats."""
    mean = [float(m) for m in img_norm_arg[0].split(',')]
    if len(mean) == 1:
        mean = 3 * mean
    std = [float(s) for s in img_norm_arg[1].split(',')]
    if len(std) == 1:
        std = 3 * std
    assert len(mean) == len(std) == 3, '--img-norm invalid: {}'.format(img_norm_arg)
    return mean, std


def get_trainval_transforms(args):
    """neuston_data.py:342-371; sets args.resize (299 only for the exact name 'inception_v3')."""
    args.resize = 299 if args.MODEL == 'inception_v3' else 224
    norm = parse_imgnorm(args.img_norm) if args.img_norm else None
    flip = args.flip or ''
    vflip, hflip = 'x' in flip, 'y' in flip                # 'x' = vertical, 'y' = horizontal (sic)
    train = RoiTransform(args.resize, norm, vflip, hflip)
    val = RoiTransform(args.resize, norm, vflip and '+V' in flip, hflip and '+V' in flip)
    return train, val


class ImageDataset(Dataset):
    """neuston_data.py:376-406 (RUN --type img).  No Normalize, as upstream (quirk: img_norm is ignored here)."""

    def __init__(self, image_paths, resize=244, input_src=None):
        self.input_src = input_src
        self.image_paths = [img for img in image_paths if img.endswith(IMG_EXTENSIONS)]
        self.transform = RoiTransform(resize)
        if len(self.image_paths) < len(image_paths):
            print('{} non-image files were ommited'.format(len(image_paths) - len(self.image_paths)))
        if len(self.image_paths) == 0:
            raise RuntimeError('No images Loaded!!')

    def __getitem__(self, index):
        path = self.image_paths[index]
        return (default_loader(path), 0), path

    def __len__(self):
        return len(self.image_paths)


class IfcbBinDataset(Dataset):
    """neuston_data.py:433-467.  ``bin`` is any object with ``.pid`` (``with_target(n)``), ``.schema`` and
    ``.images`` ({target_number: 2-D u8 array}); schema-v1 bins must already be stitched/infilled (pyifcb's
    ``InfilledImages`` is not available here: parity unpinned for that step)."""

    def __init__(self, bin, resize, img_norm=None):
        self.bin = bin
        self.images, self.pids = [], []
        self.img_norm = parse_imgnorm(img_norm) if img_norm else None
        self.resize = resize[0] if isinstance(resize, (tuple, list)) else resize
        self.transform = RoiTransform(self.resize, self.img_norm)
        if getattr(bin, 'schema', None) == 'v1' and not getattr(bin, 'stitched', False) \
                and os.environ.get('IFCBK_ALLOW_UNSTITCHED_V1', '0') == '0':
            # upstream reads old-style bins through pyifcb's InfilledImages (stitched ROI pairs, :446-449); that algorithm is
            # not available here, and classifying the raw halves would silently change the ROI set of the bin
            raise NotImplementedError('{}: schema-v1 (old-style) bin -- ROI stitching / infilling is not implemented on this path; '
                                      'set IFCBK_ALLOW_UNSTITCHED_V1=1 to classify its raw ROIs knowingly'.format(bin.pid))
        for target_number, img in bin.images.items():
            self.images.append(np.ascontiguousarray(img, dtype=np.uint8))
            self.pids.append(bin.pid.with_target(target_number))

    def __getitem__(self, item):
        return (self.images[item], 0), self.pids[item]

    def __len__(self):
        return len(self.pids)


# ------------------------------------------------------------------------------------------ batching
def collate_rois(items):
    """DataLoader collate_fn: [( (img_u8, flip), *rest )] -> (roi_batch dict, *rest lists).  One ragged u8 blob,
    an int64 offset table and int32 dims; tensors are pinned by the loader (pin_memory=True)."""
    imgs = [it[0][0] for it in items]
    flips = [it[0][1] for it in items]
    ch = 3 if any(im.ndim == 3 for im in imgs) else 1
    if ch == 3:
        imgs = [im if im.ndim == 3 else np.repeat(im[:, :, None], 3, 2) for im in imgs]
    hs = torch.tensor([im.shape[0] for im in imgs], dtype=torch.int32)
    ws = torch.tensor([im.shape[1] for im in imgs], dtype=torch.int32)
    sizes = hs.long() * ws.long() * ch
    offs = torch.zeros(len(imgs), dtype=torch.int64)
    if len(imgs) > 1:
        offs[1:] = torch.cumsum(sizes, 0)[:-1]
    blob = torch.from_numpy(np.concatenate([np.ascontiguousarray(im).reshape(-1) for im in imgs]))
    batch = dict(pixels=blob, offs=offs, hs=hs, ws=ws, flips=torch.tensor(flips, dtype=torch.uint8),
                 max_h=int(hs.max()), max_w=int(ws.max()), in_channels=ch)
    rest = list(zip(*[it[1:] for it in items]))
    out = [batch]
    for r in rest:
        r = list(r)
        out.append(torch.tensor(r, dtype=torch.int64) if isinstance(r[0], (int, np.integer)) else r)
    return tuple(out)


def rois_to_device(batch, device, transform=None):
    """upload a collated ROI batch (u8 blob + tables) and attach Normalize parameters."""
    kw = dict(pixels=batch['pixels'].to(device, non_blocking=True), offs=batch['offs'].to(device, non_blocking=True),
              hs=batch['hs'].to(device, non_blocking=True), ws=batch['ws'].to(device, non_blocking=True),
              max_h=batch['max_h'], max_w=batch['max_w'], in_channels=batch['in_channels'])
    if batch['flips'].any():
        kw['flips'] = batch['flips'].to(device, non_blocking=True)
    if transform is not None and transform.img_norm is not None:
        kw['mean'], kw['std'] = transform.img_norm
    return kw
