"""Raw IFCB bin ingest (.adc CSV + .roi u8 blob) -- the step before preprocessing on the RUN path
(``/root/reference/neuston_data.py:433-454``, ``neuston_net.py:211-251``).

PARITY UNPINNED: the reference reads bins through pyifcb (un-pinned git dependency, not under /root/reference,
not installed).  This module restates the published IFCB file layout: one ADC row per trigger; target number =
1-based row index; ROI bytes = roi_file[start_byte : start_byte + width*height] as [height, width] u8; rows
with zero area carry no image.  Column indices (0-based): schema v1 (old-style ``IFCBn_YYYY_DDD_HHMMSS`` bins)
x,y,w,h,start = 9..13; schema v2 (``DYYYYMMDDTHHMMSS_IFCBnnn``) x,y,w,h,start = 13..17.  Stitching/infilling of
schema-v1 ROI pairs (pyifcb ``InfilledImages``) is NOT implemented: v1 bins yield their raw ROIs.
"""
import os
import re

import numpy as np

SCHEMA_VERSION_1 = 'v1'
SCHEMA_VERSION_2 = 'v2'
_NEW = re.compile(r'^D(\d{4})(\d{2})(\d{2})T(\d{6})_IFCB(\d+)$')
_OLD = re.compile(r'^IFCB(\d+)_(\d{4})_(\d{3})_(\d{6})$')


class Pid:
    """the subset of pyifcb's Pid the drivers touch: .pid .lid .year .yearday .namespace .target with_target()"""

    def __init__(self, pid, target=None):
        self.lid = pid
        self.target = target
        self.namespace = ''
        m = _NEW.match(pid)
        if m:
            self.schema = SCHEMA_VERSION_2
            self.year = m.group(1)
            self.yearday = m.group(1) + m.group(2) + m.group(3)
        else:
            m = _OLD.match(pid)
            if not m:
                raise ValueError('not an IFCB bin id: %r' % pid)
            self.schema = SCHEMA_VERSION_1
            self.year = m.group(2)
            self.yearday = m.group(2) + '_' + m.group(3)

    @property
    def pid(self):
        return self.lid if self.target is None else '%s_%05d' % (self.lid, self.target)

    def with_target(self, target):
        p = Pid(self.lid, int(target))
        p.namespace = self.namespace
        return p

    def __str__(self):
        return self.namespace + self.pid if self.namespace else self.pid

    __repr__ = __str__


class Bin:
    def __init__(self, basepath):
        self.basepath = basepath
        self.pid = Pid(os.path.basename(basepath))
        self.schema = self.pid.schema
        self._images = None
        self._table = None
        self._blob = None

    @property
    def table(self):
        """the .adc read ONCE into parallel int arrays over the ROIs with an image (zero-area triggers carry none):
        ``targets`` (1-based ADC row = target number), ``offs`` (byte offset of the ROI in the .roi file), ``hs``, ``ws``.
        Together with ``blob`` this is the whole bin: ROI i is blob[offs[i] : offs[i] + hs[i]*ws[i]] as [hs[i], ws[i]] u8 --
        the form ifcbk_roi_preprocess consumes (one upload per bin)."""
        if self._table is None:
            cx = 9 if self.schema == SCHEMA_VERSION_1 else 13
            tg, of, hs, ws = [], [], [], []
            with open(self.basepath + '.adc') as f:
                for n, line in enumerate(f, 1):
                    cols = line.strip().split(',')
                    if len(cols) <= cx + 4:
                        continue
                    w, h, start = int(float(cols[cx + 2])), int(float(cols[cx + 3])), int(float(cols[cx + 4]))
                    if w * h == 0:
                        continue
                    if w < 0 or h < 0 or start < 0:
                        raise ValueError('%s.adc row %d: negative ROI size or offset (w=%d h=%d start=%d)' % (self.basepath, n, w, h, start))
                    tg.append(n); of.append(start); hs.append(h); ws.append(w)
            self._table = dict(targets=np.asarray(tg, dtype=np.int64), offs=np.asarray(of, dtype=np.int64),
                               hs=np.asarray(hs, dtype=np.int32), ws=np.asarray(ws, dtype=np.int32))
            size = os.path.getsize(self.basepath + '.roi')
            t = self._table
            if len(tg) and int((t['offs'] + t['hs'].astype(np.int64) * t['ws']).max()) > size:
                raise ValueError('%s: an ADC row points past the end of the .roi file (%d bytes)' % (self.basepath, size))
        return self._table

    @property
    def blob(self):
        """the .roi file as one u8 array (read once)"""
        if self._blob is None:
            self._blob = np.fromfile(self.basepath + '.roi', dtype=np.uint8)
        return self._blob

    @property
    def images(self):
        """{target_number: u8 [h,w]} -- all ROIs of the bin in RAM, as IfcbBinDataset expects (:446-454): views into ``blob``."""
        if self._images is None:
            t, roi = self.table, self.blob
            self._images = {int(n): roi[o:o + h * w].reshape(h, w) for n, o, h, w in zip(t['targets'], t['offs'], t['hs'], t['ws'])}
        return self._images

    def __len__(self):
        return len(self.images)


class DataDirectory:
    """recursive listing of bins (an .adc with its .roi) under ``path``; white/blacklist match on the bin id."""

    def __init__(self, path, whitelist=None, blacklist=None):
        self.path, self.whitelist, self.blacklist = path, whitelist, blacklist

    def __iter__(self):
        for pardir, dirs, files in os.walk(self.path):
            dirs.sort()
            for f in sorted(files):
                if not f.endswith('.adc'):
                    continue
                lid = f[:-4]
                if not (_NEW.match(lid) or _OLD.match(lid)) or not os.path.exists(os.path.join(pardir, lid + '.roi')):
                    continue
                if self.whitelist and not any(k in lid or k in os.path.join(pardir, lid) for k in self.whitelist):
                    continue
                if self.blacklist and any(k in lid for k in self.blacklist):
                    continue
                yield Bin(os.path.join(pardir, lid))
