"""Training-data pipeline on the device: drop-in for `generate_arrays_from_file` /
`load_seq_crop_data_masktumor_try` of train_2ddense.py:40-133 and train_hybrid.py:40-133.

The reference keeps all 131 pre-processed volumes in host RAM (`load_fast_files`, train_2ddense.py:129-170) and builds
every batch with a 14-thread pool: random scale, crop around a liver / tumour voxel, mean subtraction, one of 8 flips /
rotations, bicubic / nearest resize (scikit-image), then feeds numpy arrays through feed_dict.  At 340+ slices/s the
host pool is the bottleneck by two orders of magnitude.  Here the volumes live in HBM (MI355X: 288 GB; the whole LiTS
training set is ~60 GB in float32) and ONE kernel pair per batch (csrc/augment.hip) writes the samples straight into the
model's input / label buffers.  The host only draws the random parameters -- in the reference's order, from an explicit
numpy RandomState -- and uploads n small descriptors.

    ds = DeviceDataset(img_list, tumor_list, liver_centres, tumor_centres, minindex_list, maxindex_list, mean=48)
    gen = ds.generator(model, batch_size, size=args.input_size, cols=args.input_cols, hybrid=False, seed=0)
    model.fit_generator(gen, steps_per_epoch, epochs, ...)        # the generator yields (DeviceBatch, None)
"""
import ctypes

import numpy as np
import torch

from . import lib as _l
from . import ops

LIVERLIST = (32, 34, 38, 41, 47, 87, 89, 91, 105, 106, 114, 115, 119)     # train_2ddense.py:39: cases without tumour lines
FLIPS = 8


class DeviceBatch:
    """one batch already resident in the model's input / label buffers (Model.train_on_batch accepts it as `x`)"""

    def __init__(self, dataset, params, hybrid):
        self.dataset, self.params, self.hybrid = dataset, params, hybrid

    def fill_model(self, model):
        if getattr(self, "_filled", None) is not model:       # (the hybrid generator already filled it for its class check)
            self.dataset.fill(model, self.params, self.hybrid)
            self._filled = model


class DeviceDataset:
    """the pre-processed training volumes resident in HBM: float32 CT [rows][cols][slices] + uint8 labels of the same shape
    per case (what load_fast_files keeps in host lists), the liver bounding boxes and the liver / tumour voxel lists the
    crop centres are drawn from (myTraining_Data248/{Liver,Tumor}Pixels/*.txt, one "r c s" line per voxel)."""

    def __init__(self, img_list, tumor_list, liver_centres, tumor_centres, minindex_list, maxindex_list, mean=48.0):
        dev = ops.device()
        self.shapes, self.offsets = [], []
        off = 0
        for im, tu in zip(img_list, tumor_list):
            if im.shape != tu.shape or im.ndim != 3:
                raise ValueError("image / label volumes must be 3-D and of equal shape")
            self.shapes.append(tuple(int(v) for v in im.shape))
            self.offsets.append(off)
            off += int(np.prod(im.shape))
        self.img = torch.empty(off, dtype=torch.float32, device=dev)
        self.lab = torch.empty(off, dtype=torch.uint8, device=dev)
        for im, tu, o in zip(img_list, tumor_list, self.offsets):
            n = int(np.prod(im.shape))
            self.img[o:o + n] = torch.from_numpy(np.ascontiguousarray(im, np.float32).reshape(-1)).to(dev)
            self.lab[o:o + n] = torch.from_numpy(np.ascontiguousarray(tu).astype(np.uint8).reshape(-1)).to(dev)
        self.liver_centres = [np.asarray(c, np.int64).reshape(-1, 3) for c in liver_centres]
        self.tumor_centres = [np.asarray(c, np.int64).reshape(-1, 3) for c in tumor_centres]
        self.minindex = [np.asarray(m, np.int64) for m in minindex_list]
        self.maxindex = [np.asarray(m, np.int64) for m in maxindex_list]
        self.mean = float(mean)
        self._ws = None

    # ---- the random draws of one sample, in the reference's order (train_2ddense.py:49-72, :112-123)
    def draw(self, rng, size, cols, trainidx):
        count = int(trainidx[rng.randint(0, len(trainidx))])             # random.choice(trainidx)
        num = rng.randint(0, 6)
        if num < 3 or count in LIVERLIST or len(self.tumor_centres[count]) == 0:
            centres = self.liver_centres[count]
        else:
            centres = self.tumor_centres[count]
        scale = rng.uniform(0.8, 1.2)
        crop = int(size * scale)
        # np.random.randint(1, numid) with numid = len(lines) (train_2ddense.py:158-164, :52-58): the reference never draws
        # the LAST voxel line; kept draw for draw
        sed = rng.randint(1, len(centres)) if len(centres) > 1 else 1
        cen = centres[sed - 1]
        mn, mx = self.minindex[count], self.maxindex[count]
        a = min(max(mn[0] + crop // 2, cen[0]), mx[0] - crop // 2 - 1)
        b = min(max(mn[1] + crop // 2, cen[1]), mx[1] - crop // 2 - 1)
        c = min(max(mn[2] + cols // 2, cen[2]), mx[2] - cols // 2 - 1)
        flip = int(rng.randint(0, FLIPS))
        return dict(case=count, deps=crop, a=int(a), b=int(b), c=int(c), flip=flip)

    def _descriptor(self, prm, cols, hybrid):
        rows, vcols, vsl = self.shapes[prm["case"]]
        d = prm["deps"]
        c0 = prm["c"] - cols // 2 if hybrid else prm["c"] - 1           # train_hybrid.py:63-66 / train_2ddense.py:64-67
        a0, b0 = prm["a"] - d // 2, prm["b"] - d // 2
        crop = 2 * (d // 2)                                              # img[a - d/2 : a + d/2] has 2*(d//2) rows
        n_sl = cols if hybrid else 3
        if a0 < 0 or b0 < 0 or c0 < 0 or a0 + crop > rows or b0 + crop > vcols or c0 + n_sl > vsl:
            raise ValueError("crop leaves the volume: the liver bounding box must hold the largest crop (the reference "
                             "would silently produce a short crop here)")
        return _l.AugSample(self.offsets[prm["case"]], rows, vcols, vsl, a0, b0, c0, crop, prm["flip"])

    def fill(self, model, params, hybrid):
        """crop / flip / resize every sample of `params` straight into the model's input and label buffers"""
        n = len(params)
        dev = ops.device()
        sh = getattr(model.ctx, "shard", None)
        if sh is not None and sh.world > 1:
            # a depth shard holds its OWN planes of the volume plus one CT plane of each neighbour (Model._upload_x crops
            # per rank and exchanges them); this kernel would write the same D_local slices on every rank
            raise NotImplementedError("DeviceDataset.fill does not feed a depth-sharded model: use Model.train_on_batch(x, y) "
                                      "with this rank's planes")
        shp = model.input_shape
        size = shp[1]
        if hybrid:
            if model.kind != "hybrid" or n != 1:
                raise ValueError("the hybrid generator feeds one volume per step (args.b = 1)")
            cols = shp[3]
            x_out, xs, xp, xk = model.vol, 0, 1, size * size              # depth-major [D][H][W]
            y_out, ys, yk, lab_slice = model.loss_layer.labels, 0, size * size, -1
        else:
            if model.kind != "2d" or n != shp[0]:
                raise ValueError("batch size %d does not match the model's %d" % (n, shp[0]))
            cols = 3
            x_out, xs, xp, xk = model.x_stage, size * size * 3, 3, 1       # [N][H][W][3]
            y_out, ys, yk, lab_slice = model.loss_layer.labels, size * size, 0, 1
        tab = (_l.AugSample * n)(*[self._descriptor(p, cols, hybrid) for p in params])
        tab_dev = torch.from_numpy(np.frombuffer(bytes(tab), dtype=np.uint8).copy()).to(dev)
        if self._ws is None or self._ws.numel() < 2 * n:
            self._ws = torch.empty(2 * max(n, 16), dtype=torch.float32, device=dev)
        _l.check(_l.get().hdu_augment_batch(
            ctypes.c_void_p(self.img.data_ptr()), ctypes.c_void_p(self.lab.data_ptr()), ctypes.c_void_p(tab_dev.data_ptr()),
            n, size, cols if hybrid else 3, lab_slice, self.mean, ctypes.c_void_p(self._ws.data_ptr()),
            ctypes.c_void_p(x_out.data_ptr()), xs, xp, xk, ctypes.c_void_p(y_out.data_ptr()), ys, yk, ops.stream()),
            "hdu_augment_batch")
        self._keep = tab_dev          # stays alive until the next batch (the launch is asynchronous)

    def generator(self, model, batch_size, size, cols, trainidx=None, hybrid=False, seed=0):
        """`generate_arrays_from_file` (train_2ddense.py:108-127 / train_hybrid.py:102-133): endless generator of
        (DeviceBatch, None).  The hybrid form skips volumes that lack one of the three classes (:127-132) -- checked on
        the device labels (one 3-int read-back per step)."""
        rng = np.random.RandomState(seed)
        trainidx = list(range(len(self.shapes))) if trainidx is None else list(trainidx)
        while True:
            params = [self.draw(rng, size, cols, trainidx) for _ in range(batch_size)]
            batch = DeviceBatch(self, params, hybrid)
            if hybrid:
                batch.fill_model(model)
                cnt = torch.bincount(model.loss_layer.labels.to(torch.int64), minlength=3)[:3].tolist()
                if min(cnt) == 0:
                    continue
            yield batch, None
