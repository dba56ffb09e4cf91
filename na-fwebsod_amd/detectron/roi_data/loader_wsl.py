"""RoIDataLoader for one-process-per-GPU training.

Reference: detectron/roi_data/loader_wsl.py:53-330 — loader threads build minibatches on the
host, one enqueue thread per GPU feeds a device-side BlobsQueue, and ALL GPUs of the single
process draw from one shared permutation (:198-210).  Here every rank owns its loader: the
epoch permutation is derived from (RNG_SEED, epoch) so all ranks agree on it without
talking, rank r consumes positions r, r+world, ..., loader threads fill a
host queue, and `next_device_batch` stages the next minibatch through pinned memory on a
copy stream so the H2D transfer overlaps the previous iteration's kernels.
Bagging-mixup follows :136-168 (p = 0.2, lambda ~ Beta(a, a), rois of both images on batch
index 0).
"""
import logging
import queue
import random
import threading

import numpy as np
import numpy.random as npr

from detectron.core.config import cfg
from detectron.roi_data.minibatch_wsl import get_minibatch, get_minibatch_blob_names

logger = logging.getLogger(__name__)


def epoch_permutation(n, seed, epoch, ims_per_batch=1, widths=None, heights=None,
                      aspect_grouping=False):
    """Seed-shared permutation of the roidb for `epoch` (identical on every rank)."""
    rng = np.random.RandomState((int(seed) * 1000003 + int(epoch)) % (2 ** 31 - 1))
    if aspect_grouping and widths is not None:
        horz = np.where(np.asarray(widths) >= np.asarray(heights))[0]
        vert = np.where(np.asarray(widths) < np.asarray(heights))[0]
        horz, vert = rng.permutation(horz), rng.permutation(vert)
        mb = ims_per_batch
        inds = np.hstack((horz[:(len(horz) // mb) * mb], vert[:(len(vert) // mb) * mb]))
        inds = inds.reshape(-1, mb)
        return inds[rng.permutation(inds.shape[0])].reshape(-1)
    return rng.permutation(n)


def rank_shard(perm, rank, world, ims_per_batch=1):
    """Minibatch m of the epoch (m = 0,1,...) goes to rank m % world."""
    groups = [perm[i:i + ims_per_batch] for i in range(0, len(perm) - ims_per_batch + 1,
                                                       ims_per_batch)]
    return [g for m, g in enumerate(groups) if m % world == rank]


class RoIDataLoader(object):
    def __init__(self, roidb, num_loaders=4, minibatch_queue_size=64, blobs_queue_capacity=8,
                 rank=0, world_size=1, ims_per_batch=None):
        self._roidb = roidb
        self._rank, self._world = rank, world_size
        self._ims = ims_per_batch or cfg.NAWS.IMS_PER_GPU
        self._lock = threading.Lock()
        self._epoch = 0
        self._pending = []
        self._minibatch_queue = queue.Queue(maxsize=minibatch_queue_size)
        self._num_loaders = num_loaders
        self._stop = threading.Event()
        self._error = None
        self._output_names = get_minibatch_blob_names()
        self._mixup = bool(cfg.WEBLY.WEBLY_ON and cfg.WEBLY.BAGGING_MIXUP)
        if self._mixup:
            self._class2idx = {}
            for i, e in enumerate(roidb):
                cls = int(e['gt_classes'][np.where(e['gt_classes'] > 0)[0]][0])
                self._class2idx.setdefault(cls, []).append(i)
        self._workers = [threading.Thread(target=self._loader_thread, daemon=True)
                         for _ in range(num_loaders)]
        self._copy_stream = None

    # ------------------------------------------------------------ index stream
    def _refill(self):
        perm = epoch_permutation(len(self._roidb), cfg.RNG_SEED, self._epoch, 1,
                                 [r.get('width', 1) for r in self._roidb],
                                 [r.get('height', 1) for r in self._roidb],
                                 cfg.TRAIN.ASPECT_GROUPING)
        self._pending = rank_shard(perm, self._rank, self._world, 1)
        self._epoch += 1

    def _get_next_minibatch_inds(self):
        with self._lock:
            if not self._pending:
                self._refill()
            return [int(i) for i in self._pending.pop(0)]

    # --------------------------------------------------------------- minibatch
    def get_next_minibatch(self):
        """One image (or one mixed pair) as the six loader blobs."""
        db_inds = self._get_next_minibatch_inds()
        mix = self._mixup and npr.random() > 0.8
        if mix:
            e = self._roidb[db_inds[0]]
            cls = int(e['gt_classes'][np.where(e['gt_classes'] > 0)[0]][0])
            db_inds.extend(random.sample(self._class2idx[cls], 1))
        blobs, _valid = get_minibatch([self._roidb[i] for i in db_inds])
        if mix:
            blobs = mixup_blobs(blobs, npr.beta(cfg.WEBLY.BAGGING_MIXUP_ALPHA,
                                                cfg.WEBLY.BAGGING_MIXUP_ALPHA))
        return blobs

    def get_output_names(self):
        return self._output_names

    def _loader_thread(self):
        try:
            while not self._stop.is_set():
                blobs = self.get_next_minibatch()
                for k in self._output_names:
                    assert blobs[k].dtype in (np.int32, np.float32), \
                        'Blob {} of dtype {} must have dtype of np.int32 or np.float32'.format(
                            k, blobs[k].dtype)
                while not self._stop.is_set():
                    try:
                        self._minibatch_queue.put(blobs, timeout=0.5)
                        break
                    except queue.Full:
                        continue
        except Exception as e:  # stop everything on the first failure (coordinator.py:47-55)
            self._error = e
            self._stop.set()
            logger.exception('minibatch loader thread failed')

    def start(self, prefill=False):
        for w in self._workers:
            w.start()
        if prefill:
            import time
            while not self._minibatch_queue.full() and not self._stop.is_set():
                time.sleep(0.05)

    def has_stopped(self):
        return self._stop.is_set()

    def shutdown(self):
        self._stop.set()
        for w in self._workers:
            if w.is_alive():
                w.join(timeout=5)

    def queue_size(self):
        return self._minibatch_queue.qsize()

    # ------------------------------------------------------- host -> device
    def next_host_batch(self, n_images=None):
        """Collate `n_images` single-image minibatches into one per-GPU batch: images
        zero-padded to a common shape, rois re-indexed by position in the batch."""
        n_images = n_images or self._ims
        parts = []
        for _ in range(n_images):
            while True:
                if self._error is not None:
                    raise RuntimeError('roi_data_loader failed') from self._error
                try:
                    parts.append(self._minibatch_queue.get(timeout=0.5))
                    break
                except queue.Empty:
                    continue
        return collate(parts)

    def next_device_batch(self, device, n_images=None):
        """One per-GPU batch on `device`.  Every host array of the batch (rois, labels, the raw
        uint8 images) is packed into ONE pinned staging slot and crosses PCIe as one copy on the
        loader's copy stream; the image blob is then prepared on the device.  The caller's
        current stream is made to wait for the copy stream, so the call itself never blocks on
        the GPU and can be issued while the previous iteration is still running."""
        import torch
        host = self.next_host_batch(n_images)
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=device)
            self._staging = PinnedRing(device)
        raws, mixes = host.pop('_raw', None), host.pop('_mix', None)
        counts = np.bincount(host['rois'][:, 0].astype(np.int64), minlength=host['data'].shape[0])
        seg = [0] + np.cumsum(counts).tolist()
        arrays = {k: v for k, v in host.items() if not (raws is not None and k == 'data')}
        if raws is not None:
            for i, grp in enumerate(raws):
                for j, r in enumerate(grp):
                    arrays[('_im', i, j)] = r['im']
        with torch.cuda.stream(self._copy_stream):
            staged = self._staging.upload(arrays)
            out = {k: v for k, v in staged.items() if not isinstance(k, tuple)}
            if raws is not None:
                ims = [[staged[('_im', i, j)] for j in range(len(grp))] for i, grp in enumerate(raws)]
                out['data'] = device_prep_images(raws, mixes, device, ims)
        cur = torch.cuda.current_stream(device)
        cur.wait_stream(self._copy_stream)
        self._staging.backing.record_stream(cur)
        out['data'].record_stream(cur)
        out['_seg'] = seg            # host-side per-image row offsets (no device sync needed)
        return out


class PinnedRing(object):
    """Host->device staging without per-array pinned allocations (a fresh `pin_memory()` costs
    ~0.7 ms each on this stack, 5 ms per 2-image batch): a ring of reusable pinned byte slots;
    `upload` packs the arrays of one batch into the next slot at 256-byte offsets, issues one
    asynchronous copy into a fresh device buffer on the current stream, and returns typed views
    of that buffer.  A slot is reused only after the copy that read it has completed."""
    ALIGN = 256

    def __init__(self, device, slots=3):
        import torch
        if not _TORCH_DTYPES:
            _init_dtypes()
        self.device = device
        self._slots = [None] * slots
        self._events = [torch.cuda.Event() for _ in range(slots)]
        self._used = [False] * slots
        self._next = 0
        self.backing = None          # device buffer behind the views of the last upload

    def upload(self, arrays):
        import torch
        arrays = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
        offs, total = {}, 0
        for k, v in arrays.items():
            offs[k] = total
            total += (v.nbytes + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        total = max(total, self.ALIGN)
        i = self._next
        self._next = (i + 1) % len(self._slots)
        if self._used[i]:
            self._events[i].synchronize()
        if self._slots[i] is None or self._slots[i].numel() < total:
            self._slots[i] = torch.empty(max(total, 1 << 20) * 5 // 4, dtype=torch.uint8).pin_memory()
        slot = self._slots[i]
        flat = slot.numpy()
        for k, v in arrays.items():
            flat[offs[k]:offs[k] + v.nbytes] = v.reshape(-1).view(np.uint8)
        dev = torch.empty(total, dtype=torch.uint8, device=self.device)
        dev.copy_(slot[:total], non_blocking=True)
        self._events[i].record(torch.cuda.current_stream(self.device))
        self._used[i] = True
        self.backing = dev
        out = {}
        for k, v in arrays.items():
            t = dev[offs[k]:offs[k] + v.nbytes]
            if v.dtype != np.uint8:
                t = t.view(_TORCH_DTYPES[v.dtype.name])
            out[k] = t.reshape(v.shape)
        return out


_TORCH_DTYPES = {}


def _init_dtypes():
    import torch
    _TORCH_DTYPES.update({'float32': torch.float32, 'float64': torch.float64, 'int32': torch.int32,
                          'int64': torch.int64, 'uint8': torch.uint8, 'bool': torch.bool,
                          'float16': torch.float16, 'int16': torch.int16, 'int8': torch.int8})


def device_prep_images(raws, mixes, device, ims=None):
    """`raws`: per batch image, a list of one raw image dict (or two: a bagging-mixup pair);
    -> float32 [B,3,Hmax,Wmax] on `device`, each image prepared by naws_prep_image_fwd into its
    zero-padded slot (blob.py:67-97), pairs blended lam*im0 + (1-lam)*im1 (loader_wsl.py:152).
    `ims`: the uint8 HWC pixels already on the device, same nesting as `raws` (else uploaded here)."""
    import torch
    from naws_hip import ops, lib as L
    hmax = max(r['out_hw'][0] for grp in raws for r in grp)
    wmax = max(r['out_hw'][1] for grp in raws for r in grp)
    data = torch.zeros((len(raws), 3, hmax, wmax), device=device, dtype=torch.float32)
    means, stds = cfg.PIXEL_MEANS.reshape(-1)[:3], np.asarray(cfg.PIXEL_STDS).reshape(-1)[:3]

    def prep(i, j, out):
        r = raws[i][j]
        if ims is not None:
            im = ims[i][j]
        else:
            im = torch.from_numpy(r['im']).to(device)
        ops.prep_image(im, out, r['scale'], flip=r['flip'], crop=r['crop'], means=means, stds=stds,
                       distort=r.get('distort'))
    for i, grp in enumerate(raws):
        if len(grp) == 1:
            prep(i, 0, data[i])
            continue
        lam = float(mixes[i])
        t = torch.zeros((2, 3, hmax, wmax), device=device, dtype=torch.float32)
        prep(i, 0, t[0])
        prep(i, 1, t[1])
        a = ops.unary(L.UN_SCALE, t[0].reshape(-1), np.float32(lam))
        b = ops.unary(L.UN_SCALE, t[1].reshape(-1), np.float32(1 - lam))
        ops.binary(L.BIN_ADD, a.view(1, -1), b.view(1, -1), out=data[i].view(1, -1))
    return data


def mixup_blobs(blobs, lam):
    """data = lam*im0 + (1-lam)*im1, labels_oh likewise, all rois on batch index 0
    (loader_wsl.py:149-168)."""
    out = dict(blobs)
    if '_raw' in blobs:          # device-side preparation: blend there
        out['_raw'] = [list(blobs['_raw'])]
        out['_mix'] = [lam]
        out['data'] = blobs['data'][0:1]
    else:
        out['data'] = (lam * blobs['data'][0:1] + (1 - lam) * blobs['data'][1:2]).astype(np.float32)
    out['labels_oh'] = (lam * blobs['labels_oh'][0:1] +
                        (1 - lam) * blobs['labels_oh'][1:2]).astype(np.float32)
    rois = blobs['rois'].copy()
    rois[:, 0] = 0
    out['rois'] = rois
    out['data_ids'] = blobs['data_ids'][0:1]
    out['labels_int32'] = blobs['labels_int32'][0:1]
    return out


def collate(parts):
    """Stack single-image minibatches into one batch (per-GPU B > 1 is an extension of the
    reference, which asserts IMS_PER_BATCH == 1: wsl_heads.py:214)."""
    if '_raw' in parts[0]:
        # each part: '_raw' is [raw] (plain) or [[raw0, raw1]] (mixup, with '_mix' = [lam])
        for p in parts:
            if not isinstance(p['_raw'][0], list):
                p['_raw'], p['_mix'] = [list(p['_raw'])], [None]
    if len(parts) == 1:
        return parts[0]
    hmax = max(p['data'].shape[2] for p in parts)
    wmax = max(p['data'].shape[3] for p in parts)
    data = np.zeros((len(parts), 3, hmax, wmax), np.float32)
    rois = []
    for i, p in enumerate(parts):
        data[i, :, :p['data'].shape[2], :p['data'].shape[3]] = p['data'][0]
        r = p['rois'].copy()
        r[:, 0] = i
        rois.append(r)
    out = {'data': data, 'rois': np.concatenate(rois)}
    for k in ('data_ids', 'obn_scores', 'labels_int32', 'labels_oh'):
        out[k] = np.concatenate([p[k] for p in parts])
    if '_raw' in parts[0]:
        out['_raw'] = [p['_raw'][0] for p in parts]
        out['_mix'] = [p['_mix'][0] for p in parts]
    return out
