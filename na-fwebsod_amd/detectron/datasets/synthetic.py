"""Synthetic roidb / minibatch / weights for the hot path (no dataset, no checkpoint
can be fetched offline).  Shapes follow SURVEY.md §8(d):

  images     B x [3,600,1000] fp32 = uint8 U[0,255] minus PIXEL_MEANS (BGR)
  proposals  R boxes per image in input-frame pixels: integer x1,y1 ~ U,
             w,h = exp(U[log 21, log dim]) clipped to the image, de-duplicated
  obn_scores U[0,1] sorted descending, +1 (roi_data/wsl.py:103 adds the 1)
  labels     one random foreground class, one-hot [1,C]
  weights    seeded Kaiming-normal conv / fc6 / fc7 with zero bias, Xavier-uniform fc8*
             (XavierFill, wsl_heads.py:29-46)

The roidb entry schema is the reference's ({image, flipped, boxes, obn_scores,
gt_classes}; roi_data/wsl.py:87-166) so the same `_sample_rois` path consumes it.
"""
import numpy as np

PIXEL_MEANS_BGR = np.array([103.939, 116.779, 123.68], np.float32)


def make_boxes(rng, n, height, width):
    """[n,4] float32 (x1,y1,x2,y2), sides > 20 px, unique."""
    out = np.zeros((0, 4), np.float32)
    while out.shape[0] < n:
        m = (n - out.shape[0]) * 2 + 16
        w = np.exp(rng.uniform(np.log(21.0), np.log(width), m))
        h = np.exp(rng.uniform(np.log(21.0), np.log(height), m))
        x1 = np.floor(rng.uniform(0, width - 22, m))
        y1 = np.floor(rng.uniform(0, height - 22, m))
        x2 = np.minimum(np.floor(x1 + w), width - 1)
        y2 = np.minimum(np.floor(y1 + h), height - 1)
        b = np.stack([x1, y1, x2, y2], 1).astype(np.float32)
        b = b[((b[:, 2] - b[:, 0]) > 20) & ((b[:, 3] - b[:, 1]) > 20)]
        out = np.unique(np.concatenate([out, b], 0), axis=0)
        rng.shuffle(out)
    return out[:n]


def make_roidb(n_images, rois_per_image, num_fg_classes, height=600, width=1000, seed=11):
    rng = np.random.default_rng(seed)
    roidb = []
    for i in range(n_images):
        boxes = make_boxes(rng, rois_per_image, height, width)
        scores = np.sort(rng.uniform(0, 1, (rois_per_image, 1)).astype(np.float32), 0)[::-1].copy()
        cls = int(rng.integers(1, num_fg_classes + 1))
        gt = np.zeros((rois_per_image,), np.int32)
        gt[0] = cls                       # the image-level label rides on the first box
        roidb.append(dict(image='synthetic_%06d.jpg' % i, flipped=False, height=height,
                          width=width, boxes=boxes, obn_scores=scores, gt_classes=gt,
                          seed=int(rng.integers(0, 2 ** 31 - 1))))
    return roidb


def make_image(entry):
    """[3,H,W] fp32 BGR mean-subtracted, NCHW plane order."""
    rng = np.random.default_rng(entry['seed'])
    im = rng.integers(0, 256, (entry['height'], entry['width'], 3), dtype=np.uint8)
    return (im.astype(np.float32) - PIXEL_MEANS_BGR).transpose(2, 0, 1).copy()


def make_minibatch(entries, num_fg_classes, max_rois=2048):
    """The six loader blobs (minibatch_wsl.py:25-50 order) for a list of roidb entries,
    image scale 1, no crop: data, data_ids, rois, obn_scores, labels_int32, labels_oh."""
    data = np.stack([make_image(e) for e in entries], 0)
    rois, obn, li, lo = [], [], [], []
    for b, e in enumerate(entries):
        n = min(max_rois, e['boxes'].shape[0])
        rois.append(np.hstack([np.full((n, 1), b, np.float32), e['boxes'][:n]]))
        obn.append(e['obn_scores'][:n] + 1.0)
        cls = int(e['gt_classes'][e['gt_classes'] > 0][-1])
        oh = np.zeros((1, num_fg_classes), np.float32)
        oh[0, cls - 1] = 1
        lo.append(oh)
        li.append(np.array([cls - 1], np.int32))
    return dict(data=data, data_ids=np.zeros((len(entries), 1), np.int32),
                rois=np.concatenate(rois).astype(np.float32),
                obn_scores=np.concatenate(obn).astype(np.float32),
                labels_int32=np.concatenate(li), labels_oh=np.concatenate(lo))


CONV_SHAPES = [('conv1_1', 3, 64), ('conv1_2', 64, 64), ('conv2_1', 64, 128), ('conv2_2', 128, 128),
               ('conv3_1', 128, 256), ('conv3_2', 256, 256), ('conv3_3', 256, 256),
               ('conv4_1', 256, 512), ('conv4_2', 512, 512), ('conv4_3', 512, 512),
               ('conv5_1', 512, 512), ('conv5_2', 512, 512), ('conv5_3', 512, 512)]


def init_blobs(num_fg_classes, seed=11, device='cpu', roi_size=7, pixel_scale=1.0 / 64.0):
    """Random-init weights of the reference blob names / layouts (torch tensors).

    conv1_1 is scaled by `pixel_scale` so that activations of a +-128 pixel input are O(1)
    through the stack (a pretrained VGG maps them to a similar range); every other layer is
    plain Kaiming-normal, fc8* Xavier-uniform, biases zero."""
    import torch
    g = torch.Generator(device='cpu')
    g.manual_seed(seed)
    blobs = {}

    def normal(shape, std):
        return (torch.randn(shape, generator=g) * std).to(device)

    for name, cin, cout in CONV_SHAPES:
        std = (2.0 / (9 * cin)) ** 0.5
        if name == 'conv1_1':
            std *= pixel_scale
        blobs[name + '_w'] = normal((cout, cin, 3, 3), std)
        blobs[name + '_b'] = torch.zeros((cout,), device=device)
    k6 = 512 * roi_size * roi_size
    for pre in ('', '_[noisy]_'):
        blobs[pre + 'fc6_w'] = normal((4096, k6), (2.0 / k6) ** 0.5)
        blobs[pre + 'fc6_b'] = torch.zeros((4096,), device=device)
        blobs[pre + 'fc7_w'] = normal((4096, 4096), (2.0 / 4096) ** 0.5)
        blobs[pre + 'fc7_b'] = torch.zeros((4096,), device=device)
    lim = (3.0 / 4096) ** 0.5
    for name in ('fc8c', 'fc8d', 'noisy_fc8c', 'noisy_fc8d'):
        blobs[name + '_w'] = ((torch.rand((num_fg_classes, 4096), generator=g) * 2 - 1) * lim).to(device)
        blobs[name + '_b'] = torch.zeros((num_fg_classes,), device=device)
    return blobs


def skew_blobs(blobs, seed=11, span=6.0, bias_std=0.1, roi_size=7):
    """Non-Kaiming statistics for the parity tests and bench.py's in-run check: every conv / fc
    output channel o of layer l is multiplied by s_l[o] = 2^U(-span, span) (log-uniform over
    2^+-6 by default) and given a non-zero bias N(0, bias_std) * s_l[o]; the NEXT layer's weights
    are divided by s_l along their input channels.  ReLU and max-pool commute with a positive
    per-channel factor, so the logits are those of the un-skewed network with biases (fc8* gets
    the input compensation only) while every activation tensor has per-channel magnitudes
    spread over 12 octaves and every weight row entries spread over up to 24.  That is the
    case a per-tensor / per-row operand scale handles worst.  Returns a new dict (fp32)."""
    import torch
    g = torch.Generator(device='cpu')
    g.manual_seed(seed * 7919 + 1)

    def scale(n):
        return torch.exp2((torch.rand((n,), generator=g) * 2 - 1) * span)

    out = {}
    prev = torch.ones((3,))
    for name, _cin, cout in CONV_SHAPES:
        s = scale(cout)
        out[name + '_w'] = blobs[name + '_w'] * s[:, None, None, None] / prev[None, :, None, None]
        out[name + '_b'] = (blobs[name + '_b'] +
                            torch.randn((cout,), generator=g) * bias_std) * s
        prev = s
    k_in = prev.repeat_interleave(roi_size * roi_size)          # roi_feat index = c*49 + ph*7 + pw
    for pre, p8 in (('', ''), ('_[noisy]_', 'noisy_')):
        s6, s7 = scale(4096), scale(4096)
        out[pre + 'fc6_w'] = blobs[pre + 'fc6_w'] * s6[:, None] / k_in[None, :]
        out[pre + 'fc6_b'] = (blobs[pre + 'fc6_b'] + torch.randn((4096,), generator=g) * bias_std) * s6
        out[pre + 'fc7_w'] = blobs[pre + 'fc7_w'] * s7[:, None] / s6[None, :]
        out[pre + 'fc7_b'] = (blobs[pre + 'fc7_b'] + torch.randn((4096,), generator=g) * bias_std) * s7
        for k in ('fc8c', 'fc8d'):
            w = blobs[p8 + k + '_w']
            out[p8 + k + '_w'] = w / s7[None, :]
            out[p8 + k + '_b'] = blobs[p8 + k + '_b'] + torch.randn((w.shape[0],), generator=g) * bias_std
    return {k: v.float().contiguous() for k, v in out.items()}


def skew_images(data, dark=2.0 ** -12, amplitude=40.0):
    """data [B,3,H,W] fp32 (mean-subtracted) -> the same images plus a smooth low-frequency
    component, with the left third of every image multiplied by `dark` (2^-12): a region whose
    activations sit 12 octaves below the tensor maximum that one operand scale has to serve."""
    b, _c, h, w = data.shape
    yy = np.arange(h, dtype=np.float32)[:, None] / h
    xx = np.arange(w, dtype=np.float32)[None, :] / w
    smooth = amplitude * np.sin(2 * np.pi * 1.5 * xx) * np.cos(2 * np.pi * yy)
    out = data + smooth[None, None].astype(np.float32)
    out[:, :, :, :w // 3] *= np.float32(dark)
    return np.ascontiguousarray(out, dtype=np.float32)
