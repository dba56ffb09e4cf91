"""CPU oracle for the NA-fWebSOD hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker / the timed CPU baseline.  The product
path (na-fwebsod_amd/) never imports it.

Two layers:
  * the custom operators: plain-C restatement in naws_oracle.c (ctypes), each C
    function citing the reference file:line it follows;
  * the Caffe2 built-ins on the path (Conv, MaxPool, FC, Relu, Dropout): these
    live in the un-vendored third-party dependency pytorch v1.3.0
    (caffe2/operators/*, pinned only in the reference's README.md:34-42), so
    they are restated through torch-CPU fp32 ops with the Caffe2 defaults
    (SURVEY.md §8c): legacy floor pooling, FC = X W^T + b with W [out,in],
    Dropout scale 1/(1-ratio).

PARITY UNPINNED by the reference's own tests (it has none for this path); see
oracle/README.md for what pins the oracle instead.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, 'build', 'libnaws_oracle.so')

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])


def _lib():
    if not os.path.exists(_LIB):
        build()
    lib = C.CDLL(_LIB)
    lib.oracle_acm_sgd.restype = C.c_int64
    return lib


_L = None


def L():
    global _L
    if _L is None:
        _L = _lib()
    return _L


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_fp)


# ---------------------------------------------------------------------------
# custom operators (C restatement)
# ---------------------------------------------------------------------------
def roi_pool_f(x_nchw, rois, pooled_h=7, pooled_w=7, spatial_scale=0.125):
    x, xp = _f(x_nchw)
    r, rp = _f(rois)
    n, c, h, w = x.shape
    R = r.shape[0]
    y = np.empty((R, c, pooled_h, pooled_w), np.float32)
    am = np.empty((R, c, pooled_h, pooled_w), np.int32)
    L().oracle_roi_pool_f(xp, n, c, h, w, rp, R, pooled_h, pooled_w, C.c_float(spatial_scale),
                          y.ctypes.data_as(_fp), am.ctypes.data_as(_ip))
    return y, am


def roi_feature_boost(x, s):
    x, xp = _f(x)
    s, sp = _f(s)
    R = x.shape[0]
    F = x.size // max(R, 1)
    y = np.empty_like(x)
    L().oracle_roi_feature_boost(xp, sp, R, F, y.ctypes.data_as(_fp))
    return y


def roi_iou(rois):
    r, rp = _f(rois)
    n = r.shape[0]
    j = np.empty((n, n), np.float32)
    L().oracle_roi_iou(rp, n, j.ctypes.data_as(_fp))
    return j


def weighted_ce(x, l, w, is_mean):
    x, xp = _f(x)
    l, lp = _f(l)
    wp = None
    if w is not None:
        w, wp = _f(w)
    y = np.zeros((1,), np.float32)
    L().oracle_wce_fwd(xp, lp, wp, x.shape[0], x.shape[1], int(is_mean), y.ctypes.data_as(_fp))
    return y[0]


def weighted_ce_grad(x, l, w, dy, is_mean):
    x, xp = _f(x)
    l, lp = _f(l)
    wp = None
    if w is not None:
        w, wp = _f(w)
    dy, dyp = _f(np.reshape(dy, (1,)))
    dx = np.empty_like(x)
    L().oracle_wce_bwd(xp, lp, wp, dyp, x.shape[0], x.shape[1], int(is_mean),
                       dx.ctypes.data_as(_fp))
    return dx


def acm_sgd(grad, mom, lr, param, acm, momentum, nesterov, weight_decay, iter_size, gpu_num,
            lr_mult, iter_count):
    """In place on mom/param/acm (float32 C-contiguous numpy arrays). Returns new iter_count."""
    for a in (grad, mom, param, acm):
        assert a.dtype == np.float32 and a.flags['C_CONTIGUOUS']
    lr_a, lrp = _f(np.reshape(lr, (1,)))
    return L().oracle_acm_sgd(grad.ctypes.data_as(_fp), mom.ctypes.data_as(_fp), lrp,
                              param.ctypes.data_as(_fp), acm.ctypes.data_as(_fp),
                              C.c_int64(grad.size), C.c_float(momentum), int(nesterov),
                              C.c_float(weight_decay), int(iter_size), int(gpu_num),
                              C.c_float(lr_mult), C.c_int64(iter_count))


def stat(i, l, ai, al, init):
    i, ip_ = _f(i)
    l, lp = _f(l)
    L().oracle_stat(ip_, lp, i.size, int(init), ai.ctypes.data_as(_fp), al.ctypes.data_as(_fp))


def wsddn_outputs(fc8c, fc8d, noisy_fc8c=None, noisy_fc8d=None):
    """One image. -> alpha_cls, alpha_det, rois_pred [R,C], cls_prob [1,C]."""
    a, ap = _f(fc8c)
    b, bp = _f(fc8d)
    np_, dp_ = None, None
    if noisy_fc8c is not None:
        nc, np_ = _f(noisy_fc8c)
        nd, dp_ = _f(noisy_fc8d)
    R, Cc = a.shape
    ac = np.empty((R, Cc), np.float32)
    ad = np.empty((R, Cc), np.float32)
    rp = np.empty((R, Cc), np.float32)
    cp = np.empty((1, Cc), np.float32)
    L().oracle_wsddn_outputs_fwd(ap, bp, np_, dp_, R, Cc, ac.ctypes.data_as(_fp),
                                 ad.ctypes.data_as(_fp), rp.ctypes.data_as(_fp),
                                 cp.ctypes.data_as(_fp))
    return ac, ad, rp, cp


def wsddn_outputs_grad(alpha_cls, alpha_det, d_cls_prob):
    """One image, one branch. -> dzc, dzd [R,C] w.r.t. the branch's (summed) logits."""
    ac, acp = _f(alpha_cls)
    ad, adp = _f(alpha_det)
    g, gp = _f(np.reshape(d_cls_prob, (-1,)))
    R, Cc = ac.shape
    dzc = np.empty((R, Cc), np.float32)
    dzd = np.empty((R, Cc), np.float32)
    L().oracle_wsddn_outputs_bwd(acp, adp, gp, R, Cc, dzc.ctypes.data_as(_fp),
                                 dzd.ctypes.data_as(_fp))
    return dzc, dzd


def entropy_gate(rois, rois_pred, cls_prob, labels_oh):
    """One image. -> class_weight, class_weight_noise, hatE_sum, hatE_sum_norm, each [1,C]."""
    r, rp = _f(rois)
    p, pp = _f(rois_pred)
    y, yp = _f(np.reshape(cls_prob, (-1,)))
    l, lp = _f(np.reshape(labels_oh, (-1,)))
    R, Cc = p.shape
    outs = [np.empty((1, Cc), np.float32) for _ in range(4)]
    L().oracle_entropy_gate(rp, pp, yp, lp, R, Cc, *[o.ctypes.data_as(_fp) for o in outs])
    return tuple(outs)


# ---------------------------------------------------------------------------
# Caffe2 built-ins through torch-CPU (third-party restatement)
# ---------------------------------------------------------------------------
VGG16_LAYERS = [
    # name, cin, cout, dilation   ('P2' = MaxPool k2 s2, 'P1' = MaxPool k2 s1)
    ('conv1_1', 3, 64, 1), ('conv1_2', 64, 64, 1), 'P2',
    ('conv2_1', 64, 128, 1), ('conv2_2', 128, 128, 1), 'P2',
    ('conv3_1', 128, 256, 1), ('conv3_2', 256, 256, 1), ('conv3_3', 256, 256, 1), 'P2',
    ('conv4_1', 256, 512, 1), ('conv4_2', 512, 512, 1), ('conv4_3', 512, 512, 1), 'P1',
    ('conv5_1', 512, 512, 2), ('conv5_2', 512, 512, 2), ('conv5_3', 512, 512, 2),
]


def min_entropy_loss(x, l):
    """ref: detectron/ops/min_entropy_loss_op.cc:7-45."""
    x, xp = _f(x)
    l, lp = _f(l)
    L().oracle_min_entropy_fwd.restype = C.c_float
    return np.float32(L().oracle_min_entropy_fwd(xp, lp, x.shape[0], x.shape[1]))


def min_entropy_loss_grad(x, l, dy):
    """ref: detectron/ops/min_entropy_loss_op.cc:47-98."""
    x, xp = _f(x)
    l, lp = _f(l)
    dx = np.zeros_like(x)
    L().oracle_min_entropy_bwd(xp, lp, C.c_float(float(dy)), x.shape[0], x.shape[1],
                               dx.ctypes.data_as(_fp))
    return dx


def nms(dets, thresh):
    """Greedy NMS on [n,5] (x1,y1,x2,y2,score) -> kept indices, ascending
    (ref: detectron/utils/cython_nms.pyx:36-87).  Visiting order = stable descending score."""
    dets, dp = _f(dets)
    n = dets.shape[0]
    if n == 0:
        return np.zeros((0,), np.int64)
    order = np.ascontiguousarray(np.argsort(-dets[:, 4], kind='stable').astype(np.int64))
    sup = np.zeros((n,), np.int32)
    L().oracle_nms(dp, order.ctypes.data_as(C.c_void_p), n, C.c_float(thresh),
                   sup.ctypes.data_as(_ip))
    return np.where(sup == 0)[0]


def soft_nms(dets, sigma=0.5, overlap_thresh=0.3, score_thresh=0.001, method='linear'):
    """Soft-NMS (ref: detectron/utils/boxes.py:321-338 -> cython_nms.pyx:98-203): [n,5] float32 ->
    (dets_out [m,5], keep [m] original indices), in the reference's output order."""
    dets, dp = _f(dets)
    n = dets.shape[0]
    if n == 0:
        return dets, []
    out = np.empty((n, 5), np.float32)
    inds = np.empty((n,), np.int64)
    m = L().oracle_soft_nms(dp, n, C.c_float(sigma), C.c_float(overlap_thresh),
                            C.c_float(score_thresh), {'hard': 0, 'linear': 1, 'gaussian': 2}[method],
                            out.ctypes.data_as(_fp), inds.ctypes.data_as(C.c_void_p))
    return out[:m].copy(), inds[:m].tolist()


def roi_label(S, U, L_, CW=None, fg_thresh=0.5, bg_thresh_hi=0.5, bg_thresh_lo=-1.0, top_k=1,
              num_pos=9999, num_neg=9999, stats=None):
    """ref: detectron/ops/roi_label_op.cc:10-123 (see naws_oracle.c).  -> (RL int32 [n], RW [n])."""
    S, sp = _f(S)
    U, up = _f(U)
    Lh, lp = _f(np.reshape(L_, (-1,)))
    cwp = None
    if CW is not None:
        CW, cwp = _f(np.reshape(CW, (-1,)))
    n, cs = S.shape
    rl = np.empty((n,), np.int32)
    rw = np.empty((n,), np.float32)
    st = stats if stats is not None else np.zeros((4,), np.float32)
    rc = L().oracle_roi_label(sp, up, lp, cwp, n, cs, Lh.size, C.c_float(fg_thresh),
                              C.c_float(bg_thresh_hi), C.c_float(bg_thresh_lo), int(top_k),
                              int(num_pos), int(num_neg), rl.ctypes.data_as(_ip),
                              rw.ctypes.data_as(_fp), st.ctypes.data_as(_fp))
    if rc != 0:
        raise ValueError('binding num_pos / num_neg caps depend on the reference\'s time-seeded '
                         'shuffle: not reproducible')
    return rl, rw


def softmax_with_loss_n(X, T, W=None, scale=1.0):
    """ref: detectron/ops/softmax_with_loss_n_op.cc:152-263.  -> (P [N,D], loss)."""
    X, xp = _f(X)
    T = np.ascontiguousarray(T, dtype=np.int32)
    wp = None
    if W is not None:
        W, wp = _f(W)
    n, d = X.shape
    P = np.empty_like(X)
    loss = np.zeros((1,), np.float32)
    rc = L().oracle_softmax_with_loss_n_fwd(xp, T.ctypes.data_as(_ip), wp, n, d, C.c_float(scale),
                                            P.ctypes.data_as(_fp), loss.ctypes.data_as(_fp))
    if rc != 0:
        raise ValueError('Label seems incorrect: label value larger than number of classes')
    return P, loss[0]


def softmax_with_loss_n_grad(T, W, P, dloss, scale=1.0):
    """ref: detectron/ops/softmax_with_loss_n_op.cc:265-357.  -> dX [N,D]."""
    P, pp = _f(P)
    T = np.ascontiguousarray(T, dtype=np.int32)
    wp = None
    if W is not None:
        W, wp = _f(W)
    n, d = P.shape
    dX = np.empty_like(P)
    L().oracle_softmax_with_loss_n_bwd(T.ctypes.data_as(_ip), wp, pp, C.c_float(float(dloss)), n, d,
                                       C.c_float(scale), dX.ctypes.data_as(_fp))
    return dX


def roi_entropy(S, Cls, num_classes, rm_bg=True):
    """ref: detectron/ops/roi_entropy_op.cu:24-112.  -> E [1, num_classes]."""
    S, sp = _f(np.reshape(S, (-1,)))
    Cl, cp = _f(np.reshape(Cls, (-1,)))
    E = np.empty((1, num_classes), np.float32)
    L().oracle_roi_entropy(sp, cp, S.size, int(num_classes), int(bool(rm_bg)), E.ctypes.data_as(_fp))
    return E


def box_with_nms_limit(scores, boxes, score_thresh=0.05, nms_thresh=0.3, detections_per_im=100):
    """Caffe2 BoxWithNMSLimit in the form webly_heads.py:238-248 uses it (hard NMS, no soft-nms, one
    image): scores [n, K] (column 0 = background, skipped), boxes [n, 4K] class-tiled.  For
    j = 1..K-1: candidates scores[:, j] > score_thresh, greedy NMS (utils/cython_nms.pyx
    arithmetic), kept in descending score order; the per-class results are concatenated in class
    order; when more than detections_per_im remain, exactly the detections_per_im highest-scored
    entries are kept.  Third-party (pytorch v1.3.0 caffe2/operators/box_with_nms_limit_op.cc,
    un-vendored): restated from its published algorithm, PARITY UNPINNED.
    -> (scores_nms [m], boxes_nms [m,4], classes_nms [m] float)."""
    scores = np.asarray(scores, np.float32)
    boxes = np.asarray(boxes, np.float32)
    k = scores.shape[1]
    out_s, out_b, out_c = [], [], []
    for j in range(1, k):
        inds = np.where(scores[:, j] > score_thresh)[0]
        dets = np.hstack([boxes[inds, 4 * j:4 * j + 4], scores[inds, j:j + 1]]).astype(np.float32)
        keep = nms(dets, nms_thresh)
        keep = keep[np.argsort(-dets[keep, 4], kind='stable')]
        out_s.append(dets[keep, 4]); out_b.append(dets[keep, :4])
        out_c.append(np.full((keep.size,), j, np.float32))
    s, b, c = np.concatenate(out_s), np.concatenate(out_b), np.concatenate(out_c)
    if detections_per_im > 0 and s.size > detections_per_im:
        # the op sorts every kept (class, index) entry by score and keeps exactly the first
        # detections_per_im; among equal scores its std::sort is unspecified - first (class, row)
        # wins here
        top = np.argsort(-s, kind='stable')[:detections_per_im]
        m = np.zeros(s.shape, bool)
        m[top] = True
        s, b, c = s[m], b[m], c[m]
    return s, b.reshape(-1, 4), c


def resize_bilinear_cv2(im, im_scale):
    """cv2.resize(im, None, None, fx=im_scale, fy=im_scale, interpolation=cv2.INTER_LINEAR) for a
    float32 HxWxC image.  OpenCV is an un-vendored dependency of the reference
    (requirements.txt: opencv-python>=3.2, no pin) and is not installed here: this restates the
    published algorithm of imgproc/resize.cpp (float path): dsize = cvRound(ssize*f); scale =
    1/f; per destination index fx = float((dx+0.5)*scale-0.5), sx = floor(fx), fx -= sx, taps
    clamped at both edges with weight 1; horizontal pass, then vertical pass.  PARITY UNPINNED
    (no cv2 to check against); used at detectron/utils/blob.py:123-130."""
    im = np.asarray(im, np.float32)
    h, w = im.shape[:2]
    oh, ow = int(np.round(h * im_scale)), int(np.round(w * im_scale))
    inv = 1.0 / float(im_scale)

    def taps(n_dst, n_src):
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * inv - 0.5).astype(np.float32)
        s0 = np.floor(f).astype(np.int64)
        f = (f - s0.astype(np.float32)).astype(np.float32)
        lo = s0 < 0
        f[lo], s0[lo] = 0.0, 0
        hi = s0 >= n_src - 1
        f[hi], s0[hi] = 0.0, n_src - 1
        return s0, np.minimum(s0 + 1, n_src - 1), (np.float32(1) - f).astype(np.float32), f
    sx, sx1, a0, a1 = taps(ow, w)
    sy, sy1, b0, b1 = taps(oh, h)
    a0, a1 = a0[None, :, None], a1[None, :, None]
    rows = (im[:, sx, :] * a0 + im[:, sx1, :] * a1).astype(np.float32)        # horizontal
    return (rows[sy] * b0[:, None, None] + rows[sy1] * b1[:, None, None]).astype(np.float32)


def _cvround(x):
    return np.rint(x).astype(np.int64)       # cvRound: round half to even


def bgr2hsv_u8(im):
    """cv2.cvtColor(im, cv2.COLOR_BGR2HSV) for uint8 (H in [0,180), S,V in [0,255]).
    OpenCV is un-vendored and un-pinned in the reference; this restates the published 8-bit
    algorithm (imgproc color_hsv: RGB2HSV_b, hsv_shift = 12, division tables
    sdiv[v] = cvRound((255<<12)/v), hdiv[d] = cvRound((180<<12)/(6 d))).  PARITY UNPINNED.
    Used at detectron/roi_data/minibatch_wsl.py:128."""
    im = np.asarray(im, np.uint8).astype(np.int64)
    b, g, r = im[..., 0], im[..., 1], im[..., 2]
    v = np.maximum(np.maximum(b, g), r)
    diff = v - np.minimum(np.minimum(b, g), r)
    idx = np.arange(256, dtype=np.float64)
    with np.errstate(divide='ignore'):
        sdiv = np.where(idx > 0, _cvround((255 << 12) / np.maximum(idx, 1)), 0)
        hdiv = np.where(idx > 0, _cvround((180 << 12) / (6.0 * np.maximum(idx, 1))), 0)
    s = (diff * sdiv[v] + (1 << 11)) >> 12
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * hdiv[diff] + (1 << 11)) >> 12            # arithmetic shift (floor) for negatives
    h = h + np.where(h < 0, 180, 0)
    return np.stack([np.clip(h, 0, 255), s, v], -1).astype(np.uint8)


def hsv2bgr_u8(hsv):
    """cv2.cvtColor(hsv, cv2.COLOR_HSV2BGR) for uint8: the 8-bit path converts to float
    (h, s/255, v/255), runs the float HSV2RGB sector formula (hscale = 6/180) and stores
    saturate_cast<uchar>(x*255) (round half to even).  PARITY UNPINNED (see bgr2hsv_u8).
    Used at detectron/roi_data/minibatch_wsl.py:138."""
    hsv = np.asarray(hsv, np.uint8)
    f32 = np.float32
    h = hsv[..., 0].astype(f32) * f32(6.0 / 180.0)
    s = hsv[..., 1].astype(f32) * f32(1.0 / 255.0)
    v = hsv[..., 2].astype(f32) * f32(1.0 / 255.0)
    h = np.where(h >= f32(6), h - f32(6), h).astype(f32)     # h < 180*6/180: at most one wrap
    sector = np.floor(h).astype(np.int64)
    fr = (h - sector.astype(f32)).astype(f32)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    fr = np.where(bad, f32(0), fr).astype(f32)
    one = f32(1)
    tab = np.stack([v, (v * (one - s)).astype(f32), (v * (one - (s * fr).astype(f32))).astype(f32),
                    (v * (one - (s * (one - fr)).astype(f32))).astype(f32)], -1)
    sec = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])
    pick = sec[sector]                                        # [..., 3] -> (b, g, r) table slots
    bgr = np.take_along_axis(tab, pick, -1)
    bgr = np.where((hsv[..., 1] == 0)[..., None], v[..., None], bgr).astype(f32)
    return np.clip(np.rint((bgr * f32(255)).astype(f32)), 0, 255).astype(np.uint8)


def distort_hsv(im_u8, s0, s1):
    """minibatch_wsl.py:127-138: BGR->HSV (uint8), S *= s0, V *= s1 in float32 capped at 255,
    back to uint8 by truncation, HSV->BGR."""
    hsv = bgr2hsv_u8(im_u8).astype(np.float32)
    hsv[..., 1] = np.minimum(np.float32(s0) * hsv[..., 1], 255)
    hsv[..., 2] = np.minimum(np.float32(s1) * hsv[..., 2], 255)
    return hsv2bgr_u8(hsv.astype(np.uint8))


def prep_image(im_u8, im_scale, flip=False, crop=None, means=(0, 0, 0), stds=(1, 1, 1),
               distort=None):
    """Flip, crop, float32, mean/std, resize: minibatch_wsl.py:121-157 + blob.py:100-131.
    -> float32 [oh, ow, 3] (HWC, as prep_im_for_blob returns it)."""
    im = np.asarray(im_u8)
    if flip:
        im = im[:, ::-1, :]
    if distort is not None:
        im = distort_hsv(im, distort[0], distort[1])
    if crop is not None:
        im = im[crop[0]:crop[2] + 1, crop[1]:crop[3] + 1, :]
    im = im.astype(np.float32)
    im = (im - np.asarray(means, np.float32).reshape(1, 1, 3)).astype(np.float32)
    im = (im / np.asarray(stds, np.float32).reshape(1, 1, 3)).astype(np.float32)
    return resize_bilinear_cv2(im, im_scale)


def vgg16_conv5_body(data, blobs, stats=None):
    """ref: detectron/modeling/VGG16.py:9-48 with WSL.DILATION == 2.
    data: torch CPU [N,3,H,W]; blobs: {name_w: [O,I,3,3], name_b: [O]} -> conv5_3 NCHW.
    stats (a dict): receives name + '_rms' = the per-channel RMS of every layer's output BEFORE
    its ReLU - the scale of a channel's dot products, the yardstick of the per-channel parity
    measure (a channel's post-ReLU maximum would not do: a nearly dead channel has a tiny
    maximum but the full rounding error of its sums)."""
    import torch
    import torch.nn.functional as F
    x = data
    for item in VGG16_LAYERS:
        if item == 'P2':
            x = F.max_pool2d(x, 2, 2, 0, ceil_mode=False)
        elif item == 'P1':
            x = F.max_pool2d(x, 2, 1, 0, ceil_mode=False)
        else:
            name, _, _, dil = item
            x = F.conv2d(x, blobs[name + '_w'], blobs[name + '_b'], stride=1, padding=dil,
                         dilation=dil)
            if stats is not None:
                stats[name + '_rms'] = x.double().pow(2).mean(dim=(0, 2, 3)).sqrt().numpy()
            x = F.relu(x)
    return x


def head_forward(roi_feat, blobs, masks, train=True):
    """Both 2-fc branches + the four fc8 layers.
    ref: detectron/modeling/wsl_heads.py:674-679, webly_heads.py:490-498, wsl_heads.py:29-46,
    webly_heads.py:36-55.  masks: dict name -> keep mask (0/1) for drop6, drop7,
    _[noisy]_drop6, _[noisy]_drop7 (Dropout ratio 0.5 -> scale 2), ignored when not train.
    Returns a dict of every intermediate blob (torch CPU tensors)."""
    import torch
    import torch.nn.functional as F
    out = {}
    x = roi_feat.reshape(roi_feat.shape[0], -1)
    for pre in ('', '_[noisy]_'):
        z = F.linear(x, blobs[pre + 'fc6_w'], blobs[pre + 'fc6_b'])
        out[pre + 'fc6_rms'] = z.detach().pow(2).mean(0).sqrt()   # per unit: the scale of its dot product
        h = F.relu(z)
        if train:
            h = h * masks[pre + 'drop6'] * 2.0
        out[pre + 'drop6'] = h
        z = F.linear(h, blobs[pre + 'fc7_w'], blobs[pre + 'fc7_b'])
        out[pre + 'fc7_rms'] = z.detach().pow(2).mean(0).sqrt()
        h = F.relu(z)
        if train:
            h = h * masks[pre + 'drop7'] * 2.0
        out[pre + 'drop7'] = h
    out['fc8c'] = F.linear(out['drop7'], blobs['fc8c_w'], blobs['fc8c_b'])
    out['fc8d'] = F.linear(out['drop7'], blobs['fc8d_w'], blobs['fc8d_b'])
    out['noisy_fc8c'] = F.linear(out['_[noisy]_drop7'], blobs['noisy_fc8c_w'],
                                 blobs['noisy_fc8c_b'])
    out['noisy_fc8d'] = F.linear(out['_[noisy]_drop7'], blobs['noisy_fc8d_w'],
                                 blobs['noisy_fc8d_b'])
    return out


def loss_tail(fc8, rois, labels_oh, is_mean=True):
    """One image: WSDDN outputs, entropy gate, both weighted CE losses and the gradients
    w.r.t. the four fc8 logit matrices (loss gradient seed 1.0 per loss,
    ref: detectron/utils/blob.py:167-173, webly_heads.py:167-197)."""
    fc8c, fc8d, nfc8c, nfc8d = [np.asarray(fc8[k], np.float32)
                                for k in ('fc8c', 'fc8d', 'noisy_fc8c', 'noisy_fc8d')]
    ac, ad, rp, cp = wsddn_outputs(fc8c, fc8d)
    acn, adn, rpn, cpn = wsddn_outputs(fc8c, fc8d, nfc8c, nfc8d)
    cw, cwn, hs, hsn = entropy_gate(rois, rp, cp, labels_oh)
    lab = np.reshape(labels_oh, (1, -1)).astype(np.float32)
    loss = weighted_ce(cp, lab, cw, is_mean)
    loss_n = weighted_ce(cpn, lab, cwn, is_mean)
    one = np.ones((1,), np.float32)
    g = weighted_ce_grad(cp, lab, cw, one, is_mean)
    gn = weighted_ce_grad(cpn, lab, cwn, one, is_mean)
    dzc, dzd = wsddn_outputs_grad(ac, ad, g)
    dzcn, dzdn = wsddn_outputs_grad(acn, adn, gn)
    return dict(alpha_cls=ac, alpha_det=ad, rois_pred=rp, cls_prob=cp,
                alpha_cls_noise=acn, alpha_det_noise=adn, rois_pred_noise=rpn, cls_prob_noise=cpn,
                class_weight=cw, class_weight_noise=cwn, hatE_sum=hs, hatE_sum_norm=hsn,
                loss_cls=loss, loss_cls_noise=loss_n, d_cls_prob=g, d_cls_prob_noise=gn,
                d_fc8c=dzc + dzcn, d_fc8d=dzd + dzdn, d_noisy_fc8c=dzcn, d_noisy_fc8d=dzdn)


def full_forward_backward(blobs, mb, masks, num_fg_classes, is_mean=True, spatial_scale=0.125,
                          roi_size=7, train=True, conv5=None, roi_feat=None, backward=True,
                          timings=None):
    """Whole hot path on the CPU for a minibatch of B images (each image is one reference
    'GPU': per-image softmax-over-proposals / ReduceSum / gate / loss; gradients summed over
    images, exactly what the all-reduce does).
    blobs: torch CPU tensors in the reference layouts; mb: dict of numpy loader blobs;
    masks: dict of 0/1 keep masks [Rt,4096] for drop6, drop7, _[noisy]_drop6, _[noisy]_drop7.
    conv5 / roi_feat: results of an earlier call on the same images / proposals (they do not
    depend on the head parameters), to skip recomputing them; backward=False stops after the
    per-image loss tails.  timings: a dict that receives the wall seconds of each stage
    (conv / roi_pool / head_fwd / loss / head_bwd), accumulated - bench.py's cpu_baseline.
    Returns dict(losses per image, grads per trainable blob, intermediates)."""
    import time
    import torch
    clock = [time.perf_counter()]

    def lap(name):
        now = time.perf_counter()
        if timings is not None:
            timings[name] = timings.get(name, 0.0) + now - clock[0]
        clock[0] = now
    rois = mb['rois']
    argmax = None
    conv_stats = {}
    if conv5 is None:
        data = torch.from_numpy(mb['data'])
        with torch.no_grad():                                # StopGradient: forward only
            conv5 = vgg16_conv5_body(data, blobs, conv_stats if timings is None else None).numpy()
    lap('conv')
    if roi_feat is None:
        pooled, argmax = roi_pool_f(conv5, rois, roi_size, roi_size, spatial_scale)
        roi_feat = roi_feature_boost(pooled, mb['obn_scores'].reshape(-1))
    lap('roi_pool')
    x = torch.from_numpy(roi_feat.reshape(rois.shape[0], -1))
    names = ['fc6_w', 'fc6_b', 'fc7_w', 'fc7_b', '_[noisy]_fc6_w', '_[noisy]_fc6_b',
             '_[noisy]_fc7_w', '_[noisy]_fc7_b', 'fc8c_w', 'fc8c_b', 'fc8d_w', 'fc8d_b',
             'noisy_fc8c_w', 'noisy_fc8c_b', 'noisy_fc8d_w', 'noisy_fc8d_b']
    params = {n: blobs[n].clone().requires_grad_(True) for n in names}
    tmasks = {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in (masks or {}).items()}
    act = head_forward(x, params, tmasks, train=train)
    lap('head_fwd')
    fc8 = {k: act[k].detach().numpy() for k in ('fc8c', 'fc8d', 'noisy_fc8c', 'noisy_fc8d')}
    b = rois[:, 0].astype(np.int64)
    n_img = mb['data'].shape[0]
    tails, dl = [], {k: np.zeros_like(v) for k, v in fc8.items()}
    for i in range(n_img):
        sel = np.where(b == i)[0]
        t = loss_tail({k: v[sel] for k, v in fc8.items()}, rois[sel], mb['labels_oh'][i], is_mean)
        tails.append(t)
        for k in dl:
            dl[k][sel] = t['d_' + k]
    lap('loss')
    res = dict(conv5_3=conv5, roi_feat=roi_feat, roi_argmax=argmax, conv_stats=conv_stats,
               act={k: v.detach().numpy() for k, v in act.items()}, tails=tails, d_logits=dl)
    if backward:
        outs = [act[k] for k in ('fc8c', 'fc8d', 'noisy_fc8c', 'noisy_fc8d')]
        gouts = [torch.from_numpy(dl[k]) for k in ('fc8c', 'fc8d', 'noisy_fc8c', 'noisy_fc8d')]
        grads = torch.autograd.grad(outs, [params[n] for n in names], gouts)
        res['grads'] = {n: g.numpy() for n, g in zip(names, grads)}
        lap('head_bwd')
    return res


def head_float64(roi_feat, rois, labels_oh, blobs, masks, class_weights, is_mean=True, train=True):
    """The ARBITER for gradient comparisons: the two 2-fc branches, the WSDDN outputs, both
    weighted cross entropies and their backward, evaluated in float64 by torch autograd from the
    mathematical definitions (wsl_heads.py:29-56,213-227,674-679, webly_heads.py:36-74,167-197,
    cross_entropy_wsl_op.cc:87-132 without its 1e-20 / 1e4 clamps, which never bind on finite
    probabilities).  Two fp32 evaluations (this oracle's, the HIP path's) of the
    cancellation-heavy softmax backward differ from each other by more than either differs from
    this one; tests bound both distances.  The entropy-gate weights are StopGradient constants
    (webly_heads.py:390-391): `class_weights` = [(class_weight, class_weight_noise)] per image,
    taken from the fp32 evaluation.  -> dict(losses, d_logits [R,4C], grads per blob, act = the
    four dropout outputs), float64."""
    import torch
    f64 = torch.float64
    x = torch.from_numpy(np.asarray(roi_feat, np.float32).reshape(rois.shape[0], -1)).to(f64)
    names = ['fc6_w', 'fc6_b', 'fc7_w', 'fc7_b', '_[noisy]_fc6_w', '_[noisy]_fc6_b',
             '_[noisy]_fc7_w', '_[noisy]_fc7_b', 'fc8c_w', 'fc8c_b', 'fc8d_w', 'fc8d_b',
             'noisy_fc8c_w', 'noisy_fc8c_b', 'noisy_fc8d_w', 'noisy_fc8d_b']
    params = {n: blobs[n].detach().to(f64).requires_grad_(True) for n in names}
    tmasks = {k: torch.from_numpy(np.asarray(v, np.float32)).to(f64)
              for k, v in (masks or {}).items()}
    act = head_forward(x, params, tmasks, train=train)
    for k in ('fc8c', 'fc8d', 'noisy_fc8c', 'noisy_fc8d'):
        act[k].retain_grad()
    b = torch.from_numpy(rois[:, 0].astype(np.int64))
    lab = torch.from_numpy(np.asarray(labels_oh, np.float32)).to(f64)
    n_img, c = lab.shape
    norm = float(c) if is_mean else 1.0
    total, losses, probs = 0.0, [], []
    for i in range(n_img):
        sel = torch.nonzero(b == i).reshape(-1)
        per = []
        for zc, zd, w in ((act['fc8c'], act['fc8d'], class_weights[i][0]),
                          (act['fc8c'] + act['noisy_fc8c'], act['fc8d'] + act['noisy_fc8d'],
                           class_weights[i][1])):
            p = (torch.softmax(zc[sel], 1) * torch.softmax(zd[sel], 0)).sum(0)
            wt = torch.from_numpy(np.asarray(w, np.float32).reshape(-1)).to(f64)
            ce = -(lab[i] * torch.log(p) + (1 - lab[i]) * torch.log(1 - p)) * wt
            per.append(ce.sum() / norm)
            probs.append(p.detach().numpy())
            total = total + per[-1]
        losses.append([float(v.detach()) for v in per])
    total.backward()
    dl = torch.cat([act[k].grad for k in ('fc8c', 'fc8d', 'noisy_fc8c', 'noisy_fc8d')], 1)
    return dict(losses=losses, d_logits=dl.numpy(),
                cls_prob=np.stack(probs[0::2]), cls_prob_noise=np.stack(probs[1::2]),
                logits=torch.cat([act[k].detach() for k in
                                  ('fc8c', 'fc8d', 'noisy_fc8c', 'noisy_fc8d')], 1).numpy(),
                grads={n: params[n].grad.numpy() for n in names},
                act={k: act[k].detach().numpy() for k in ('drop6', '_[noisy]_drop6', 'drop7',
                                                          '_[noisy]_drop7', 'fc6_rms',
                                                          '_[noisy]_fc6_rms', 'fc7_rms',
                                                          '_[noisy]_fc7_rms')})
