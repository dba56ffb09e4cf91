"""Operator API of the hot path: the reference's Caffe2 operator names as Python callables
over HIP tensors (same argument names and defaults), each backed by hand-written gfx950
kernels through the C ABI (include/naws.h).

Custom ops (reference detectron/ops/*):
  RoIFeatureBoost(+Gradient)              roi_feature_boost_op.cc:8-66, schema :70-101
  RoIIoU                                  roi_iou_op.cu:27-84, roi_iou_op.cc:11-24
  WeightedCrossEntropyWithLogits(+Grad.)  cross_entropy_wsl_op.cc:87-180, schema :242-266
  CrossEntropyWithLogits(+Gradient)       cross_entropy_wsl_op.cc:7-85, schema :214-233
  Stat                                    stat_op.cu:24-78, stat_op.cc:11-23
  RoILabel                                roi_label_op.cc:10-123          (WSL.OICR)
  SoftmaxWithLossN(+Gradient)             softmax_with_loss_n_op.cc:152-357
  RoIEntropy, BoxWithNMSLimit (gate form) roi_entropy_op.cu:24-142, webly_heads.py:238-248
  ACMWeightDecayMomentumSGDUpdate         acm_weightdecay_momentum_sgd_op.h:48-112
Caffe2 built-ins used by the path (pytorch v1.3.0 caffe2/operators, restated): RoIPoolF,
Conv, Relu, MaxPool, FC, Dropout, Softmax, Transpose, Add/Sub/Mul/Div (numpy-style
broadcast), ReduceSum, Log, Scale, ReplaceNaN, MatMul, LeakyRelu, Clip, ConstantFill,
Shape/Cast (host), AveragedLoss, Accuracy, StopGradient, Concat/Split (views).

Tensors are NCHW / row-major like the reference blobs.  There is no CPU fallback: CPU
tensors raise TypeError, shape violations raise naws_hip.lib.NawsError where the reference
op would CAFFE_ENFORCE.
"""
import math

import numpy as np
import torch

from naws_hip import lib as _L
from naws_hip import ops as _k

NawsError = _L.NawsError


# --------------------------------------------------------------------------- RoI ops
def RoIPoolF(X, R, pooled_h=1, pooled_w=1, spatial_scale=1.0, sampling_ratio=0):
    """-> (Y [n,C,ph,pw], argmax int32).  sampling_ratio is ignored (detector.py:321-329)."""
    return _k.roi_pool_f(X, R, pooled_h, pooled_w, spatial_scale, layout='NCHW', with_argmax=True)


def RoIFeatureBoost(X, S, out=None):
    return _k.roi_feature_boost(X, S, out=out)


def RoIFeatureBoostGradient(dY, S):
    return _k.roi_feature_boost_grad(dY, S)


def RoIIoU(R):
    return _k.roi_iou(R)


# ------------------------------------------------------------------------------ losses
def WeightedCrossEntropyWithLogits(X, L, W, cpg=None, is_mean=False):
    if X.dim() != 2:
        raise NawsError('WeightedCrossEntropyWithLogits', _L.ERR_SHAPE)
    return _k.weighted_ce(X, L, W, is_mean)[0]


def WeightedCrossEntropyWithLogitsGradient(X, L, W, dY, is_mean=False):
    return _k.weighted_ce_grad(X, L, W, dY.reshape(1), is_mean)


def MinEntropyLoss(X, L, cpg=None):
    """detectron/ops/min_entropy_loss_op.cc:7-45 (schema :105-113): Y[] scalar."""
    return _k.min_entropy_loss(X.contiguous(), L.contiguous())[0]


def MinEntropyLossGradient(X, L, dY):
    """detectron/ops/min_entropy_loss_op.cc:47-98."""
    return _k.min_entropy_loss_grad(X.contiguous(), L.contiguous(), dY.reshape(1))


def CrossEntropyWithLogits(X, L, cpg=None, is_mean=False):
    if X.dim() != 2:
        raise NawsError('CrossEntropyWithLogits', _L.ERR_SHAPE)
    return _k.weighted_ce(X, L, None, is_mean)[0]


def CrossEntropyWithLogitsGradient(X, L, dY, is_mean=False):
    return _k.weighted_ce_grad(X, L, None, dY.reshape(1), is_mean)


def AveragedLoss(X):
    """Mean over all elements; the path only ever averages a scalar (identity)."""
    if X.numel() == 1:
        return X.reshape(())
    return _k.unary(_L.UN_SCALE, _k.reduce_sum_axis0(X.reshape(-1, 1)), 1.0 / X.numel()).reshape(())


def Accuracy(prediction, label):
    """Top-1 accuracy of [N,C] predictions vs int32 labels (host-side metric)."""
    p = prediction.detach().cpu().numpy()
    l = label.detach().cpu().numpy().reshape(-1)
    return float((p.argmax(1) == l).mean())


class Stat(object):
    """Running masked mean printer; state = the op's cur_iter_ / init_ (stat_op.h:17-34)."""

    def __init__(self, display=1280, prefix='', printer=print):
        self.display, self.prefix, self.printer = int(display), prefix, printer
        self.cur_iter, self.init = 0, True
        self.AI = self.AL = None

    def __call__(self, I, L, gpu_id=0):
        if gpu_id != 0:            # the op runs on GPU 0 only (stat_op.cu:26-30)
            return self.AI, self.AL
        if self.AI is None:
            self.AI, self.AL = torch.zeros_like(I), torch.zeros_like(L)
        _k.stat_accumulate(I.contiguous(), L.contiguous(), self.AI, self.AL, self.init)
        self.init = False
        self.cur_iter += 1
        if self.cur_iter % self.display == 0 or self.cur_iter == 1:
            ai, al = self.AI.cpu().numpy().reshape(-1), self.AL.cpu().numpy().reshape(-1)
            with np.errstate(divide='ignore', invalid='ignore'):
                vals = ai / al
            self.printer('\t' + self.prefix + ' Stat #iter_: ' + str(self.cur_iter) +
                         ''.join(' %.2f' % v for v in vals))
            self.init = True
        return self.AI, self.AL


# ------------------------------------------------ OICR refinement / mining gate (f-4)
class RoILabel(object):
    """detectron/ops/roi_label_op.cc:10-123 (schema :133-145, state roi_label_op.h:14-56): pseudo
    labels of the OICR refinement branches.  (S [n,c(+1)], U [n,n], L [1,c][, CW [c]]) ->
    (RL int32 [n], RW [n]).  State = the op's display counters; every `display` calls it prints
    the reference's line.  num_pos / num_neg below n would make the result depend on the
    reference's time-seeded shuffle: rejected (NawsError, UNSUPPORTED)."""

    def __init__(self, display=1280, uuid=0, fg_thresh=0.5, bg_thresh_hi=0.5, bg_thresh_lo=-1.0,
                 num_pos=9999, num_neg=9999, top_k=1, debug_info=False, printer=print):
        self.display, self.uuid, self.printer = int(display), int(uuid), printer
        self.fg_thresh, self.bg_thresh_hi, self.bg_thresh_lo = fg_thresh, bg_thresh_hi, bg_thresh_lo
        self.num_pos, self.num_neg, self.top_k = int(num_pos), int(num_neg), int(top_k)
        self.cur_iter, self.stats = 0, None

    def __call__(self, S, U, L, CW=None):
        if self.stats is None:
            self.stats = torch.zeros((4,), device=S.device, dtype=torch.float32)
        rl, rw = _k.roi_label(S.contiguous(), U.contiguous(), L.contiguous(),
                              None if CW is None else CW.contiguous().reshape(-1),
                              self.fg_thresh, self.bg_thresh_hi, self.bg_thresh_lo, self.top_k,
                              self.num_pos, self.num_neg, stats=self.stats)
        self.cur_iter += 1
        if self.cur_iter % self.display == 0:
            fg, bg, fw, bw = self.stats.cpu().tolist()
            with np.errstate(divide='ignore', invalid='ignore'):
                self.printer('RoILabel %d\tfg_rois: %d\tbg_rois: %d\tfg_weight: %f\tbg_weight: %f' % (
                    self.uuid, int(fg) // self.display, int(bg) // self.display,
                    float(np.float32(fw) / np.float32(fg)), float(np.float32(bw) / np.float32(bg))))
            self.stats.zero_()
        return rl, rw


def SoftmaxWithLossN(X, T, W=None, scale=1.0):
    """detectron/ops/softmax_with_loss_n_op.cc:152-263 (label mode, axis 1): -> (P, loss [])."""
    p, loss = _k.softmax_with_loss_n(X.contiguous(), T.contiguous().reshape(-1),
                                     None if W is None else W.contiguous().reshape(-1), scale)
    return p, loss.reshape(())


def SoftmaxWithLossNGradient(X, T, W, P, d_avg_loss, scale=1.0):
    """detectron/ops/softmax_with_loss_n_op.cc:265-357 -> dX."""
    return _k.softmax_with_loss_n_grad(T.contiguous().reshape(-1),
                                       None if W is None else W.contiguous().reshape(-1),
                                       P.contiguous(), d_avg_loss.reshape(1).contiguous(), scale)


class RoIEntropy(object):
    """detectron/ops/roi_entropy_op.cu:69-142 (state roi_entropy_op.h:14-39): (S [n], C [n]) ->
    E [1, num_classes]; keeps the op's running mean_ and prints it every `display` calls."""

    def __init__(self, display=1280, num_classes=20, rm_bg=True, debug_info=False, printer=print):
        self.display, self.num_classes, self.rm_bg = int(display), int(num_classes), bool(rm_bg)
        self.printer = printer
        self.cur_iter, self.init, self.mean = 0, True, None

    def __call__(self, S, C):
        if self.mean is None:
            self.mean = torch.zeros((self.num_classes,), device=S.device, dtype=torch.float32)
        e = _k.roi_entropy(S.contiguous().reshape(-1), C.contiguous().reshape(-1), self.num_classes,
                           self.rm_bg, mean=self.mean, init=self.init)
        self.init = False
        self.cur_iter += 1
        if self.cur_iter % self.display == 0 or self.cur_iter == 1:
            self.printer('RoIEntropy #iter_: %d' % self.cur_iter)
            self.printer(''.join('  %g' % v for v in self.mean.cpu().tolist()))
            self.init = True
        return e


def BoxWithNMSLimit(scores, boxes, score_thresh=0.05, nms=0.3, detections_per_im=100):
    """Caffe2 BoxWithNMSLimit as webly_heads.py:238-248 uses it (hard NMS, one image): scores
    [n, K] (column 0 = background), boxes [n, 4K] class-tiled -> (scores_nms [m], boxes_nms [m,4],
    classes_nms [m] float): per class 1..K-1 the candidates above score_thresh after greedy NMS
    (naws_nms_sorted_fwd: all classes in one launch pair), in descending score order, classes
    concatenated; the image-wide detections_per_im cut keeps exactly the detections_per_im highest
    scores."""
    n, k = scores.shape
    fg = scores[:, 1:].contiguous()
    keep = _k.nms_per_class(boxes.contiguous(), fg, score_thresh, nms)            # [K-1, n] bool
    st = fg.t()
    key = torch.where(keep, st, torch.full_like(st, -float('inf')))
    order = torch.sort(key, dim=1, descending=True, stable=True).indices           # kept ones first
    cnt = keep.sum(dim=1)
    sel = torch.arange(n, device=scores.device)[None, :] < cnt[:, None]            # [K-1, n]
    cls = torch.arange(1, k, device=scores.device)[:, None].expand(k - 1, n)
    rows = order[sel]
    classes = cls[sel]
    s = st[classes - 1, rows]
    b = boxes.view(n, k, 4)[rows, classes]
    if detections_per_im > 0 and s.numel() > detections_per_im:
        # the op sorts all kept (class, row) entries by score and keeps exactly the first
        # detections_per_im of them (ties beyond the limit are dropped; which of equal scores
        # survives is unspecified upstream - std::sort - and is the earlier (class, row) here)
        top = torch.sort(s, descending=True, stable=True).indices[:detections_per_im]
        m = torch.zeros_like(s, dtype=torch.bool)
        m[top] = True
        s, b, classes = s[m], b[m], classes[m]
    return s.contiguous(), b.contiguous(), classes.to(torch.float32)


def Mean(*Xs):
    """Elementwise mean of equally shaped blobs (wsl_heads.py:156: the OICR test-time ensemble)."""
    acc = Xs[0]
    for x in Xs[1:]:
        acc = _bin(_L.BIN_ADD, acc, x).view(Xs[0].shape)
    return _k.unary(_L.UN_SCALE, acc.contiguous(), 1.0 / len(Xs)).view(Xs[0].shape)


def Max(A, B):
    """Elementwise maximum (webly_heads.py:259)."""
    return torch.maximum(A, B)


def Tile(X, tiles=1, axis=1):
    return X.repeat_interleave(1, dim=axis).repeat(*[tiles if d == axis else 1 for d in range(X.dim())])


class ACMWeightDecayMomentumSGDUpdate(object):
    """One instance per parameter blob, state = iter_count_.  In place on momentum / param /
    acmgrad like the reference op; grad is read-only."""

    def __init__(self, momentum=0.0, nesterov=0, weight_decay=0.0, iter_size=1, gpu_num=1,
                 lr_mult=1.0):
        self.momentum, self.nesterov = float(momentum), int(nesterov)
        self.weight_decay, self.iter_size = float(weight_decay), int(iter_size)
        self.gpu_num, self.lr_mult = int(gpu_num), float(lr_mult)
        self.iter_count = 0
        self._tables = None

    def __call__(self, grad, momentum, lr, param, acmgrad):
        if lr.numel() != 1 or grad.numel() != momentum.numel() or grad.numel() != acmgrad.numel():
            raise NawsError('ACMWeightDecayMomentumSGDUpdate', _L.ERR_SHAPE)
        n = param.numel()
        if n % 4 != 0:
            # the fused kernel works on float4s: blobs such as the 21-entry cls_score biases go
            # through zero-padded copies (elementwise op: the padding never mixes in)
            n4 = (n + 3) // 4 * 4
            pads = [torch.zeros((n4,), device=param.device, dtype=torch.float32) for _ in range(4)]
            for dst, src in zip(pads, (grad, momentum, param, acmgrad)):
                dst[:n] = src.reshape(-1)
            self.__call__(pads[0], pads[1], lr, pads[2], pads[3])
            for dst, src in zip((momentum, param, acmgrad), pads[1:]):
                dst.reshape(-1).copy_(src[:n])
            return grad, momentum, param, acmgrad
        if self._tables is None or self._tables[0].item() != n:
            d = param.device
            self._tables = (torch.tensor([n], dtype=torch.int64, device=d),
                            torch.tensor([self.lr_mult], dtype=torch.float32, device=d),
                            torch.tensor([self.weight_decay], dtype=torch.float32, device=d))
        _k.acm_sgd_update(grad.reshape(-1), momentum.reshape(-1), lr, param.reshape(-1),
                          acmgrad.reshape(-1), *self._tables, self.momentum, self.nesterov,
                          self.iter_size, self.gpu_num, self.iter_count)
        self.iter_count += 1
        return grad, momentum, param, acmgrad


# ------------------------------------------------------------------- Caffe2 built-ins
def Conv(X, W, b, kernel=3, pad=1, stride=1, dilation=1, relu=False):
    """NCHW 3x3 stride-1 convolution (the only shape on the path).  `relu` fuses the
    following in-place Relu."""
    if kernel != 3 or stride != 1 or pad != dilation:
        raise NawsError('Conv', _L.ERR_UNSUPPORTED)
    cin = X.shape[1]
    if cin == 3:
        y = _k.conv3x3_c3_nchw_to_nhwc(X, W, b, relu)
    else:
        y = _k.conv3x3_nhwc(_k.nchw_to_nhwc(X), _k.conv3x3_pack_weight(W), b, dilation, relu)
    return _k.nhwc_to_nchw(y)


def Relu(X, out=None):
    return _k.unary(_L.UN_RELU, X, out=out)


def ReluGradient(Y, dY):
    """dX = dY where Y > 0."""
    return _k.binary(_L.BIN_GATE_POS, dY.reshape(Y.shape[0], -1).contiguous(),
                     Y.reshape(Y.shape[0], -1).contiguous()).view(Y.shape)


def MaxPool(X, kernel=2, pad=0, stride=2):
    if kernel != 2 or pad != 0:
        raise NawsError('MaxPool', _L.ERR_UNSUPPORTED)
    return _k.nhwc_to_nchw(_k.maxpool2x2_nhwc(_k.nchw_to_nhwc(X), stride))


def _pad_rows4(t):
    """[n, k] -> [n4, k] with n4 = n rounded up to 4 (zero rows): the MFMA GEMM's 16-byte loads
    want output widths that are multiples of 4 (cls_score has num_classes = 21 / 81 outputs)."""
    n = t.shape[0]
    n4 = (n + 3) // 4 * 4
    if n4 == n:
        return t
    out = torch.zeros((n4,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
    out[:n] = t
    return out


def FC(X, W, b):
    x2 = X.reshape(X.shape[0], -1)
    n = W.shape[0]
    if n % 4 == 0:
        return _k.gemm(x2, W, False, True, epilogue=_L.EPI_BIAS, bias=b)
    return _k.gemm(x2, _pad_rows4(W), False, True, epilogue=_L.EPI_BIAS,
                   bias=_pad_rows4(b))[:, :n].contiguous()


def FCGradient(X, W, dY):
    """-> dW, db, dX."""
    x2 = X.reshape(X.shape[0], -1)
    n = W.shape[0]
    if n % 4 != 0:
        n4 = (n + 3) // 4 * 4
        dYp = torch.zeros((dY.shape[0], n4), device=dY.device, dtype=dY.dtype)
        dYp[:, :n] = dY
        dW = _k.gemm(dYp, x2, True, False)[:n].contiguous()
        db = _k.colsum(dYp)[:n].contiguous()
        dX = _k.gemm(dYp, _pad_rows4(W), False, False)
        return dW, db, dX.view(X.shape)
    dW = _k.gemm(dY, x2, True, False)
    db = _k.colsum(dY)
    dX = _k.gemm(dY, W, False, False)
    return dW, db, dX.view(X.shape)


def Dropout(X, ratio=0.5, is_test=False, seed=0):
    """-> (Y, mask).  Y = X * mask / (1 - ratio); mask from the counter-based generator."""
    if is_test or ratio <= 0:
        return X, None
    mask = _k.dropout_mask(seed, ratio, X.numel(), X.device).view(X.shape)
    y = _k.binary(_L.BIN_MUL, X.reshape(X.shape[0], -1), mask.reshape(X.shape[0], -1))
    return _k.unary(_L.UN_SCALE, y, 1.0 / (1.0 - ratio)).view(X.shape), mask


def DropoutGradient(dY, mask, ratio=0.5):
    d = _k.binary(_L.BIN_MUL, dY.reshape(dY.shape[0], -1), mask.reshape(dY.shape[0], -1))
    return _k.unary(_L.UN_SCALE, d, 1.0 / (1.0 - ratio)).view(dY.shape)


def Softmax(X, axis=1):
    if X.dim() != 2 or axis != 1:
        raise NawsError('Softmax', _L.ERR_UNSUPPORTED)
    return _k.softmax_rows(X.contiguous())


def SoftmaxGradient(Y, dY):
    return _k.softmax_rows_grad(Y, dY.contiguous())


def Transpose(X, axes=(1, 0)):
    if X.dim() != 2 or tuple(axes) != (1, 0):
        raise NawsError('Transpose', _L.ERR_UNSUPPORTED)
    return _k.transpose2d(X.contiguous())


def _bin(op, A, B):
    a2 = A.reshape(1, 1) if A.dim() == 0 else (A.reshape(1, -1) if A.dim() == 1 else A)
    b2 = B.reshape(1, 1) if B.dim() == 0 else (B.reshape(1, -1) if B.dim() == 1 else B)
    return _k.binary(op, a2.contiguous(), b2.contiguous())


def Add(A, B, broadcast=True):
    return _bin(_L.BIN_ADD, A, B)


def Sub(A, B, broadcast=True):
    return _bin(_L.BIN_SUB, A, B)


def Mul(A, B, broadcast=True):
    return _bin(_L.BIN_MUL, A, B)


def Div(A, B, broadcast=True):
    return _bin(_L.BIN_DIV, A, B)


def ReduceSum(X, axes=(0,), keepdims=True):
    if X.dim() != 2 or tuple(axes) != (0,):
        raise NawsError('ReduceSum', _L.ERR_UNSUPPORTED)
    y = _k.reduce_sum_axis0(X.contiguous())
    return y if keepdims else y.reshape(-1)


def Log(X):
    return _k.unary(_L.UN_LOG, X.contiguous())


def Scale(X, scale=1.0):
    return _k.unary(_L.UN_SCALE, X.contiguous(), scale)


def ReplaceNaN(X, value=0.0):
    return _k.unary(_L.UN_REPLACE_NAN, X.contiguous(), value)


def LeakyRelu(X, alpha=0.01):
    return _k.unary(_L.UN_LEAKY_RELU, X.contiguous(), alpha)


def Clip(X, min=-3.4028234e38, max=3.4028234e38):
    return _k.unary(_L.UN_CLIP, X.contiguous(), min, max)


def MatMul(A, B):
    """[M,K] x [K,N].  Operand dims that are not multiples of 4 are zero-padded (the MFMA
    GEMM uses 16-byte loads)."""
    m, k = A.shape
    n = B.shape[1]
    kp, np_ = (k + 3) // 4 * 4, (n + 3) // 4 * 4
    if kp != k or np_ != n:
        A2 = torch.zeros((m, kp), device=A.device)
        A2[:, :k] = A
        B2 = torch.zeros((kp, np_), device=A.device)
        B2[:k, :n] = B
        return _k.gemm(A2, B2)[:, :n].contiguous()
    return _k.gemm(A.contiguous(), B.contiguous())


def ConstantFill(like=None, shape=None, value=0.0, device=None):
    if like is not None:
        return torch.full_like(like, value)
    return torch.full(tuple(shape), value, device=device)
