"""Operator API of the hot path: the reference's Caffe2 operator names as Python callables
over HIP tensors (same argument names and defaults), each backed by hand-written gfx950
kernels through the C ABI (include/naws.h).

Custom ops (reference detectron/ops/*):
  RoIFeatureBoost(+Gradient)              roi_feature_boost_op.cc:8-66, schema :70-101
  RoIIoU                                  roi_iou_op.cu:27-84, roi_iou_op.cc:11-24
  WeightedCrossEntropyWithLogits(+Grad.)  cross_entropy_wsl_op.cc:87-180, schema :242-266
  CrossEntropyWithLogits(+Gradient)       cross_entropy_wsl_op.cc:7-85, schema :214-233
  Stat                                    stat_op.cu:24-78, stat_op.cc:11-23
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
            raise NawsError('ACMWeightDecayMomentumSGDUpdate', _L.ERR_ARG)
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


def FC(X, W, b):
    x2 = X.reshape(X.shape[0], -1)
    return _k.gemm(x2, W, False, True, epilogue=_L.EPI_BIAS, bias=b)


def FCGradient(X, W, dY):
    """-> dW, db, dX."""
    x2 = X.reshape(X.shape[0], -1)
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
