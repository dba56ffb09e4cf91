#!/usr/bin/env python3
"""Numeric golden vectors for the COMPOSITION of the WSDDN outputs, cls_pred, the spatial entropy
gate and the loss seeds, produced by the IMPORTED
reference's own builders

    detectron/modeling/webly_heads.py:32-74    add_webly_outputs  (-> wsl_heads.add_wsl_outputs :23-56)
    detectron/modeling/wsl_heads.py:213-227    add_cls_pred
    detectron/modeling/webly_heads.py:265-391  add_spatial_entropy_weight
    detectron/modeling/webly_heads.py:123-216  add_webly_losses

run against a `model` whose `net.<Op>` methods EXECUTE in numpy float32 instead of appending to a
Caffe2 NetDef.  What the reference contributes is therefore the graph: which operator, on which
blobs, in which order, with which arguments.  What this script has to assume is the arithmetic of
each Caffe2 v1.3 built-in it meets (Caffe2 is not vendored in /root/reference); every one is
elementary, and each assumption is listed in ASSUMPTIONS below and written into the fixture:

The two custom C++ operators on this stretch are NOT executed by the reference here (their
sources need the Caffe2 headers): RoIIoU's output is SUPPLIED (oracle.roi_iou on the same rois -
test infrastructure, bit-checked against hand-derived values in tests/test_oracle_kat.py) and
WeightedCrossEntropyWithLogits is computed by oracle.weighted_ce; blobs downstream of it
(`cross_entropy*`, `loss_cls*`) are stored under the `unpinned_` prefix and say so.

Runs ONLY in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_graph_numeric.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
from make_golden_from_reference import REF, _StubFinder  # noqa: E402

F = np.float32

ASSUMPTIONS = {
    'FC': 'Y = X W^T + b, fp32 (caffe2/operators/fully_connected_op.h; sgemm summation order unspecified)',
    'Softmax': 'axis=1 over a 2-D blob, fp32: subtract the row maximum, exp, divide by the row sum '
               '(caffe2/operators/softmax_utils.cc SoftmaxCPU)',
    'Transpose': 'axes=(1, 0): matrix transpose',
    'Mul/Add/Sub/Div': 'elementwise fp32 with numpy-style broadcasting of the second operand '
                       '(the only broadcasts in this graph: [1] against [1, C], and the legacy '
                       '`broadcast=True` Div of [1, C] by [1, C])',
    'ReduceSum': 'axes=[0], keepdims=True: fp32 column sums, rows added in order',
    'Log': 'fp32 natural log; log(0) = -inf',
    'Scale': 'x * scale in fp32',
    'ReplaceNaN': 'NaN -> value, default value 0.0 (caffe2/operators/replace_nan_op.cc)',
    'MatMul': 'A B, fp32 (sgemm summation order unspecified)',
    'LeakyRelu': 'x >= 0 ? x : alpha x, default alpha 0.01 (caffe2/operators/leaky_relu_op.cc)',
    'Clip': 'min(max(x, min), max); a NaN input is left NaN here - Caffe2\'s CPU (Eigen cwiseMax / '
            'cwiseMin) and CUDA (fminf / fmaxf) kernels differ on NaN, so the one NaN case of the '
            'fixture (`nan_case`) is comparable only under this reading',
    'ConstantFill': 'with an input blob: a tensor of the input\'s shape filled with `value`',
    'Shape/Cast': 'Shape(axes=[0]) -> int64 [1] = R; Cast(to=1) -> fp32',
    'StopGradient/Stat': 'identity / no numeric output',
    'AveragedLoss': 'mean of the input\'s elements',
    'Accuracy': 'fraction of rows whose arg-max equals the int32 label',
    'RoIIoU': 'SUPPLIED: oracle.roi_iou(rois) (custom op, not executed by the reference here)',
    'WeightedCrossEntropyWithLogits': 'UNPINNED: oracle.weighted_ce (custom op, not executed by the '
                                      'reference here)',
}


class NumpyNet(object):
    """Executes the operators the reference builders call, eagerly, on a dict of named blobs."""

    def __init__(self, blobs, num_classes, train, supplied_J, wce):
        self.b = blobs
        self.train = train
        self.num_classes = num_classes
        self.net = self
        self.param_init_net = self
        self.losses, self.metrics = [], []
        self.trace = []
        self._J, self._wce = supplied_J, wce

    # ---- helper-level methods (detectron/modeling/detector.py wrappers)
    def AddLosses(self, l):
        self.losses += l if isinstance(l, list) else [l]

    def AddMetrics(self, m):
        self.metrics += m if isinstance(m, list) else [m]

    def _out(self, name, value):
        name = str(name)
        self.b[name] = value
        return name

    def _rec(self, op, ins, outs):
        self.trace.append([op, [str(i) for i in ins], [str(o) for o in outs]])

    def FC(self, blob_in, blob_out, dim_in, dim_out, **kw):
        x, w, bias = self.b[str(blob_in)], self.b[blob_out + '_w'], self.b[blob_out + '_b']
        assert w.shape == (dim_out, dim_in) and x.dtype == F
        self._rec('FC', [blob_in, blob_out + '_w', blob_out + '_b'], [blob_out])
        return self._out(blob_out, (x @ w.T + bias).astype(F))

    def Softmax(self, blob_in, blob_out, axis=1, **kw):
        x = self.b[str(blob_in)]
        assert x.ndim == 2 and axis == 1
        e = np.exp(x - x.max(axis=1, keepdims=True), dtype=F)
        self._rec('Softmax', [blob_in], [blob_out])
        return self._out(blob_out, (e / e.sum(axis=1, keepdims=True, dtype=F)).astype(F))

    def Transpose(self, blob_in, blob_out, axes=None, **kw):
        self._rec('Transpose', [blob_in], [blob_out])
        return self._out(blob_out, np.ascontiguousarray(np.transpose(self.b[str(blob_in)], axes)))

    def _binary(self, op, fn, ins, out, **kw):
        a, b2 = self.b[str(ins[0])], self.b[str(ins[1])]
        out = out[0] if isinstance(out, list) else out
        self._rec(op, ins, [out])
        with np.errstate(all='ignore'):
            return self._out(out, fn(a, b2).astype(F))

    def Mul(self, ins, out, **kw):
        return self._binary('Mul', np.multiply, ins, out)

    def Add(self, ins, out, **kw):
        return self._binary('Add', np.add, ins, out)

    def Sub(self, ins, out, **kw):
        return self._binary('Sub', np.subtract, ins, out)

    def Div(self, ins, out, **kw):
        return self._binary('Div', np.divide, ins, out)

    def ReduceSum(self, blob_in, blob_out, axes=None, keepdims=True, **kw):
        x = self.b[str(blob_in)]
        assert list(axes) == [0] and keepdims
        s = np.zeros((1, x.shape[1]), F)
        for r in range(x.shape[0]):          # rows added in order, fp32
            s[0] += x[r]
        self._rec('ReduceSum', [blob_in], [blob_out])
        return self._out(blob_out, s)

    def Log(self, blob_in, blob_out, **kw):
        self._rec('Log', [blob_in], [blob_out])
        with np.errstate(all='ignore'):
            return self._out(blob_out, np.log(self.b[str(blob_in)], dtype=F))

    def Scale(self, blob_in, blob_out, scale=1.0, **kw):
        self._rec('Scale', [blob_in], [blob_out])
        return self._out(blob_out, (self.b[str(blob_in)] * F(scale)).astype(F))

    def ReplaceNaN(self, blob_in, blob_out, value=0.0, **kw):
        x = self.b[str(blob_in)].copy()
        x[np.isnan(x)] = F(value)
        self._rec('ReplaceNaN', [blob_in], [blob_out])
        return self._out(blob_out, x)

    def MatMul(self, ins, out, **kw):
        a, b2 = self.b[str(ins[0])], self.b[str(ins[1])]
        self._rec('MatMul', ins, [out])
        return self._out(out, (a @ b2).astype(F))

    def LeakyRelu(self, blob_in, blob_out, alpha=0.01, **kw):
        x = self.b[str(blob_in)]
        self._rec('LeakyRelu', [blob_in], [blob_out])
        return self._out(blob_out, np.where(x >= 0, x, F(alpha) * x).astype(F))

    def Clip(self, blob_in, blob_out, min=None, max=None, **kw):
        x = self.b[str(blob_in)]
        self._rec('Clip', [blob_in], [blob_out])
        return self._out(blob_out, np.clip(x, F(min), F(max)).astype(F))

    def ConstantFill(self, ins, out, value=0.0, **kw):
        src = ins[0] if isinstance(ins, list) else ins
        out = out[0] if isinstance(out, list) else out
        self._rec('ConstantFill', [src], [out])
        return self._out(out, np.full_like(np.asarray(self.b[str(src)], F), F(value)))

    def Shape(self, blob_in, blob_out, axes=None, **kw):
        self._rec('Shape', [blob_in], [blob_out])
        return self._out(blob_out, np.array([self.b[str(blob_in)].shape[a] for a in axes], np.int64))

    def Cast(self, blob_in, blob_out, to=1, **kw):
        assert to == 1
        self._rec('Cast', [blob_in], [blob_out])
        return self._out(blob_out, self.b[str(blob_in)].astype(F))

    def StopGradient(self, blob_in, blob_out, **kw):
        self._rec('StopGradient', [blob_in], [blob_out])
        return self._out(blob_out, self.b[str(blob_in)])

    def Stat(self, ins, outs, **kw):
        self._rec('Stat', ins, outs)
        return outs

    def Split(self, blob_in, outs, split=None, axis=1, **kw):
        x = self.b[str(blob_in)]
        parts = np.split(x, np.cumsum(split)[:-1], axis=axis)
        self._rec('Split', [blob_in], outs)
        for o, p in zip(outs, parts):
            self._out(o, np.ascontiguousarray(p))
        return outs

    def Concat(self, ins, outs, axis=1, **kw):
        self._rec('Concat', ins, outs)
        self._out(outs[0], np.concatenate([self.b[str(i)] for i in ins], axis=axis))
        return outs

    def RoIIoU(self, ins, outs, **kw):
        self._rec('RoIIoU', ins, outs)
        return self._out(outs[0], self._J)

    def WeightedCrossEntropyWithLogits(self, ins, outs, is_mean=True, **kw):
        x, l, w = (self.b[str(i)] for i in ins[:3])
        self._rec('WeightedCrossEntropyWithLogits', ins, outs)
        return self._out(outs[0], np.asarray(self._wce(x, l, w, is_mean), F).reshape(-1))

    def AveragedLoss(self, ins, outs, **kw):
        self._rec('AveragedLoss', ins, outs)
        x = self.b[str(ins[0])]
        return self._out(outs[0], np.asarray(x.mean(dtype=F), F).reshape(()))

    def Accuracy(self, ins, out, **kw):
        p, lab = self.b[str(ins[0])], self.b[str(ins[1])]
        self._rec('Accuracy', ins, [out])
        return self._out(out, np.asarray((p.argmax(axis=1) == lab).mean(), F))


def make_inputs(rng, R, C, dim, nan_case):
    b = {}
    for n in ('drop7', '_[noisy]_drop7'):
        b[n] = np.maximum(rng.standard_normal((R, dim)), 0).astype(F) * F(2.0)
    for n in ('fc8c', 'fc8d', 'noisy_fc8c', 'noisy_fc8d'):
        b[n + '_w'] = (rng.standard_normal((C, dim)) * (0.35 if n.startswith('fc8') else 0.05)).astype(F)
        b[n + '_b'] = (rng.standard_normal((C,)) * 0.1).astype(F)
    # proposals: (batch 0, x1, y1, x2, y2) at input resolution, some fractional, a few nested
    x1 = np.floor(rng.uniform(0, 700, R)); y1 = np.floor(rng.uniform(0, 400, R))
    w = np.exp(rng.uniform(np.log(24), np.log(600), R)); h = np.exp(rng.uniform(np.log(24), np.log(400), R))
    rois = np.stack([np.zeros(R), x1, y1, np.minimum(x1 + w, 999), np.minimum(y1 + h, 599)], 1).astype(F)
    rois[::5, 1:] = rois[::5, 1:] * F(1.171875)
    rois[3, 1:] = rois[2, 1:]                                   # an exact duplicate
    b['rois'] = rois
    lab = np.zeros((1, C), F)
    lab[0, 3] = 1.0
    lab[0, C - 2] = 0.4                                         # a mixup-style fractional label
    b['labels_oh'] = lab
    b['labels_int32'] = np.array([3], np.int32)
    # p = 0 entries: proposals whose det-stream logit of one class is so low that the softmax
    # over proposals underflows to exactly 0 there (E = ReplaceNaN(0 * -inf) = 0 while D > 0)
    b['_p0_rows'] = np.array([5, 11, 17], np.int64)
    b['_p0_class'] = np.int64(7)
    if nan_case:
        # D = 0: an isolated proposal (no other box overlaps it) with p = 0 -> E = 0, D = E = 0,
        # G = 0 / 0: the column's hatE_sum is NaN in the reference graph
        rois[9, 1:] = [5000, 5000, 5050, 5040]
        b['_p0_rows'] = np.array([5, 9, 11, 17], np.int64)
    return b


def run_case(R, C, dim, seed, nan_case, builders, oracle):
    webly_heads, wsl_heads = builders
    rng = np.random.default_rng(seed)
    blobs = make_inputs(rng, R, C, dim, nan_case)
    p0_rows, p0_class = blobs.pop('_p0_rows'), int(blobs.pop('_p0_class'))
    J = oracle.roi_iou(blobs['rois'])
    m = NumpyNet(blobs, C + 1, True, J, oracle.weighted_ce)
    # the det-stream logit of the p = 0 rows is pushed down through a dedicated input feature: the
    # last feature is 0 for every proposal but those rows (300 there) and has weight -1 towards
    # fc8d's class p0_class only, 0 everywhere else -> fc8d[row, p0_class] ~ -300, nothing else moves
    for n in ('drop7', '_[noisy]_drop7'):
        blobs[n][:, dim - 1] = 0
    for n in ('fc8c', 'fc8d', 'noisy_fc8c', 'noisy_fc8d'):
        blobs[n + '_w'][:, dim - 1] = 0
    blobs['fc8d_w'][p0_class, dim - 1] = -1.0
    blobs['drop7'][p0_rows, dim - 1] = 300.0
    inputs = {k: v.copy() for k, v in blobs.items()}
    webly_heads.add_webly_outputs(m, ['drop7', '_[noisy]_drop7'], [dim, dim])
    webly_heads.add_webly_losses(m)
    assert m.losses == ['loss_cls', 'loss_cls_noise'], m.losses
    keep = ['rois_pred', 'rois_pred_noise', 'cls_prob', 'cls_prob_noise', 'rois_pred_hatE_sum',
            'rois_pred_hatE_sum_norm', 'rois_class_weight', 'rois_class_weight_noise',
            'loss_cls_grad', 'loss_cls_noise_grad', 'accuracy_cls', 'accuracy_cls_noise']
    if R <= 64:          # every intermediate at the small size (the fixture stays < 1 MB)
        keep += ['fc8c', 'fc8d', 'noisy_fc8c', 'noisy_fc8d', 'fc8c_noise', 'fc8d_noise',
                 'alpha_cls', 'alpha_det', 'alpha_cls_noise', 'alpha_det_noise',
                 'rois_pred_E', 'rois_pred_D', 'rois_pred_G', 'rois_pred_hatE', 'rois_J']
    out = {'in_' + k: v for k, v in inputs.items()}
    out.update({'out_' + k: blobs[k] for k in keep})
    for k in ('cross_entropy', 'cross_entropy_noise', 'loss_cls', 'loss_cls_noise'):
        out['unpinned_' + k] = blobs[k]
    assert (blobs['rois_pred'][p0_rows, p0_class] == 0).all(), blobs['rois_pred'][p0_rows, p0_class]
    has_nan = bool(np.isnan(blobs['rois_pred_hatE_sum']).any())
    assert has_nan == nan_case, (has_nan, nan_case)
    return out, m.trace, dict(R=R, C=C, dim=dim, seed=seed, nan_case=nan_case,
                              p0_rows=[int(r) for r in p0_rows], p0_class=p0_class)


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, ROOT)
    from oracle import oracle            # RoIIoU supplier + the unpinned WCE (test infrastructure)
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, REF)
    import future.utils
    future.utils.iteritems = lambda d: iter(d.items())
    import yaml
    import detectron.utils.env as envu
    envu.yaml_load = lambda s: yaml.load(s, Loader=yaml.FullLoader)
    from detectron.core import config as rcfg
    assert rcfg.__file__.startswith(REF)
    rcfg.merge_cfg_from_file(os.path.join(REF, 'configs/flickr_voc/na_wsddn_V-16-C5_1x.yaml'))
    rcfg.merge_cfg_from_list(['NUM_GPUS', 1])
    from detectron.modeling import wsl_heads, webly_heads
    for mod in (wsl_heads, webly_heads):
        mod.const_fill = lambda v: ('ConstantFill', {'value': v})
        mod.gauss_fill = lambda s: ('GaussianFill', {'std': s})
    cases = [(64, 20, 32, 101, False), (64, 80, 32, 102, False), (300, 20, 16, 103, False),
             (300, 80, 16, 104, False), (64, 20, 32, 105, True)]
    arrays, meta, trace0 = {}, [], None
    for i, (R, C, dim, seed, nan_case) in enumerate(cases):
        out, trace, info = run_case(R, C, dim, seed, nan_case, (webly_heads, wsl_heads), oracle)
        for k, v in out.items():
            arrays['c%d_%s' % (i, k)] = v
        meta.append(info)
        if trace0 is None:
            trace0 = trace
        assert [t[0] for t in trace] == [t[0] for t in trace0]
        print('case', info, 'loss', float(out['unpinned_loss_cls']), float(out['unpinned_loss_cls_noise']),
              'class_weight_noise max', float(np.nanmax(out['out_rois_class_weight_noise'])))
    np.savez_compressed(os.path.join(HERE, 'reference_graph_numeric.npz'), **arrays)
    with open(os.path.join(HERE, 'reference_graph_numeric.json'), 'w') as f:
        json.dump(dict(cases=meta, assumptions=ASSUMPTIONS, executed_ops=trace0,
                       reference=['detectron/modeling/webly_heads.py:32-74,123-216,265-391',
                                  'detectron/modeling/wsl_heads.py:23-56,213-227,292-302',
                                  'detectron/utils/blob.py:167-173']), f, indent=1)
    print('ops executed per case:', len(trace0),
          '; fixture', os.path.getsize(os.path.join(HERE, 'reference_graph_numeric.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
