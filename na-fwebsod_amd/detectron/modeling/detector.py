"""DetectionModelHelper: records the operator graph the model-builder functions emit and
generates its backward ops — the role Caffe2's CNNModelHelper + core.Net + AddGradientOperators
play for the reference (detectron/modeling/detector.py:43-586).

Same builder-facing surface for the hot path: `model.Conv/Relu/MaxPool/FC/Dropout/Softmax/
Transpose/StopGradient/Accuracy/RoIFeatureTransform`, `model.net.<AnyOp>(inputs, outputs,
**args)`, `model.param_init_net.ConstantFill`, `AddLosses/AddMetrics/TrainableParams`,
`UpdateWorkspaceLr`.  One process drives one GPU, so blob names carry no `gpu_i/` scope.
Execution lives in detectron/core/executor.py (fused MI355X plan or op-by-op).
"""
import collections

from detectron.core.config import cfg

Op = collections.namedtuple('Op', ['type', 'inputs', 'outputs', 'args'])


def _as_list(x):
    if x is None:
        return []
    return [str(b) for b in x] if isinstance(x, (list, tuple)) else [str(x)]


class Net(object):
    """An ordered op list; `net.SomeOp(inputs, outputs, **args)` appends one op."""

    def __init__(self, name):
        self.name = name
        self.ops = []

    def Proto(self):
        return self

    def add(self, op_type, inputs, outputs, args):
        ins, outs = _as_list(inputs), _as_list(outputs if outputs is not None else inputs)
        self.ops.append(Op(op_type, ins, outs, dict(args)))
        return outs[0] if len(outs) == 1 else tuple(outs)

    def __getattr__(self, op_type):
        if op_type.startswith('_'):
            raise AttributeError(op_type)

        def emit(inputs, outputs=None, **args):
            return self.add(op_type, inputs, outputs, args)
        return emit


# op type -> indices of the inputs that receive a gradient (others are treated as constants)
_GRAD_INPUTS = {
    'FC': (0, 1, 2), 'Relu': (0,), 'Dropout': (0,), 'Softmax': (0,), 'Transpose': (0,),
    'Mul': (0, 1), 'Add': (0, 1), 'ReduceSum': (0,), 'AveragedLoss': (0,),
    'WeightedCrossEntropyWithLogits': (0,), 'CrossEntropyWithLogits': (0,),
    'RoIFeatureBoost': (0,), 'MinEntropyLoss': (0,), 'SoftmaxWithLossN': (0,),
}
_NO_GRAD = {'StopGradient', 'RoIIoU', 'Stat', 'Accuracy', 'ConstantFill', 'Shape', 'Cast',
            'DequeueBlobs', 'RoILabel', 'RoIEntropy', 'BoxWithNMSLimit'}


class DetectionModelHelper(object):
    def __init__(self, name='', train=False, num_classes=-1, init_params=False):
        assert num_classes > 0, 'num_classes must be > 0'
        self.name = name
        self.train = train
        self.num_classes = num_classes
        self.init_params = init_params
        self.net = Net(name)
        self.param_init_net = Net(name + '_init')
        self.params, self.weights, self.biases = [], [], []
        self.param_shapes, self.param_inits = {}, {}
        self.param_to_grad = {}
        self.losses, self.metrics = [], []
        self.do_not_update_params, self.gn_params = [], []
        self.roi_data_loader = None
        self.only_build_forward_pass = False
        self.target_gpu_id = 0
        self.grad_ops = []
        self.update_ops = []
        self.allreduce_ops = []
        self.executor = None

    # ---------------------------------------------------------------- parameters
    def _create_param(self, name, shape, init, is_bias):
        if name not in self.param_shapes:
            self.params.append(name)
            (self.biases if is_bias else self.weights).append(name)
            self.param_shapes[name] = tuple(int(s) for s in shape)
            self.param_inits[name] = init
            self.param_init_net.add(init[0], [], [name], dict(init[1], shape=list(shape)))
        return name

    def TrainableParams(self, gpu_id=-1):
        return [p for p in self.params
                if p in self.param_to_grad and p not in self.do_not_update_params]

    # --------------------------------------------------------------- layer helpers
    def Conv(self, blob_in, blob_out, dim_in, dim_out, kernel, weight_init=None, bias_init=None,
             **kwargs):
        w = self._create_param(blob_out + '_w', (dim_out, dim_in, kernel, kernel),
                               weight_init or ('XavierFill', {}), False)
        b = self._create_param(blob_out + '_b', (dim_out,),
                               bias_init or ('ConstantFill', {'value': 0.0}), True)
        return self.net.add('Conv', [blob_in, w, b], [blob_out], dict(kwargs, kernel=kernel))

    def FC(self, blob_in, blob_out, dim_in, dim_out, weight_init=None, bias_init=None, **kwargs):
        w = self._create_param(blob_out + '_w', (dim_out, dim_in),
                               weight_init or ('XavierFill', {}), False)
        b = self._create_param(blob_out + '_b', (dim_out,),
                               bias_init or ('ConstantFill', {'value': 0.0}), True)
        return self.net.add('FC', [blob_in, w, b], [blob_out], kwargs)

    def FCShared(self, blob_in, blob_out, dim_in, dim_out, weight=None, bias=None, **kwargs):
        return self.net.add('FC', [blob_in, weight, bias], [blob_out], kwargs)

    def Relu(self, blob_in, blob_out):
        return self.net.add('Relu', [blob_in], [blob_out], {})

    def MaxPool(self, blob_in, blob_out, **kwargs):
        return self.net.add('MaxPool', [blob_in], [blob_out], kwargs)

    def Dropout(self, blob_in, blob_out, **kwargs):
        return self.net.add('Dropout', [blob_in], [blob_out, '_' + str(blob_out) + '_mask'],
                            kwargs)[0]

    def Softmax(self, blob_in, blob_out, **kwargs):
        return self.net.add('Softmax', [blob_in], [blob_out], kwargs)

    def Transpose(self, blob_in, blob_out, **kwargs):
        return self.net.add('Transpose', [blob_in], [blob_out], kwargs)

    def StopGradient(self, blob_in, blob_out):
        return self.net.add('StopGradient', [blob_in], [blob_out], {})

    def Accuracy(self, blobs_in, blob_out, **kwargs):
        return self.net.add('Accuracy', blobs_in, [blob_out], kwargs)

    def DropoutIfTraining(self, blob_in, dropout_rate):
        if self.train and dropout_rate > 0:
            return self.Dropout(blob_in, blob_in, ratio=dropout_rate, is_test=False)
        return blob_in

    def RoIFeatureTransform(self, blobs_in, blob_out, blob_rois='rois', method='RoIPoolF',
                            resolution=7, spatial_scale=1. / 16., sampling_ratio=0):
        """Single feature level only (FPN is outside the hot path).  ref: detector.py:268-331."""
        assert method in {'RoIPoolF'}, 'Unknown pooling method: {}'.format(method)
        assert not isinstance(blobs_in, list), 'FPN RoI transforms are not on the hot path'
        out = self.net.add(method, [blobs_in, blob_rois], [blob_out, '_argmax_' + blob_out],
                           dict(pooled_w=resolution, pooled_h=resolution,
                                spatial_scale=spatial_scale, sampling_ratio=sampling_ratio))
        return out[0]

    # ---------------------------------------------------------- losses / metrics
    def AddLosses(self, losses):
        for l in _as_list(losses):
            if l not in self.losses:
                self.losses.append(l)

    def AddMetrics(self, metrics):
        for m in _as_list(metrics):
            if m not in self.metrics:
                self.metrics.append(m)

    def GetLossScale(self):
        return 1.0 / cfg.NUM_GPUS

    # ------------------------------------------------------------------ backward
    def AddGradientOperators(self, loss_gradients):
        """Reverse-mode op generation over the recorded forward ops.  `loss_gradients` maps a
        loss blob to the blob holding its gradient seed (blob.py:167-173).  Blobs consumed by
        several ops accumulate (`<name>_grad` summed); StopGradient and the ops in _NO_GRAD cut
        the flow, so nothing upstream of `roi_feat` / `conv5_3` or inside the entropy gate gets
        a gradient op."""
        grad_of = {str(k): str(v) for k, v in loss_gradients.items()}
        ops = []
        params = set(self.params)
        for op in reversed(self.net.ops):
            if op.type in _NO_GRAD:
                for o in op.outputs:
                    grad_of.pop(o, None)
                continue
            gouts = [grad_of.get(o) for o in op.outputs]
            if not any(gouts):
                continue
            if op.type not in _GRAD_INPUTS:
                raise NotImplementedError('no gradient for op {} (blob {} needs one)'.format(
                    op.type, op.outputs[0]))
            gin = []
            for i, name in enumerate(op.inputs):
                gin.append(name + '_grad' if i in _GRAD_INPUTS[op.type] else None)
            # in-place ops (Relu fc6->fc6): the output's gradient is consumed here
            for o in op.outputs:
                if o in op.inputs:
                    grad_of.pop(o, None)
            accumulate = [g is not None and op.inputs[i] in grad_of and op.inputs[i] not in params
                          for i, g in enumerate(gin)]
            ops.append(Op(op.type + 'Gradient', list(op.inputs) + list(op.outputs),
                          [g for g in gin if g], dict(op.args, _gout=gouts, _gin=gin,
                                                      _accumulate=accumulate)))
            for i, g in enumerate(gin):
                if g:
                    grad_of[op.inputs[i]] = g
        for p in self.params:
            if p in grad_of:
                self.param_to_grad[p] = grad_of[p]
        self.grad_ops = ops
        return ops

    # ------------------------------------------------------------------------ lr
    def UpdateWorkspaceLr(self, cur_iter, new_lr):
        """ref: detector.py:509-559 (lr feed + momentum correction); the executor owns the
        device-side lr scalar and the momentum buffers."""
        return self.executor.update_lr(cur_iter, new_lr)
