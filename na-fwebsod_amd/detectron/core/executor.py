"""Net executor: runs the graph a DetectionModelHelper recorded on one MI355X.

Replaces `workspace.RunNet(model.net)` + the Caffe2 dag executor of the reference
(detectron/utils/train_wsl.py:59, detectron/modeling/detector.py:63-64).  Two plans:

  * FUSED — chosen when the recorded op list is exactly the na_wsddn graph (the one
    `VGG16.add_VGG16_conv5_body_origin` + `webly_heads.add_VGG16_roi_2fc_noise_head` +
    `add_webly_outputs` + `add_webly_losses` emit): the blobs run through
    naws_hip.engine.WsddnEngine (fused kernels, any number of images per process).
  * INTERPRETED — any other graph made of the supported operators runs op by op through
    detectron.ops (same HIP kernels, one launch group per op, reference semantics of one
    image per process).  It is also the cross-check of the fused plan in the tests.

Blob names are the reference's unscoped names (one process per GPU).
"""
import numpy as np
import torch

from detectron.core.config import cfg
import detectron.ops as O
from naws_hip import lib as L
from naws_hip import ops as K


def _signature(ops_list):
    return [(o.type, tuple(o.inputs), o.outputs[0]) for o in ops_list]


def canonical_na_wsddn_signature(train, num_classes):
    """The op list the reference builders emit for the hot-path config under the current cfg."""
    from detectron.modeling.detector import DetectionModelHelper
    from detectron.modeling import VGG16, webly_heads
    m = DetectionModelHelper(name='canon', train=train, num_classes=num_classes, init_params=train)
    blob, dim, scale = VGG16.add_VGG16_conv5_body_origin(m)
    m.StopGradient(blob, blob)
    ls, dims = webly_heads.add_VGG16_roi_2fc_noise_head(m, blob, dim, scale)
    webly_heads.add_webly_outputs(m, ls, dims)
    if train:
        webly_heads.add_webly_losses(m)
    return _signature(m.net.ops)


def _fill(init, shape, gen, device):
    kind, args = init
    if kind == 'ConstantFill':
        return torch.full(shape, float(args.get('value', 0.0)), device=device)
    if kind == 'GaussianFill':
        return (torch.randn(shape, generator=gen) * args.get('std', 1.0) +
                args.get('mean', 0.0)).to(device)
    if kind == 'XavierFill':       # U(+-sqrt(3/fan_in)), fan_in = prod(shape[1:])
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        lim = (3.0 / fan_in) ** 0.5
        return ((torch.rand(shape, generator=gen) * 2 - 1) * lim).to(device)
    if kind == 'MSRAFill':
        fan_out = int(np.prod(shape)) // shape[1] if len(shape) > 1 else shape[0]
        return (torch.randn(shape, generator=gen) * (2.0 / fan_out) ** 0.5).to(device)
    raise NotImplementedError('initializer ' + kind)


class NetExecutor(object):
    def __init__(self, model, device, process_group=None, world_size=1, rank=0,
                 force_interpreted=False, images_per_process=1, disable_dropout=False):
        self.model, self.device = model, device
        self.pg, self.world, self.rank = process_group, int(world_size), int(rank)
        self.ws = {}
        self.step = 0
        self.lr = torch.zeros((1,), device=device)
        self.ims = int(images_per_process)
        self.disable_dropout = bool(disable_dropout)   # parity tests only
        model.executor = self
        sig = _signature(model.net.ops)
        fused_ok = (not force_interpreted and cfg.WEBLY.WEBLY_ON and cfg.WEBLY.ENTROPY and
                    cfg.TRAIN.FREEZE_CONV_BODY and
                    sig == canonical_na_wsddn_signature(model.train, model.num_classes))
        self.plan = 'fused' if fused_ok else 'interpreted'
        self.engine = None
        if fused_ok:
            from naws_hip.engine import WsddnEngine
            self.engine = WsddnEngine(
                model.num_classes, device, dilation=cfg.WSL.DILATION,
                roi_size=cfg.FAST_RCNN.ROI_XFORM_RESOLUTION,
                dropout=0.0 if disable_dropout else 0.5,
                is_mean=cfg.WSL.MEAN_LOSS, momentum=cfg.SOLVER.MOMENTUM,
                weight_decay=cfg.SOLVER.WEIGHT_DECAY, iter_size=cfg.WSL.ITER_SIZE,
                gpu_num=self.world * self.ims, seed=cfg.RNG_SEED, process_group=process_group,
                world_size=world_size, allreduce_chunks=cfg.NAWS.ALLREDUCE_CHUNKS,
                mfma_dtype=cfg.NAWS.MFMA_DTYPE, scale_momentum=cfg.SOLVER.SCALE_MOMENTUM,
                scale_momentum_threshold=cfg.SOLVER.SCALE_MOMENTUM_THRESHOLD,
                sharded_update=cfg.NAWS.SHARDED_UPDATE, rank=self.rank,
                pipeline_update=cfg.NAWS.PIPELINE_UPDATE)
        else:
            if self.ims != 1:
                raise NotImplementedError('the op-by-op plan follows the reference: one image '
                                          'per process (wsl_heads.py:214)')
            self._stats, self._sgd = {}, {}

    # -------------------------------------------------------------- parameters
    def init_params(self, seed=None):
        gen = torch.Generator().manual_seed(cfg.RNG_SEED if seed is None else seed)
        blobs = {n: _fill(self.model.param_inits[n], self.model.param_shapes[n], gen, self.device)
                 for n in self.model.params}
        self.load_blobs(blobs)

    def load_blobs(self, blobs):
        """blobs: {unscoped name: tensor} in the reference layouts (FC [out,in], conv OIHW)."""
        if self.engine is not None:
            self.engine.set_conv_blobs(blobs)
            self.engine.set_head_blobs(blobs)
            for n in self.model.params:
                if n + '_momentum' in blobs and n in self.engine.arena.offsets:
                    self.engine.momentum_blob(n).copy_(blobs[n + '_momentum'].to(self.device))
        else:
            for n in self.model.params:
                self.ws[n] = blobs[n].to(self.device, torch.float32).contiguous()
                self.ws[n + '_momentum'] = torch.zeros_like(self.ws[n])
                self.ws[n + '_acmgrad'] = torch.zeros_like(self.ws[n])

    def blobs(self, with_momentum=True):
        if self.engine is not None:
            return self.engine.export_blobs(with_momentum)
        out = {}
        for n in self.model.params:
            out[n] = self.ws[n]
            if with_momentum:
                out[n + '_momentum'] = self.ws[n + '_momentum']
        return out

    def broadcast_parameters(self):
        """Rank 0's parameters to every rank (net_wsl.py:183-207 does host copies)."""
        if self.pg is None or self.world <= 1:
            return
        import torch.distributed as dist
        if self.engine is not None:
            self.engine.broadcast_parameters(0)
        else:
            for n in self.model.params:
                dist.broadcast(self.ws[n], 0, group=self.pg)

    # ---------------------------------------------------------------------- lr
    def update_lr(self, cur_iter, new_lr):
        new_lr = float(np.float32(new_lr))
        if self.engine is not None:
            self.lr.fill_(new_lr)
            return self.engine.set_lr(new_lr)
        cur = float(self.lr.item())
        if cur != new_lr:
            ratio = max(new_lr / max(cur, 1e-10), cur / max(new_lr, 1e-10))
            self.lr.fill_(new_lr)
            if cfg.SOLVER.SCALE_MOMENTUM and cur > 1e-7 and \
                    ratio > cfg.SOLVER.SCALE_MOMENTUM_THRESHOLD:
                for n in self.model.TrainableParams():
                    m = self.ws[n + '_momentum']
                    # a float32 quotient, as in the reference (detector.py:536-537) and the engine
                    K.unary(L.UN_SCALE, m, float(np.float32(new_lr) / np.float32(cur)), out=m)
        return new_lr

    # -------------------------------------------------------------------- run
    def feed(self, blobs):
        self.ws.pop('_seg', None)      # host-side row offsets only ever describe the current rois
        for k, v in blobs.items():
            self.ws[k] = v

    def fetch(self, name):
        return self.ws[name]

    def run(self):
        """One iteration: forward (+ backward, all-reduce, SGD when training)."""
        if self.engine is not None:
            self._run_fused()
        else:
            self._run_interpreted()
        self.step += 1

    def _run_fused(self):
        ws, eng = self.ws, self.engine
        if self.model.train:
            out = eng.train_step(ws['data'], ws['rois'], ws['obn_scores'], ws['labels_oh'],
                                 seg=ws.get('_seg'))
            ws['loss_cls'], ws['loss_cls_noise'] = out['loss_cls'], out['loss_cls_noise']
            ws['cls_prob'], ws['cls_prob_noise'] = out['cls_prob'], out['cls_prob_noise']
            ws['rois_class_weight'] = out['class_weight']
            ws['rois_class_weight_noise'] = out['class_weight_noise']
            ws['rois_pred'] = out['rois_pred']
            if self.rank == 0:
                eng.stat_update(out, ws['labels_oh'], max(1, int(1280 / cfg.NUM_GPUS)))
        else:
            ws['cls_prob'] = eng.infer(ws['data'], ws['rois'], ws['obn_scores'], seg=ws.get('_seg'))

    # ------------------------------------------------------ op-by-op plan
    def _run_interpreted(self):
        ws = self.ws
        for idx, op in enumerate(self.model.net.ops):
            self._forward(idx, op, ws)
        if not self.model.train:
            return
        for op in self.model.grad_ops:
            self._backward(op, ws)
        params = self.model.TrainableParams()
        if self.pg is not None and self.world > 1:
            import torch.distributed as dist
            for p in params:
                dist.all_reduce(ws[self.model.param_to_grad[p]], group=self.pg)
        for p in params:
            if p not in self._sgd:
                bias = p in self.model.biases
                lm = (2.0 if bias else 1.0) * (10.0 if '_lrm10_' in p else 1.0)
                self._sgd[p] = O.ACMWeightDecayMomentumSGDUpdate(
                    momentum=cfg.SOLVER.MOMENTUM, weight_decay=0.0 if bias else
                    cfg.SOLVER.WEIGHT_DECAY, iter_size=cfg.WSL.ITER_SIZE, gpu_num=self.world,
                    lr_mult=lm)
            self._sgd[p](ws[self.model.param_to_grad[p]].contiguous(), ws[p + '_momentum'],
                         self.lr, ws[p], ws[p + '_acmgrad'])

    def _forward(self, idx, op, ws):
        t, a = op.type, op.args
        x = [ws[n] for n in op.inputs if n in ws]
        out = op.outputs
        if t == 'Conv':
            ws[out[0]] = O.Conv(x[0], x[1], x[2], kernel=a.get('kernel', 3), pad=a.get('pad', 1),
                                stride=a.get('stride', 1), dilation=a.get('dilation', 1))
        elif t == 'Relu':
            ws[out[0]] = O.Relu(x[0])
        elif t == 'MaxPool':
            ws[out[0]] = O.MaxPool(x[0], kernel=a['kernel'], pad=a['pad'], stride=a['stride'])
        elif t == 'StopGradient':
            ws[out[0]] = x[0]
        elif t == 'RoIPoolF':
            ws[out[0]], ws[out[1]] = O.RoIPoolF(x[0], x[1], a['pooled_h'], a['pooled_w'],
                                                a['spatial_scale'])
        elif t == 'RoIFeatureBoost':
            ws[out[0]] = O.RoIFeatureBoost(x[0], x[1])
        elif t == 'FC':
            ws[out[0]] = O.FC(x[0], x[1], x[2])
        elif t == 'Dropout':
            y, mask = O.Dropout(x[0], ratio=a.get('ratio', 0.5),
                                is_test=a.get('is_test', False) or self.disable_dropout,
                                seed=(cfg.RNG_SEED * 1000003 + self.step * 131 + idx))
            ws[out[0]] = y
            ws[out[1]] = mask
        elif t == 'Softmax':
            ws[out[0]] = O.Softmax(x[0], axis=a.get('axis', 1))
        elif t == 'Transpose':
            ws[out[0]] = O.Transpose(x[0], axes=a.get('axes', (1, 0)))
        elif t in ('Add', 'Sub', 'Mul', 'Div'):
            ws[out[0]] = getattr(O, t)(x[0], x[1])
        elif t == 'ReduceSum':
            ws[out[0]] = O.ReduceSum(x[0], axes=a.get('axes', [0]), keepdims=a.get('keepdims', True))
        elif t == 'RoIIoU':
            ws[out[0]] = O.RoIIoU(x[0])
        elif t == 'Log':
            ws[out[0]] = O.Log(x[0])
        elif t == 'Scale':
            ws[out[0]] = O.Scale(x[0], scale=a.get('scale', 1.0))
        elif t == 'ReplaceNaN':
            ws[out[0]] = O.ReplaceNaN(x[0], value=a.get('value', 0.0))
        elif t == 'MatMul':
            ws[out[0]] = O.MatMul(x[0], x[1])
        elif t == 'LeakyRelu':
            ws[out[0]] = O.LeakyRelu(x[0], alpha=a.get('alpha', 0.01))
        elif t == 'Shape':
            ws[out[0]] = [x[0].shape[i] for i in a.get('axes', range(x[0].dim()))]
        elif t == 'Cast':
            ws[out[0]] = torch.tensor([float(v) for v in x[0]], device=self.device)
        elif t == 'Clip':
            ws[out[0]] = O.Clip(x[0], min=a.get('min', -3.4e38), max=a.get('max', 3.4e38))
        elif t == 'ConstantFill':
            ws[out[0]] = O.ConstantFill(like=x[0] if x else None, shape=a.get('shape'),
                                        value=a.get('value', 0.0), device=self.device)
        elif t == 'Stat':
            if idx not in self._stats:
                self._stats[idx] = O.Stat(display=a.get('display', 1280), prefix=a.get('prefix', ''))
            ws[out[0]], ws[out[1]] = self._stats[idx](x[0].reshape(-1), x[1].reshape(-1),
                                                      gpu_id=self.rank)
        elif t == 'WeightedCrossEntropyWithLogits':
            ws[out[0]] = O.WeightedCrossEntropyWithLogits(x[0], x[1], x[2],
                                                          is_mean=a.get('is_mean', False))
        elif t == 'CrossEntropyWithLogits':
            ws[out[0]] = O.CrossEntropyWithLogits(x[0], x[1], is_mean=a.get('is_mean', False))
        elif t == 'AveragedLoss':
            ws[out[0]] = O.AveragedLoss(x[0])
        elif t == 'MinEntropyLoss':
            ws[out[0]] = O.MinEntropyLoss(x[0], x[1])
        elif t == 'Accuracy':
            ws[out[0]] = O.Accuracy(x[0], x[1])
        elif t == 'RoILabel':
            if idx not in self._stats:
                keys = ('display', 'uuid', 'fg_thresh', 'bg_thresh_hi', 'bg_thresh_lo', 'num_pos',
                        'num_neg', 'top_k')
                self._stats[idx] = O.RoILabel(**{k: a[k] for k in keys if k in a})
            ws[out[0]], ws[out[1]] = self._stats[idx](*x)
        elif t == 'SoftmaxWithLossN':
            ws[out[0]], ws[out[1]] = O.SoftmaxWithLossN(x[0], x[1], x[2] if len(x) > 2 else None,
                                                        scale=a.get('scale', 1.0))
        elif t == 'RoIEntropy':
            if idx not in self._stats:
                self._stats[idx] = O.RoIEntropy(display=a.get('display', 1280),
                                                num_classes=a.get('num_classes', 20),
                                                rm_bg=a.get('rm_bg', True))
            ws[out[0]] = self._stats[idx](x[0], x[1])
        elif t == 'BoxWithNMSLimit':
            ws[out[0]], ws[out[1]], ws[out[2]] = O.BoxWithNMSLimit(
                x[0], x[1], score_thresh=a.get('score_thresh', 0.05), nms=a.get('nms', 0.3),
                detections_per_im=a.get('detections_per_im', 100))
        elif t == 'Mean':
            ws[out[0]] = O.Mean(*x)
        elif t == 'Max':
            ws[out[0]] = O.Max(x[0], x[1])
        elif t == 'Tile':
            ws[out[0]] = O.Tile(x[0], tiles=a.get('tiles', 1), axis=a.get('axis', 1))
        elif t == 'Split':
            parts = torch.split(x[0], a['split'], dim=a.get('axis', 1))
            for n, p in zip(out, parts):
                ws[n] = p
        elif t == 'Concat':
            ws[out[0]] = torch.cat(x, dim=a.get('axis', 1)).contiguous()
        else:
            raise NotImplementedError('operator {} is not on the MI355X hot path'.format(t))

    def _backward(self, op, ws):
        t, a = op.type[:-len('Gradient')], op.args
        n_in = len(a['_gin'])
        ins, outs = op.inputs[:n_in], op.inputs[n_in:]
        gout = [ws[g] if g else None for g in a['_gout']]
        res = [None] * n_in
        if t == 'FC':
            dW, db, dX = O.FCGradient(ws[ins[0]], ws[ins[1]], gout[0].contiguous())
            res = [dX, dW, db]
        elif t == 'Relu':
            res = [O.ReluGradient(ws[outs[0]], gout[0])]
        elif t == 'Dropout':
            res = [gout[0] if ws[outs[1]] is None else
                   O.DropoutGradient(gout[0], ws[outs[1]], a.get('ratio', 0.5))]
        elif t == 'Softmax':
            res = [O.SoftmaxGradient(ws[outs[0]], gout[0])]
        elif t == 'Transpose':
            res = [O.Transpose(gout[0])]
        elif t == 'Mul':
            res = [O.Mul(gout[0], ws[ins[1]]), O.Mul(gout[0], ws[ins[0]])]
        elif t == 'Add':
            res = [gout[0], gout[0]]
        elif t == 'ReduceSum':      # broadcast the [1,C] gradient down the rows
            res = [O.Mul(O.ConstantFill(like=ws[ins[0]], value=1.0), gout[0])]
        elif t == 'AveragedLoss':
            res = [O.Scale(gout[0].reshape(1), 1.0 / max(ws[ins[0]].numel(), 1))]
        elif t == 'WeightedCrossEntropyWithLogits':
            res = [O.WeightedCrossEntropyWithLogitsGradient(ws[ins[0]], ws[ins[1]], ws[ins[2]],
                                                            gout[0], is_mean=a.get('is_mean', False))]
        elif t == 'CrossEntropyWithLogits':
            res = [O.CrossEntropyWithLogitsGradient(ws[ins[0]], ws[ins[1]], gout[0],
                                                    is_mean=a.get('is_mean', False))]
        elif t == 'RoIFeatureBoost':
            res = [O.RoIFeatureBoostGradient(gout[0], ws[ins[1]])]
        elif t == 'MinEntropyLoss':
            res = [O.MinEntropyLossGradient(ws[ins[0]], ws[ins[1]], gout[0])]
        elif t == 'SoftmaxWithLossN':      # inputs X, T[, W]; outputs P, loss; seed = d(loss)
            res = [O.SoftmaxWithLossNGradient(ws[ins[0]], ws[ins[1]],
                                              ws[ins[2]] if n_in > 2 else None, ws[outs[0]],
                                              gout[1], scale=a.get('scale', 1.0))] + [None] * (n_in - 1)
        else:
            raise NotImplementedError('gradient of ' + t)
        for i, g in enumerate(a['_gin']):
            if g is None or res[i] is None:
                continue
            if a['_accumulate'][i] and g in ws:
                ws[g] = O.Add(ws[g], res[i]).view(res[i].shape)
            else:
                ws[g] = res[i]
