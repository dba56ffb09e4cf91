"""WSDDN heads on the hot path.  Mirrors, from detectron/modeling/wsl_heads.py:
add_wsl_outputs (:23-78), add_cls_pred (:213-227), add_cross_entropy_loss (:292-302),
add_VGG16_roi_2fc_head (:654-681), DropoutIfTraining (:1259-1267); and, one cfg flag away
(WSL.OICR): add_wsl_oicr_outputs (:134-156), add_wsl_losses (:375-458),
add_oicr_losses (:512-560).  The PCL / CMIL / CSC / context / center-loss heads of that file are
other WSOD methods (cfg switches that core/config.py rejects)."""
from detectron.core.config import cfg
from detectron.utils.c2 import const_fill, gauss_fill
import detectron.utils.blob as blob_utils


def add_wsl_outputs(model, blob_in, dim, prefix=''):
    """fc8c / fc8d -> softmax over classes x softmax over proposals -> rois_pred."""
    n_fg = model.num_classes - 1
    fc8c = model.FC(blob_in, prefix + 'fc8c', dim, n_fg, weight_init=('XavierFill', {}),
                    bias_init=const_fill(0.0))
    fc8d = model.FC(blob_in, prefix + 'fc8d', dim, n_fg, weight_init=('XavierFill', {}),
                    bias_init=const_fill(0.0))
    _dual_softmax(model, fc8c, fc8d, prefix, '')
    if not model.train:
        # background column = copy of the first foreground score (:58-67)
        model.net.Split(prefix + 'rois_pred', [prefix + 'rois_bg_pred', prefix + 'notuse'],
                        split=[1, model.num_classes - 2], axis=1)
        model.net.Concat([prefix + 'rois_bg_pred', prefix + 'rois_pred'],
                         [prefix + 'cls_prob', prefix + 'cls_prob_concat_dims'], axis=1)
    if cfg.WSL.OICR:
        add_wsl_oicr_outputs(model, blob_in, dim, prefix=prefix)


def add_wsl_oicr_outputs(model, blob_in, dim, prefix=''):
    """The K = 3 OICR refinement classifiers (num_classes outputs incl. background); at test time
    their softmaxes are averaged into cls_prob (wsl_heads.py:134-156)."""
    K = 3
    for k in range(1, K + 1):
        model.FC(blob_in, prefix + 'cls_score' + str(k), dim, model.num_classes,
                 weight_init=gauss_fill(0.01), bias_init=const_fill(0.0))
    if not model.train:
        all_cls_prob = []
        for k in range(1, K + 1):
            all_cls_prob.append(model.Softmax(prefix + 'cls_score' + str(k),
                                              prefix + 'cls_prob' + str(k), axis=1))
        model.net.Mean(all_cls_prob, prefix + 'cls_prob')


def add_wsl_losses(model, prefix=''):
    """The plain WSDDN image-level loss (+ OICR refinement losses): wsl_heads.py:375-458 for the
    switches the MI355X path accepts (no CPG / CSC / center loss / CMIL / PCL)."""
    add_cls_pred(prefix + 'rois_pred', prefix + 'cls_prob', model, prefix='')
    add_cross_entropy_loss(model, prefix + 'cls_prob', 'labels_oh', prefix + 'cross_entropy',
                           weight=None, cpg=None)
    loss_cls = model.net.AveragedLoss([prefix + 'cross_entropy'], [prefix + 'loss_cls'])
    loss_gradients = blob_utils.get_loss_gradients(model, [loss_cls])
    model.Accuracy([prefix + 'cls_prob', 'labels_int32'], prefix + 'accuracy_cls')
    model.AddLosses([prefix + 'loss_cls'])
    model.AddMetrics(prefix + 'accuracy_cls')
    if cfg.WSL.MIN_ENTROPY_LOSS:
        loss_gradients.update(add_min_entropy_loss(model, prefix + 'rois_pred', 'labels_oh',
                                                   prefix + 'loss_entropy', cpg=None))
    if cfg.WSL.OICR:
        loss_gradients.update(add_oicr_losses(model, prefix))
    return loss_gradients


def add_oicr_losses(model, prefix=''):
    """wsl_heads.py:512-560: RoIIoU once; per branch k RoILabel (pseudo labels from the previous
    branch's scores - rois_pred for k = 1) -> SoftmaxWithLossN on cls_score_k."""
    loss_gradients = {}
    model.net.RoIIoU([prefix + 'rois'], [prefix + 'rois_iou'])
    import uuid
    uu = uuid.uuid4().int % 10000
    K = 3
    for k in range(1, K + 1):
        first = prefix + 'rois_pred' if k == 1 else prefix + 'cls_prob' + str(k - 1)
        model.net.RoILabel([first, prefix + 'rois_iou', 'labels_oh', prefix + 'cls_prob'],
                           [prefix + 'rois_labels_int32' + str(k), prefix + 'rois_weight' + str(k)],
                           display=int(1280 / cfg.NUM_GPUS), uuid=uu)
        cls_prob, loss_cls = model.net.SoftmaxWithLossN(
            [prefix + 'cls_score' + str(k), prefix + 'rois_labels_int32' + str(k),
             prefix + 'rois_weight' + str(k)],
            [prefix + 'cls_prob' + str(k), prefix + 'loss_cls' + str(k)])
        if cfg.WSL.MEAN_LOSS:
            lg = blob_utils.get_loss_gradients(model, [loss_cls])
        else:
            lg = get_loss_gradients_weighted(model, [loss_cls], 1. * (cfg.MODEL.NUM_CLASSES - 1))
        loss_gradients.update(lg)
        model.Accuracy([prefix + 'cls_prob' + str(k), prefix + 'rois_labels_int32' + str(k)],
                       prefix + 'accuracy_cls' + str(k))
        model.AddLosses([prefix + 'loss_cls' + str(k)])
        model.AddMetrics(prefix + 'accuracy_cls' + str(k))
    return loss_gradients


def _dual_softmax(model, fc8c, fc8d, prefix, suffix):
    model.Softmax(fc8c, prefix + 'alpha_cls' + suffix, axis=1)
    model.Transpose(fc8d, prefix + 'fc8d_t' + suffix, axes=(1, 0))
    model.Softmax(prefix + 'fc8d_t' + suffix, prefix + 'alpha_det_t' + suffix, axis=1)
    model.Transpose(prefix + 'alpha_det_t' + suffix, prefix + 'alpha_det' + suffix, axes=(1, 0))
    model.net.Mul([prefix + 'alpha_cls' + suffix, prefix + 'alpha_det' + suffix],
                  prefix + 'rois_pred' + suffix)


def add_cls_pred(in_blob, out_blob, model, prefix=''):
    """Image-level class scores = sum of the proposal scores.  The reference graph is one image
    per GPU (:214); the MI355X executor segments per image when a process holds several."""
    model.net.ReduceSum(in_blob, out_blob, axes=[0], keepdims=True)


def add_cross_entropy_loss(model, pred, label, loss, weight=None, cpg=None):
    ins = [pred, label]
    if cpg:
        ins.append(cpg)
    if weight:
        ins.insert(2, weight)
        model.net.WeightedCrossEntropyWithLogits(ins, [loss], is_mean=cfg.WSL.MEAN_LOSS)
    else:
        model.net.CrossEntropyWithLogits(ins, [loss], is_mean=cfg.WSL.MEAN_LOSS)


def DropoutIfTraining(model, blob_in, blob_out, dropout_rate):
    if model.train and dropout_rate > 0:
        return model.Dropout(blob_in, blob_out, ratio=dropout_rate, is_test=False)
    return blob_in


def _two_fc(model, blob, fc6, drop6, fc7, drop7, dim_in):
    blob = model.FC(blob, fc6, dim_in, 4096)
    blob = model.Relu(blob, fc6)
    blob = DropoutIfTraining(model, blob, drop6, 0.5)
    blob = model.FC(blob, fc7, 4096, 4096)
    blob = model.Relu(blob, fc7)
    return DropoutIfTraining(model, blob, drop7, 0.5)


def add_VGG16_roi_2fc_head(model, blob_in, dim_in, spatial_scale, prefix=''):
    roi_size = cfg.FAST_RCNN.ROI_XFORM_RESOLUTION
    feat = model.RoIFeatureTransform(
        blob_in, prefix + 'roi_feat', blob_rois=prefix + 'rois',
        method=cfg.FAST_RCNN.ROI_XFORM_METHOD, resolution=roi_size,
        sampling_ratio=cfg.FAST_RCNN.ROI_XFORM_SAMPLING_RATIO, spatial_scale=spatial_scale)
    feat = model.net.RoIFeatureBoost([feat, prefix + 'obn_scores'], feat)
    if cfg.TRAIN.FREEZE_CONV_BODY:
        feat = model.StopGradient(feat, feat)       # "save memory": no backward into RoIPool
    out = _two_fc(model, feat, prefix + 'fc6', prefix + 'drop6', prefix + 'fc7', prefix + 'drop7',
                  dim_in * roi_size * roi_size)
    return out, 4096


def add_min_entropy_loss(model, pred, label, loss, cpg=None):
    """wsl_heads.py:279-289: MinEntropyLoss on rois_pred with the loss gradient weighted 0.1."""
    in_blobs = [pred, label]
    if cpg:
        in_blobs.append(cpg)
    loss_entropy = model.net.MinEntropyLoss(in_blobs, [loss])
    loss_gradients = get_loss_gradients_weighted(model, [loss_entropy], 0.1)
    model.AddLosses([loss])
    return loss_gradients


def get_loss_gradients_weighted(model, loss_blobs, loss_weight):
    """wsl_heads.py:610-617: a gradient seed of loss_weight per loss blob."""
    loss_gradients = {}
    for b in loss_blobs:
        loss_grad = model.net.ConstantFill(b, [str(b) + '_grad'], value=1.0 * loss_weight)
        loss_gradients[str(b)] = str(loss_grad)
    return loss_gradients
