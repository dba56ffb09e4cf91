"""WSDDN heads on the hot path.  Mirrors, from detectron/modeling/wsl_heads.py:
add_wsl_outputs (:23-78), add_cls_pred (:213-227), add_cross_entropy_loss (:292-302),
add_VGG16_roi_2fc_head (:654-681), DropoutIfTraining (:1259-1267).  The OICR / PCL / CMIL /
CSC / context / center-loss heads of that file are other WSOD methods (cfg switches that
core/config.py rejects)."""
from detectron.core.config import cfg
from detectron.utils.c2 import const_fill


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
