"""Noise-aware (webly) heads.  Mirrors, from detectron/modeling/webly_heads.py:
add_webly_outputs (:32-74), add_webly_losses (:123-216), add_spatial_entropy_weight
(:265-440), add_VGG16_roi_2fc_noise_head (:463-502)."""
from detectron.core.config import cfg
from detectron.utils.c2 import const_fill
import detectron.utils.blob as blob_utils
from detectron.modeling.wsl_heads import (_dual_softmax, _two_fc, add_cls_pred,
                                          add_cross_entropy_loss, add_VGG16_roi_2fc_head,
                                          add_wsl_outputs)


def add_webly_outputs(model, blob_in, dim, prefix=''):
    """Clean branch outputs, then the residual noise branch: noisy_fc8{c,d} on the noisy
    features are ADDED to the clean logits before the same dual softmax."""
    add_wsl_outputs(model, blob_in[0], dim[0], prefix=prefix)
    n_fg = model.num_classes - 1
    for name in ('noisy_fc8c', 'noisy_fc8d'):
        model.FC(blob_in[1], prefix + name, dim[1], n_fg, weight_init=('XavierFill', {}),
                 bias_init=const_fill(0.0))
    model.net.Add([prefix + 'fc8c', prefix + 'noisy_fc8c'], [prefix + 'fc8c_noise'])
    model.net.Add([prefix + 'fc8d', prefix + 'noisy_fc8d'], [prefix + 'fc8d_noise'])
    _dual_softmax(model, prefix + 'fc8c_noise', prefix + 'fc8d_noise', prefix, '_noise')


def add_webly_losses(model, prefix=''):
    add_cls_pred(prefix + 'rois_pred', prefix + 'cls_prob', model, prefix='')
    add_cls_pred(prefix + 'rois_pred_noise', prefix + 'cls_prob_noise', model, prefix='')
    weight = weight_noise = None
    if cfg.WEBLY.ENTROPY:
        add_spatial_entropy_weight(model, prefix + 'rois_pred', prefix + 'cls_prob',
                                   prefix + 'rois')
        weight = prefix + 'rois' + '_class_weight'
        weight_noise = prefix + 'rois' + '_class_weight_noise'
    loss_gradients = {}
    for suffix, w in (('', weight), ('_noise', weight_noise)):
        add_cross_entropy_loss(model, prefix + 'cls_prob' + suffix, 'labels_oh',
                               prefix + 'cross_entropy' + suffix, weight=w, cpg=None)
        loss = model.net.AveragedLoss([prefix + 'cross_entropy' + suffix],
                                      [prefix + 'loss_cls' + suffix])
        loss_gradients.update(blob_utils.get_loss_gradients(model, [loss]))
        model.Accuracy([prefix + 'cls_prob' + suffix, 'labels_int32'],
                       prefix + 'accuracy_cls' + suffix)
        model.AddLosses([prefix + 'loss_cls' + suffix])
        model.AddMetrics(prefix + 'accuracy_cls' + suffix)
    if cfg.WSL.MIN_ENTROPY_LOSS:       # webly_heads.py:208-214
        from detectron.modeling.wsl_heads import add_min_entropy_loss
        loss_gradients.update(add_min_entropy_loss(model, prefix + 'rois_pred', 'labels_oh',
                                                   prefix + 'loss_entropy', cpg=None))
    return loss_gradients


def add_entropy_weight(model, rois_pred_blob, rois_blob):
    """webly_heads.py:219-262 - the class weights from the entropy of the NMS-ed detections
    (BoxWithNMSLimit at NMS 0.9 -> RoIEntropy -> max with 1 - labels).  The reference defines it
    but never calls it (the call at :131 is commented out; add_spatial_entropy_weight replaced it);
    it is mirrored so that a user builder can."""
    model.net.Split(rois_pred_blob, [rois_pred_blob + '_bg', rois_pred_blob + '_useless'],
                    split=[1, model.num_classes - 2], axis=1)
    model.net.Concat([rois_pred_blob + '_bg', rois_pred_blob],
                     [rois_pred_blob + '_fgbg', rois_pred_blob + '_fgbg_concat_dims'], axis=1)
    model.net.Split(rois_blob, [rois_blob + '_useless', rois_blob + '_4'], split=[1, 4], axis=1)
    model.net.Tile(rois_blob + '_4', rois_blob + '_fgbg', axis=1, tiles=model.num_classes)
    model.net.BoxWithNMSLimit(
        [rois_pred_blob + '_fgbg', rois_blob + '_fgbg'],
        [rois_pred_blob + '_nms', rois_blob + '_nms', rois_blob + '_classes_nms'],
        score_thresh=0.00000000001, nms=0.9, detections_per_im=999999)
    model.net.RoIEntropy([rois_pred_blob + '_nms', rois_blob + '_classes_nms'],
                         [rois_blob + '_entropy'], display=int(1280 / cfg.NUM_GPUS),
                         num_classes=model.num_classes - 1)
    model.net.ConstantFill('labels_oh', 'labels_oh_one', value=1.0)
    model.net.Sub(['labels_oh_one', 'labels_oh'], 'labels_oh_inv')
    model.net.Max([rois_blob + '_entropy', 'labels_oh_inv'], rois_blob + '_class_weight')
    return rois_blob + '_class_weight'


def add_spatial_entropy_weight(model, rois_pred, cls_prob, rois):
    """Per-class loss weights from the IoU-graph-smoothed proposal entropy (no gradient)."""
    net = model.net
    net.RoIIoU([rois], [rois + '_J'])
    # E = ReplaceNaN(-(p log p))
    net.Log(rois_pred, rois_pred + '_log')
    net.Mul([rois_pred, rois_pred + '_log'], rois_pred + '__E')
    net.Scale(rois_pred + '__E', rois_pred + '_E', scale=-1.0)
    net.ReplaceNaN(rois_pred + '_E', rois_pred + '_E')
    # hatE = E * (E / LeakyRelu(J @ E)), summed over proposals
    net.MatMul([rois + '_J', rois_pred + '_E'], rois_pred + '_D')
    net.LeakyRelu(rois_pred + '_D', rois_pred + '_D')
    net.Div([rois_pred + '_E', rois_pred + '_D'], rois_pred + '_G')
    net.Mul([rois_pred + '_E', rois_pred + '_G'], rois_pred + '_hatE')
    net.ReduceSum(rois_pred + '_hatE', rois_pred + '_hatE_sum', axes=[0], keepdims=True)
    # normaliser (log N - log y) * y
    net.Shape(rois_pred, rois_pred + '_N', axes=[0])
    net.Cast(rois_pred + '_N', rois_pred + '_N_float', to=1)
    net.Log(cls_prob, cls_prob + '_logy')
    net.Log(rois_pred + '_N_float', rois_pred + '_logN')
    net.Sub([rois_pred + '_logN', cls_prob + '_logy'], rois_pred + '_logN__logy')
    net.Mul([rois_pred + '_logN__logy', cls_prob], rois_pred + '_y_logN__logy')
    net.Div([rois_pred + '_hatE_sum', rois_pred + '_y_logN__logy'],
            [rois_pred + '_hatE_sum_norm'], broadcast=True)
    net.Clip(rois_pred + '_hatE_sum_norm', rois_pred + '_hatE_sum_norm', max=1.0, min=0.0)
    # gate only the classes the image is NOT labelled with
    net.ConstantFill(['labels_oh'], 'labels_oh_one', value=1.0)
    net.Sub(['labels_oh_one', 'labels_oh'], ['labels_oh_bg'])
    net.Mul([rois_pred + '_hatE_sum_norm', 'labels_oh_bg'], [rois + '_class_weight_noise'])
    net.Sub(['labels_oh_one', rois + '_class_weight_noise'], [rois + '_class_weight'])
    model.StopGradient(rois + '_class_weight', rois + '_class_weight')
    model.StopGradient(rois + '_class_weight_noise', rois + '_class_weight_noise')
    display = int(1280 / cfg.NUM_GPUS)
    stats = (
        (rois + '_class_weight', 'labels_oh_bg', rois + '_class_weight_stat', 'labels_oh_stat0',
         'class_weight      '),
        (rois + '_class_weight_noise', 'labels_oh_bg', rois + '_class_weight_noise_stat',
         'labels_oh_stat1', 'class_weight_noise'),
        (rois_pred + '_hatE_sum', 'labels_oh_bg', rois_pred + '_hatE_sum_bg_stat',
         'labels_oh_one_stat2', 'hatE_sum bg       '),
        (rois_pred + '_hatE_sum', 'labels_oh', rois_pred + '_hatE_sum_fg_stat',
         'labels_oh_one_stat3', 'hatE_sum fg       '),
        (rois_pred + '_hatE_sum_norm', 'labels_oh_bg', rois_pred + '_hatE_sum_norm_bg_stat',
         'labels_oh_one_stat4', 'hatE_sum_norm bg  '),
        (rois_pred + '_hatE_sum_norm', 'labels_oh', rois_pred + '_hatE_sum_norm_fg_stat',
         'labels_oh_one_stat5', 'hatE_sum_norm fg  '),
    )
    for i_blob, l_blob, ai, al, label in stats:
        net.Stat([i_blob, l_blob], [ai, al], display=display, prefix=label)


def add_VGG16_roi_2fc_noise_head(model, blob_in, dim_in, spatial_scale, prefix=''):
    """Clean 2-fc head plus a second 2-fc stack ('_[noisy]_*' params) on the SAME roi_feat."""
    clean, dim_out = add_VGG16_roi_2fc_head(model, blob_in, dim_in, spatial_scale, prefix=prefix)
    roi_size = cfg.FAST_RCNN.ROI_XFORM_RESOLUTION
    tag = '_[' + prefix + 'noisy]_'
    noisy = _two_fc(model, prefix + 'roi_feat', tag + 'fc6', tag + 'drop6', tag + 'fc7',
                    tag + 'drop7', dim_in * roi_size * roi_size)
    return [clean, noisy], [dim_out, 4096]
