"""Model construction by name.  Mirrors detectron/modeling/model_builder_wsl.py: `create`
(:163-180), `get_func` (:183-208), `generalized_wsl` (:100-108),
`build_generic_wsl_detection_model` (:289-370), `_add_webly_head` (:434-456),
`add_training_inputs` (:604-639), `add_inference_inputs` (:642-655).  MODEL.TYPE,
MODEL.CONV_BODY and FAST_RCNN.ROI_BOX_HEAD are resolved relative to detectron.modeling, so
a user module dropped in there plugs in exactly as upstream."""
import importlib
import logging

from detectron.core.config import cfg
from detectron.modeling.detector import DetectionModelHelper
import detectron.modeling.optimizer_wsl as optim_wsl
import detectron.modeling.webly_heads as webly_heads
import detectron.roi_data.minibatch_wsl as roi_data_minibatch

logger = logging.getLogger(__name__)


def generalized_wsl(model):
    return build_generic_wsl_detection_model(
        model, get_func(cfg.MODEL.CONV_BODY),
        add_roi_box_head_func=get_func(cfg.FAST_RCNN.ROI_BOX_HEAD),
        add_roi_mask_head_func=get_func(cfg.MRCNN.ROI_MASK_HEAD),
        add_roi_keypoint_head_func=get_func(cfg.KRCNN.ROI_KEYPOINTS_HEAD),
        freeze_conv_body=cfg.TRAIN.FREEZE_CONV_BODY)


def create(model_type_func, train=False, gpu_id=0):
    model = DetectionModelHelper(name=model_type_func, train=train,
                                 num_classes=cfg.MODEL.NUM_CLASSES, init_params=train)
    model.only_build_forward_pass = False
    model.target_gpu_id = gpu_id
    return get_func(model_type_func)(model)


def get_func(func_name):
    """A function of this module, or `module.func` below detectron.modeling."""
    if func_name == '':
        return None
    try:
        parts = func_name.split('.')
        if len(parts) == 1:
            return globals()[parts[0]]
        module = importlib.import_module('detectron.modeling.' + '.'.join(parts[:-1]))
        return getattr(module, parts[-1])
    except Exception:
        logger.error('Failed to find function: {}'.format(func_name))
        raise


def build_generic_wsl_detection_model(model, add_conv_body_func, add_roi_box_head_func=None,
                                      add_roi_mask_head_func=None,
                                      add_roi_keypoint_head_func=None, freeze_conv_body=False):
    def _single_gpu_build_func(model):
        blob_conv, dim_conv, spatial_scale_conv = add_conv_body_func(model)
        if freeze_conv_body:
            model.StopGradient(blob_conv, blob_conv)
        if cfg.RPN.RPN_ON or cfg.FPN.FPN_ON or cfg.MODEL.MASK_ON or cfg.MODEL.KEYPOINTS_ON:
            raise NotImplementedError('RPN / FPN / mask / keypoint heads are outside the hot path')
        if cfg.WEBLY.WEBLY_ON:
            loss_gradients = _add_webly_head(model, add_roi_box_head_func, blob_conv, dim_conv,
                                             spatial_scale_conv)
        else:
            loss_gradients = _add_wsl_head(model, add_roi_box_head_func, blob_conv, dim_conv,
                                           spatial_scale_conv)
        return loss_gradients if model.train else None

    optim_wsl.build_data_parallel_model(model, _single_gpu_build_func)
    return model


def _add_webly_head(model, add_roi_box_head_func, blob_in, dim_in, spatial_scale_in):
    blob_frcn, dim_frcn = add_roi_box_head_func(model, blob_in, dim_in, spatial_scale_in)
    webly_heads.add_webly_outputs(model, blob_frcn, dim_frcn)
    if model.train and cfg.WEBLY.MINING:
        # The reference's branch (model_builder_wsl.py:443-452) calls webly_heads.add_webly_mining
        # and get_func(ROI_BOX_HEAD + '_shared') - neither function exists anywhere in the
        # reference tree, so WEBLY.MINING: True dies there with exactly this error.
        raise AttributeError("module 'detectron.modeling.webly_heads' has no attribute "
                             "'add_webly_mining' (nor does the reference: WEBLY.MINING is dead code "
                             "upstream, model_builder_wsl.py:443-452)")
    return webly_heads.add_webly_losses(model) if model.train else None


def _add_wsl_head(model, add_roi_box_head_func, blob_in, dim_in, spatial_scale_in):
    """model_builder_wsl.py:404-431 without the webly head: plain WSDDN (+ WSL.OICR refinement)."""
    import detectron.modeling.wsl_heads as wsl_heads
    blob_frcn, dim_frcn = add_roi_box_head_func(model, blob_in, dim_in, spatial_scale_in)
    wsl_heads.add_wsl_outputs(model, blob_frcn, dim_frcn)
    return wsl_heads.add_wsl_losses(model) if model.train else None


def add_training_inputs(model, roidb=None, rank=0, world_size=1):
    """Attach the data loader; the executor pulls one device batch per iteration (the reference
    splices DequeueBlobs ops in front of the net)."""
    assert model.train, 'Training inputs can only be added to a trainable model'
    if roidb is not None:
        from detectron.roi_data.loader_wsl import RoIDataLoader
        model.roi_data_loader = RoIDataLoader(
            roidb, num_loaders=cfg.DATA_LOADER.NUM_THREADS,
            minibatch_queue_size=cfg.DATA_LOADER.MINIBATCH_QUEUE_SIZE,
            blobs_queue_capacity=cfg.DATA_LOADER.BLOBS_QUEUE_CAPACITY, rank=rank,
            world_size=world_size)
    model.input_blob_names = roi_data_minibatch.get_minibatch_blob_names(is_training=True)


def add_inference_inputs(model):
    model.input_blob_names = roi_data_minibatch.get_minibatch_blob_names(is_training=False)
