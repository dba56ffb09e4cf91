"""VGG-16 conv5 body of the WSL path.  Mirrors detectron/modeling/VGG16.py:9-48
(`add_VGG16_conv5_body_origin`): thirteen 3x3 conv + in-place ReLU, k2s2 max-pools after
blocks 1-3, and with WSL.DILATION == 2 a stride-1 pool4 followed by dilation-2 conv5_x
(feature stride 8); FREEZE_AT == 2 stops gradients at pool2."""
from detectron.core.config import cfg

_BLOCKS = (
    (1, (3, 64, 64)),
    (2, (64, 128, 128)),
    (3, (128, 256, 256, 256)),
    (4, (256, 512, 512, 512)),
)


def _conv_relu(model, blob_in, name, dim_in, dim_out, pad, dilation):
    if dilation == 1:
        model.Conv(blob_in, name, dim_in, dim_out, 3, pad=pad, stride=1)
    else:
        model.Conv(blob_in, name, dim_in, dim_out, 3, pad=pad, stride=1, dilation=dilation)
    return model.Relu(name, name)


def add_VGG16_conv5_body_origin(model):
    blob = 'data'
    for idx, dims in _BLOCKS:
        for j in range(1, len(dims)):
            blob = _conv_relu(model, blob, 'conv%d_%d' % (idx, j), dims[j - 1], dims[j], 1, 1)
        if idx < 4:
            blob = model.MaxPool(blob, 'pool%d' % idx, kernel=2, pad=0, stride=2)
            if idx == 2 and cfg.TRAIN.FREEZE_AT == 2:
                model.StopGradient(blob, blob)
    dilated = cfg.WSL.DILATION == 2
    blob = model.MaxPool(blob, 'pool4', kernel=2, pad=0, stride=1 if dilated else 2)
    d = 2 if dilated else 1
    for j in (1, 2, 3):
        blob = _conv_relu(model, blob, 'conv5_%d' % j, 512, 512, d, d)
    return blob, 512, 1. / 8. if dilated else 1. / 16.
