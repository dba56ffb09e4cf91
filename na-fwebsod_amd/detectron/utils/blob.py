"""Blob helpers used by the builders (reference: detectron/utils/blob.py:167-173)."""


def get_loss_gradients(model, loss_blobs):
    """A gradient seed of 1.0 per loss blob (NOT scaled by 1/NUM_GPUS: the SGD op divides)."""
    out = {}
    for b in loss_blobs:
        g = model.net.ConstantFill(b, [str(b) + '_grad'], value=1.0)
        out[str(b)] = str(g)
    return out
