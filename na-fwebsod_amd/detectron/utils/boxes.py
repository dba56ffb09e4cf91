"""Box helpers of the data path (reference: detectron/utils/boxes.py:66-131, and
detectron/utils/cython_bbox.pyx `bbox_overlaps` restated in numpy: +1 pixel areas)."""
import numpy as np


def unique_boxes(boxes, scale=1.0):
    """Indices of unique boxes (first occurrence), ascending (boxes.py:66-71)."""
    v = np.array([1, 1e3, 1e6, 1e9])
    hashes = np.round(boxes * scale).dot(v).astype(np.int64)
    _, index = np.unique(hashes, return_index=True)
    return np.sort(index)


def filter_small_boxes(boxes, min_size):
    """Keep boxes with width and height both greater than min_size (boxes.py:108-113)."""
    w = boxes[:, 2] - boxes[:, 0] + 1
    h = boxes[:, 3] - boxes[:, 1] + 1
    return np.where((w > min_size) & (h > min_size))[0]


def xywh_to_xyxy(xywh):
    if isinstance(xywh, (list, tuple)):
        assert len(xywh) == 4
        x1, y1 = xywh[0], xywh[1]
        return (x1, y1, x1 + np.maximum(0., xywh[2] - 1.), y1 + np.maximum(0., xywh[3] - 1.))
    if isinstance(xywh, np.ndarray):
        return np.hstack((xywh[:, 0:2], xywh[:, 0:2] + np.maximum(0, xywh[:, 2:4] - 1)))
    raise TypeError('Argument xywh must be a list, tuple, or numpy array.')


def xyxy_to_xywh(xyxy):
    if isinstance(xyxy, (list, tuple)):
        assert len(xyxy) == 4
        return (xyxy[0], xyxy[1], xyxy[2] - xyxy[0] + 1, xyxy[3] - xyxy[1] + 1)
    if isinstance(xyxy, np.ndarray):
        return np.hstack((xyxy[:, 0:2], xyxy[:, 2:4] - xyxy[:, 0:2] + 1))
    raise TypeError('Argument xyxy must be a list, tuple, or numpy array.')


def clip_xyxy_to_image(x1, y1, x2, y2, height, width):
    x1 = np.minimum(width - 1., np.maximum(0., x1))
    y1 = np.minimum(height - 1., np.maximum(0., y1))
    x2 = np.minimum(width - 1., np.maximum(0., x2))
    y2 = np.minimum(height - 1., np.maximum(0., y2))
    return x1, y1, x2, y2


def bbox_overlaps(boxes, query_boxes):
    """[N,K] IoU with +1 pixel areas (cython_bbox.pyx:27-63), bit-identical to the compiled
    reference (tests/test_datasets.py against oracle/_ref): coordinate differences are float32,
    the `+ 1` and the area products are evaluated in double (Cython emits the literal as 1.0),
    box_area / iw / ih are stored as float32, the union `float(...)` is a double expression
    stored as float32, and iw*ih/ua is float32."""
    b = np.asarray(boxes, np.float32)
    q = np.asarray(query_boxes, np.float32)
    out = np.zeros((b.shape[0], q.shape[0]), np.float32)
    if b.shape[0] == 0 or q.shape[0] == 0:
        return out
    f64 = np.float64
    qa = (((q[:, 2] - q[:, 0]).astype(f64) + 1.0) * ((q[:, 3] - q[:, 1]).astype(f64) + 1.0)).astype(np.float32)
    barea = ((b[:, 2] - b[:, 0]).astype(f64) + 1.0) * ((b[:, 3] - b[:, 1]).astype(f64) + 1.0)
    iw = ((np.minimum(b[:, None, 2], q[None, :, 2]) - np.maximum(b[:, None, 0], q[None, :, 0])).astype(f64)
          + 1.0).astype(np.float32)
    ih = ((np.minimum(b[:, None, 3], q[None, :, 3]) - np.maximum(b[:, None, 1], q[None, :, 1])).astype(f64)
          + 1.0).astype(np.float32)
    ok = (iw > 0) & (ih > 0)
    inter = (np.where(ok, iw, 0).astype(np.float32) * np.where(ok, ih, 0).astype(np.float32))
    ua = (barea[:, None] + qa.astype(f64)[None, :] - inter.astype(f64)).astype(np.float32)
    out[ok] = (inter / ua)[ok]
    return out


def crowd_iou(boxes_xywh, crowd_xywh):
    """pycocotools mask.iou(d, g, iscrowd=[1..]) for boxes: inter / area(d) (continuous
    coordinates, no +1) - the rule _filter_crowd_proposals uses (json_dataset_wsl.py:703-720)."""
    d = np.asarray(boxes_xywh, np.float64)
    g = np.asarray(crowd_xywh, np.float64)
    iw = np.minimum(d[:, None, 0] + d[:, None, 2], g[None, :, 0] + g[None, :, 2]) - \
        np.maximum(d[:, None, 0], g[None, :, 0])
    ih = np.minimum(d[:, None, 1] + d[:, None, 3], g[None, :, 1] + g[None, :, 3]) - \
        np.maximum(d[:, None, 1], g[None, :, 1])
    inter = np.maximum(iw, 0) * np.maximum(ih, 0)
    area = (d[:, 2] * d[:, 3])[:, None]
    return np.where(area > 0, inter / np.maximum(area, 1e-30), 0.0)
