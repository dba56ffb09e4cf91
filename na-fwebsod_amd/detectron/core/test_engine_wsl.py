"""Test engine: inference of a trained model over a dataset, in one process or as one FRESH child
process per GPU over image ranges (reference: detectron/core/test_engine_wsl.py:70-352).

Same entry points and file formats as the reference:
  run_inference(weights_file, ind_range=None, multi_gpu_testing=False, gpu_id=0)
  test_net_on_dataset / multi_gpu_test_net_on_dataset / test_net
  detections.pkl / detection_range_<s>_<e>.pkl =
      {all_boxes[cls][image] = N x 5, all_segms[cls][image] = [], all_keyps[cls][image] = [],
       cfg = yaml of the cfg tree}

The parent of a multi-GPU run never touches the GPU: it counts the images, starts the children
(detectron/utils/subprocess.py) and collates their range files.  Dataset evaluation
(task_evaluation.evaluate_all) is outside the hot path: the returned results hold
the detection counts only.  Datasets that are not on disk fall back to the seeded synthetic roidb
(cfg.NAWS.SYNTHETIC_TEST_IMAGES entries) so that the whole chain runs on a box without data."""
import logging
import os

import numpy as np

from detectron.core.config import cfg, get_output_dir
from detectron.utils.net_wsl import save_object
import detectron.utils.env as envu
import detectron.utils.subprocess as subprocess_utils

logger = logging.getLogger(__name__)

SYNTHETIC = 'synthetic'


def get_inference_dataset(index, is_parent=True):
    """(dataset name, proposal file) of TEST.DATASETS[index] (reference :50-67)."""
    assert is_parent or len(cfg.TEST.DATASETS) <= 1, \
        'The child inference process can only work on a single dataset'
    if not len(cfg.TEST.DATASETS):
        return SYNTHETIC, None
    dataset_name = cfg.TEST.DATASETS[index]
    if cfg.TEST.PRECOMPUTED_PROPOSALS and len(cfg.TEST.PROPOSAL_FILES):
        assert is_parent or len(cfg.TEST.PROPOSAL_FILES) == 1, \
            'The child inference process can only work on a single proposal file'
        assert len(cfg.TEST.PROPOSAL_FILES) == len(cfg.TEST.DATASETS), \
            'If proposals are used, one proposal file must be specified for each dataset'
        return dataset_name, cfg.TEST.PROPOSAL_FILES[index]
    return dataset_name, None


def dataset_on_disk(dataset_name):
    from detectron.datasets import dataset_catalog
    return dataset_name != SYNTHETIC and dataset_catalog.contains(dataset_name) and \
        os.path.exists(dataset_catalog.get_ann_fn(dataset_name))


def num_dataset_images(dataset_name):
    """len(dataset.get_roidb()) without proposals (reference :134) - host work only."""
    if dataset_on_disk(dataset_name):
        from detectron.datasets.json_dataset_wsl import JsonDataset
        return len(JsonDataset(dataset_name).get_roidb())
    return int(cfg.NAWS.SYNTHETIC_TEST_IMAGES)


def get_roidb_and_dataset(dataset_name, proposal_file, ind_range):
    """-> (roidb[start:end], real, start, end, total) (reference :354-379)."""
    real = dataset_on_disk(dataset_name)
    if real:
        from detectron.datasets.json_dataset_wsl import JsonDataset
        roidb = JsonDataset(dataset_name).get_roidb(proposal_file=proposal_file,
                                                    proposal_limit=cfg.TEST.PROPOSAL_LIMIT)
    else:
        from detectron.datasets import synthetic
        roidb = synthetic.make_roidb(int(cfg.NAWS.SYNTHETIC_TEST_IMAGES),
                                     min(cfg.TEST.PROPOSAL_LIMIT, 2000),
                                     cfg.MODEL.NUM_CLASSES - 1, seed=cfg.RNG_SEED)
    total = len(roidb)
    if ind_range is not None:
        start, end = ind_range
        roidb = roidb[start:end]
    else:
        start, end = 0, total
    return roidb, real, start, end, total


def empty_results(num_classes, num_images):
    """all_boxes / all_segms / all_keyps: [cls][image] lists (reference :382-394)."""
    all_boxes = [[[] for _ in range(num_images)] for _ in range(num_classes)]
    all_segms = [[[] for _ in range(num_images)] for _ in range(num_classes)]
    all_keyps = [[[] for _ in range(num_images)] for _ in range(num_classes)]
    return all_boxes, all_segms, all_keyps


def extend_results(index, all_res, im_res):
    """Class 0 (background) is skipped (reference :397-402)."""
    for cls_idx in range(1, len(im_res)):
        all_res[cls_idx][index] = im_res[cls_idx]


def check_weights_file(weights_file):
    """A non-empty TEST.WEIGHTS that does not exist is an error, as in the reference (its
    initialize_gpu_from_weights_file opens the file, utils/net_wsl.py:64-66): evaluating the
    randomly initialised model instead would write detections.pkl and exit 0.  Only
    TEST.WEIGHTS == '' selects the no-weights synthetic smoke path."""
    if weights_file and not os.path.exists(weights_file):
        raise FileNotFoundError('TEST.WEIGHTS {!r} does not exist (an empty TEST.WEIGHTS runs the '
                                'randomly initialised model on purpose)'.format(weights_file))


def run_inference(weights_file, ind_range=None, multi_gpu_testing=False, gpu_id=0,
                  check_expected_results=False):
    """Parent (ind_range None): every dataset of TEST.DATASETS, in this process or - with
    multi_gpu_testing - through one child per GPU.  Child (ind_range given): that range of the
    single dataset named on its command line (reference :70-122)."""
    check_weights_file(weights_file)         # before any child is started or any model is built
    is_parent = ind_range is None
    if is_parent:
        all_results = {}
        for i in range(max(1, len(cfg.TEST.DATASETS))):
            dataset_name, proposal_file = get_inference_dataset(i)
            output_dir = get_output_dir(dataset_name, training=False)
            all_results.update(test_net_on_dataset(weights_file, dataset_name, proposal_file,
                                                   output_dir, multi_gpu=multi_gpu_testing))
        return all_results
    dataset_name, proposal_file = get_inference_dataset(0, is_parent=False)
    output_dir = get_output_dir(dataset_name, training=False)
    return test_net(weights_file, dataset_name, proposal_file, output_dir, ind_range=ind_range,
                    gpu_id=gpu_id)


def test_net_on_dataset(weights_file, dataset_name, proposal_file, output_dir, multi_gpu=False,
                        gpu_id=0):
    import time
    t0 = time.time()
    if multi_gpu:
        num_images = num_dataset_images(dataset_name)
        all_boxes, _segms, _keyps = multi_gpu_test_net_on_dataset(
            weights_file, dataset_name, proposal_file, num_images, output_dir)
    else:
        all_boxes, _segms, _keyps = test_net(weights_file, dataset_name, proposal_file, output_dir,
                                             gpu_id=gpu_id)
    logger.info('Total inference time: {:.3f}s'.format(time.time() - t0))
    n = sum(len(d) for c in all_boxes[1:] for d in c)
    # (the reference hands all_boxes to task_evaluation.evaluate_all here: out of scope)
    return {dataset_name: {'box': {'num_images': len(all_boxes[1]) if len(all_boxes) > 1 else 0,
                                   'num_detections': int(n)}}}


def multi_gpu_test_net_on_dataset(weights_file, dataset_name, proposal_file, num_images, output_dir):
    """cfg.NUM_GPUS children over np.array_split ranges, collated in range order into
    detections.pkl (reference :148-200).  No GPU work in this process."""
    binary = os.path.join(envu.get_runtime_dir(), 'test_net_wsl' + envu.get_py_bin_ext())
    assert os.path.exists(binary), 'Binary \'{}\' not found'.format(binary)
    opts = []
    if dataset_name != SYNTHETIC:
        opts += ['TEST.DATASETS', '("{}",)'.format(dataset_name)]
    opts += ['TEST.WEIGHTS', weights_file or '']
    if proposal_file:
        opts += ['TEST.PROPOSAL_FILES', '("{}",)'.format(proposal_file)]
    outputs = subprocess_utils.process_in_parallel('detection', num_images, binary, output_dir, opts)
    all_boxes = [[] for _ in range(cfg.MODEL.NUM_CLASSES)]
    all_segms = [[] for _ in range(cfg.MODEL.NUM_CLASSES)]
    all_keyps = [[] for _ in range(cfg.MODEL.NUM_CLASSES)]
    for det_data in outputs:
        for cls_idx in range(1, cfg.MODEL.NUM_CLASSES):
            all_boxes[cls_idx] += det_data['all_boxes'][cls_idx]
            all_segms[cls_idx] += det_data['all_segms'][cls_idx]
            all_keyps[cls_idx] += det_data['all_keyps'][cls_idx]
    det_file = os.path.join(output_dir, 'detections.pkl')
    save_detections(det_file, all_boxes, all_segms, all_keyps)
    logger.info('Wrote detections to: {}'.format(os.path.abspath(det_file)))
    return all_boxes, all_segms, all_keyps


def save_detections(det_file, all_boxes, all_segms, all_keyps):
    """The reference's dict, its four keys and nothing else (test_engine_wsl.py:189-197,
    :297-305).  `cfg` holds the reference's own keys only - tools/reval.py merges it key by key
    and refuses unknown ones - in the AttrDict-tagged yaml its loader rebuilds."""
    save_object(dict(all_boxes=all_boxes, all_segms=all_segms, all_keyps=all_keyps,
                     cfg=envu.yaml_dump(cfg, reference_format=True)), det_file)


def initialize_model_from_cfg(weights_file, gpu_id=0):
    """Test-mode model + executor on cuda:<gpu_id> with the weights loaded (reference :308-333)."""
    import torch
    from detectron.core.executor import NetExecutor
    import detectron.modeling.model_builder_wsl as model_builder
    import detectron.utils.net_wsl as nu
    check_weights_file(weights_file)
    device = torch.device('cuda', int(gpu_id))
    torch.cuda.set_device(device)
    model = model_builder.create(cfg.MODEL.TYPE, train=False)
    ex = NetExecutor(model, device)
    ex.init_params()
    check_weights_file(weights_file)
    if weights_file:
        nu.initialize_from_weights_file(model, weights_file, ex, broadcast=False)
    return model, ex


def test_net(weights_file, dataset_name, proposal_file, output_dir, ind_range=None, gpu_id=0,
             printer=print):
    """All images, or [start, end) of them, on one GPU (reference :203-306).  A range file's
    lists hold the range's images only; the parent concatenates them."""
    from detectron.core import test_wsl
    roidb, real, start_ind, end_ind, total = get_roidb_and_dataset(dataset_name, proposal_file,
                                                                  ind_range)
    _model, ex = initialize_model_from_cfg(weights_file, gpu_id=gpu_id)
    num_classes = cfg.MODEL.NUM_CLASSES
    all_boxes, all_segms, all_keyps = empty_results(num_classes, len(roidb))
    if real:
        from detectron.roi_data.minibatch_wsl import _read_image
    else:
        from detectron.datasets import synthetic
    for i, e in enumerate(roidb):
        if real:
            im = _read_image(e).astype(np.float32)
            sel = e['gt_classes'] == 0       # proposals only (reference :232-233)
        else:
            im = (synthetic.make_image(e).transpose(1, 2, 0) + synthetic.PIXEL_MEANS_BGR).astype(np.float32)
            sel = slice(None)
        if e['boxes'][sel].shape[0] == 0:    # reference :234-235: the image keeps its empty lists
            printer('image %d: no proposals' % (start_ind + i))
            continue
        cls_boxes = test_wsl.im_detect_all(ex, im, e['boxes'][sel], e['obn_scores'][sel])
        extend_results(i, all_boxes, cls_boxes)
        n_det = sum(len(cls_boxes[j]) for j in range(1, num_classes))
        top = max([cls_boxes[j][:, 4].max() for j in range(1, num_classes) if len(cls_boxes[j])] or [0.0])
        printer('im_detect: range [%d, %d] of %d: %d/%d: %d proposals -> %d detections, top score %.4g'
                % (start_ind + 1, end_ind, total, start_ind + i + 1, start_ind + len(roidb),
                   e['boxes'][sel].shape[0], n_det, top))
    det_name = 'detection_range_%s_%s.pkl' % tuple(ind_range) if ind_range is not None \
        else 'detections.pkl'
    det_file = os.path.join(output_dir, det_name)
    save_detections(det_file, all_boxes, all_segms, all_keyps)
    logger.info('Wrote detections to: {}'.format(os.path.abspath(det_file)))
    printer('Wrote detections to: {}'.format(os.path.abspath(det_file)))
    del ex
    return all_boxes, all_segms, all_keyps
