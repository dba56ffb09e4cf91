"""the inference post-processing kernels (csrc/infer_ops.hip) against the numpy
path of the reference they replace (detectron/core/test_wsl.py:125-133, :173-176, :181-281,
:803-863): index outputs and kept sets bit-exact, scores bit-identical."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
YAML = os.path.join(ROOT, 'na-fwebsod_amd', 'configs', 'flickr_voc', 'na_wsddn_V-16-C5_1x.yaml')


@pytest.fixture
def cfgmod():
    from detectron.core import config as c
    c.reset_cfg()
    c.merge_cfg_from_file(YAML)
    c.merge_cfg_from_list(['NUM_GPUS', 1])
    yield c
    c.reset_cfg()


@pytest.mark.parametrize('n', [1, 37, 4000, 9999])
def test_roi_dedup_matches_numpy_unique(dev, cfgmod, n):
    from detectron.core import test_wsl
    from detectron.datasets import synthetic
    from naws_hip import ops
    rng = np.random.default_rng(n)
    w_im, h_im = 500, 375
    boxes = synthetic.make_boxes(rng, n, h_im, w_im)
    if n > 10:                     # exact duplicates and near twins that collide on the 1/8 grid
        boxes[5] = boxes[2]
        boxes[7] = boxes[3] + np.array([1, 0, 1, 0], np.float32)
        boxes[n // 2:n // 2 + n // 10] = boxes[:n // 10] + 1.0
    obn = rng.uniform(0, 1, (n, 1)).astype(np.float32)
    specs = [(688.0 / 375.0, w_im, False, 0.0), (688.0 / 375.0, w_im, True, 1.0),
             (1200.0 / 375.0, w_im, False, 0.0), (480.0 / 375.0, w_im, True, 0.0)]
    dd = ops.roi_dedup(torch.from_numpy(boxes).to(dev), torch.from_numpy(obn.reshape(-1)).to(dev),
                       specs, 0.125)
    counts = dd['count'].cpu().numpy()
    for p, (sc, w, fl, b) in enumerate(specs):
        bx = test_wsl.flip_boxes(boxes, w) if fl else boxes
        rois = test_wsl.project_rois(bx, sc)
        u, index, inv = test_wsl.dedup_rois(rois, 0.125)
        m = int(counts[p])
        assert m == u.shape[0], (p, m, u.shape)
        got = dd['rois'][p, :m].cpu().numpy()
        assert np.array_equal(got[:, 1:], u[:, 1:]) and (got[:, 0] == b).all()
        assert np.array_equal(dd['index'][p, :m].cpu().numpy(), index)
        assert np.array_equal(dd['inv'][p].cpu().numpy(), inv)
        assert np.array_equal(dd['obn'][p, :m].cpu().numpy(), (obn + 1.0).astype(np.float32)[index, 0])


def test_tta_mean_is_numpy_mean(dev):
    from naws_hip import ops
    rng = np.random.default_rng(0)
    n, k, npass = 4000, 21, 10
    passes, invs = [], []
    for p in range(npass):
        m = int(rng.integers(n // 2, n))
        s = (rng.uniform(0, 1, (m, k)) ** 8).astype(np.float32)
        inv = rng.integers(0, m, (n,)).astype(np.int32)
        passes.append(s)
        invs.append(inv)
    want = np.mean([s[i] for s, i in zip(passes, invs)], axis=0)
    acc = torch.empty((n, k), device=dev)
    for p in range(npass):
        ops.tta_accumulate(torch.from_numpy(passes[p]).to(dev), torch.from_numpy(invs[p]).to(dev),
                           acc, first=(p == 0))
    ops.tta_finish(acc, npass)
    assert want.dtype == np.float32 and np.array_equal(acc.cpu().numpy(), want)


@pytest.mark.parametrize('limit', [100, 0, 5])
def test_det_limit_matches_host_cut(dev, cfgmod, limit):
    from detectron.core import test_wsl
    from detectron.datasets import synthetic
    from naws_hip import ops
    c = cfgmod
    c.merge_cfg_from_list(['TEST.DETECTIONS_PER_IM', limit, 'NAWS.DEVICE_POST', False])
    rng = np.random.default_rng(limit)
    n, k = 600, 21
    boxes = synthetic.make_boxes(rng, n, 375, 500)
    scores = (rng.uniform(0, 1, (n, k)) ** 6).astype(np.float32)
    scores[rng.integers(0, n, 40), rng.integers(1, k, 40)] = np.float32(0.5)     # ties at a cut
    want = test_wsl.box_results_with_nms_and_limit(scores, boxes)[2]
    sd = torch.from_numpy(scores).to(dev)
    keep = ops.nms_per_class(torch.from_numpy(boxes).to(dev), sd[:, 1:].contiguous(),
                             c.cfg.TEST.SCORE_THRESH, c.cfg.TEST.NMS)
    cap = max(4 * limit, 1024) if limit else n * (k - 1)
    ints, sc = ops.det_limit(sd, keep, limit, cap)
    ints, sc = ints.cpu().numpy(), sc.cpu().numpy()
    cnt = int(ints[0])
    assert cnt == sum(len(b) for b in want[1:]) and cnt <= cap
    cls_i, row_i = ints[1:1 + cnt], ints[1 + cap:1 + cap + cnt]
    for j in range(1, k):
        m = cls_i == j
        assert np.array_equal(boxes[row_i[m]], want[j][:, :4]), j
        assert np.array_equal(sc[:cnt][m], want[j][:, 4]), j
    # a too-small buffer reports the full count
    ints2, _ = ops.det_limit(sd, keep, limit, 3)
    assert int(ints2[0]) == cnt


@pytest.mark.parametrize('aug', [True, False])
def test_im_detect_all_device_equals_host_path(dev, cfgmod, aug):
    """The whole per-image inference (10-pass TTA shape, scaled down) with the post-processing on
    the device == the numpy path: identical detections, bit for bit."""
    from detectron.core import test_wsl
    from detectron.core.executor import NetExecutor
    from detectron.datasets import synthetic
    import detectron.modeling.model_builder_wsl as mb
    c = cfgmod
    c.merge_cfg_from_list(['TEST.SCALE', 64, 'TEST.MAX_SIZE', 200, 'TEST.BBOX_AUG.ENABLED', aug,
                           'TEST.BBOX_AUG.SCALES', '(48, 80, 112)', 'TEST.BBOX_AUG.MAX_SIZE', 200,
                           'TEST.DETECTIONS_PER_IM', 30])
    model = mb.create('generalized_wsl', train=False)
    ex = NetExecutor(model, dev)
    ex.load_blobs(synthetic.init_blobs(20, seed=3))
    e = synthetic.make_roidb(1, 300, 20, 64, 96, seed=9)[0]
    e['boxes'][1] = e['boxes'][0]
    rng = np.random.default_rng(e['seed'])
    im = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    assert test_wsl.device_post_supported(ex, im)
    got = test_wsl.im_detect_all(ex, im, e['boxes'], e['obn_scores'])
    c.cfg.NAWS.DEVICE_POST = False
    assert not test_wsl.device_post_supported(ex, im)
    want = test_wsl.im_detect_all(ex, im, e['boxes'], e['obn_scores'])
    c.cfg.NAWS.DEVICE_POST = True
    assert len(got) == len(want) == 21 and sum(len(b) for b in want[1:]) > 0
    for j in range(1, 21):
        assert np.array_equal(got[j], want[j]), j
