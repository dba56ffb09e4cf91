"""Graph-level plugin surface on the GPU: the builders + NetExecutor, fused plan vs op-by-op
plan on the same blobs and inputs, the train CLI and the weight-file round trip."""
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


def _inputs(dev, n_rois=16):
    from detectron.datasets import synthetic
    mb = synthetic.make_minibatch(synthetic.make_roidb(1, n_rois, 20, 64, 96, seed=5), 20)
    return {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}


def _build(dev, train, interpreted, blobs=None):
    import detectron.modeling.model_builder_wsl as mb
    from detectron.core.executor import NetExecutor
    model = mb.create('generalized_wsl', train=train)
    ex = NetExecutor(model, dev, force_interpreted=interpreted, disable_dropout=True)
    if blobs is None:
        ex.init_params(seed=7)
    else:
        ex.load_blobs(blobs)
    return model, ex


def test_fused_plan_equals_op_by_op_plan(dev, cfgmod):
    from detectron.datasets import synthetic
    blobs = synthetic.init_blobs(20, seed=3)
    m1, fused = _build(dev, True, False, blobs)
    m2, interp = _build(dev, True, True, blobs)
    assert fused.plan == 'fused' and interp.plan == 'interpreted'
    assert len(m1.net.ops) == 103 and len(m1.TrainableParams()) == 16
    assert len(m1.update_ops) == 16 and m1.losses == ['loss_cls', 'loss_cls_noise']
    t = _inputs(dev)
    for ex, m in ((fused, m1), (interp, m2)):
        m.UpdateWorkspaceLr(0, 1e-3)
        ex.feed(t)
        ex.run()
    for k in ('loss_cls', 'loss_cls_noise'):
        a, b = float(fused.fetch(k).reshape(-1)[0]), float(interp.fetch(k).reshape(-1)[0])
        assert abs(a - b) <= 1e-4 * abs(b), (k, a, b)
    np.testing.assert_allclose(fused.fetch('cls_prob').reshape(-1).cpu().numpy(),
                               interp.fetch('cls_prob').reshape(-1).cpu().numpy(), rtol=1e-4)
    np.testing.assert_allclose(fused.fetch('rois_class_weight').reshape(-1).cpu().numpy(),
                               interp.fetch('rois_class_weight').reshape(-1).cpu().numpy(),
                               rtol=1e-4, atol=1e-6)
    fb, ib = fused.blobs(), interp.blobs()
    for name in m1.TrainableParams():
        a, b = fb[name].cpu().numpy(), ib[name].cpu().numpy()
        upd = np.abs(b - blobs[name].numpy()).max()
        assert np.abs(a - b).max() <= 2e-3 * upd + 1e-9, name          # the applied update agrees
    # conv body is frozen: untouched by training
    assert torch.equal(fb['conv3_2_w'].cpu(), blobs['conv3_2_w'])


def test_inference_plans_agree(dev, cfgmod):
    from detectron.datasets import synthetic
    blobs = synthetic.init_blobs(20, seed=3)
    _m1, fused = _build(dev, False, False, blobs)
    _m2, interp = _build(dev, False, True, blobs)
    t = _inputs(dev)
    for ex in (fused, interp):
        ex.feed({k: t[k] for k in ('data', 'rois', 'obn_scores')})
        ex.run()
    a, b = fused.fetch('cls_prob').cpu().numpy(), interp.fetch('cls_prob').cpu().numpy()
    assert a.shape == (16, 21)
    np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-9)
    np.testing.assert_array_equal(a[:, 0], a[:, 1])       # background = copy of first fg column


def test_other_graph_falls_back_to_op_by_op(dev, cfgmod):
    """A head that is not the na_wsddn graph still runs (operator API), just not fused."""
    import detectron.modeling.model_builder_wsl as mb
    from detectron.core.executor import NetExecutor
    cfgmod.cfg.WEBLY.ENTROPY = False          # unweighted CrossEntropyWithLogits instead
    model = mb.create('generalized_wsl', train=True)
    ex = NetExecutor(model, dev, disable_dropout=True)
    assert ex.plan == 'interpreted'
    ex.init_params(seed=1)
    model.UpdateWorkspaceLr(0, 1e-3)
    ex.feed(_inputs(dev))
    ex.run()
    assert np.isfinite(float(ex.fetch('loss_cls')))


def test_weights_file_roundtrip_and_alias(dev, cfgmod, tmp_path):
    import detectron.utils.net_wsl as nu
    from detectron.datasets import synthetic
    blobs = synthetic.init_blobs(20, seed=3)
    model, ex = _build(dev, True, False, blobs)
    f = str(tmp_path / 'model_iter9.pkl')
    nu.save_model_to_weights_file(f, model, ex)
    saved = nu.load_object(f)
    assert set(saved) == {'blobs', 'cfg'} and saved['blobs']['fc6_w'].shape == (4096, 25088)
    assert '_[noisy]_fc6_w_momentum' in saved['blobs']
    # a VGG-style file without the noisy branch: '_[noisy]_fc6_w' is initialised from 'fc6_w'
    src = {k: v for k, v in saved['blobs'].items() if 'noisy' not in k and 'momentum' not in k}
    src['fc1000_w'] = np.zeros((4, 4), np.float32)
    nu.save_object({'blobs': src}, str(tmp_path / 'vgg.pkl'))
    model2, ex2 = _build(dev, True, False, None)
    nu.initialize_from_weights_file(model2, str(tmp_path / 'vgg.pkl'), ex2, broadcast=False)
    b2 = ex2.blobs(False)
    assert torch.equal(b2['_[noisy]_fc6_w'].cpu(), blobs['fc6_w'])
    assert torch.equal(b2['conv1_1_w'].cpu(), blobs['conv1_1_w'])
    assert '__preserve__/fc1000_w' in model2.preserved_blobs


@pytest.mark.parametrize('plan', [(), ('NAWS.MFMA_DTYPE', 'bf16', 'NUM_GPUS', '1', 'TRAIN.IMS_PER_BATCH', '2',
                                       'WEBLY.BAGGING_MIXUP', 'False')])
def test_train_cli_two_iterations(dev, cfgmod, tmp_path, capsys, plan):
    """The training tool end to end; second case: the bf16 plan (configs[3]'s arithmetic) with two
    images per process - per-image conv chains, RoIPoolF writing fc6's operand, fc6_w updated in its
    wgrad epilogue, the deferred kernel for the rest."""
    import importlib.util
    cfgmod.reset_cfg()
    spec = importlib.util.spec_from_file_location(
        'train_net_wsl', os.path.join(ROOT, 'na-fwebsod_amd', 'tools', 'train_net_wsl.py'))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    tool.main(['--cfg', YAML, '--skip-test', '--max-iter', '2', 'OUTPUT_DIR', str(tmp_path),
               'TRAIN.SCALES', '(64,)', 'TRAIN.MAX_SIZE', '96', 'TRAIN.BATCH_SIZE_PER_IM', '32',
               'WSL.USE_DISTORTION', 'False', 'DATA_LOADER.NUM_THREADS', '1',
               'SOLVER.BASE_LR', '1e-5'] + list(plan))
    out = capsys.readouterr().out
    assert 'json_stats: {' in out and '"loss_cls"' in out
    assert 'class_weight       Stat #iter_: 1' in out
    assert os.path.exists(os.path.join(str(tmp_path), 'train', 'flickr_voc', 'generalized_wsl',
                                       'model_final.pkl'))


def test_tta_inference_and_nms(dev, cfgmod):
    """BASELINE configs[4] shape of work, scaled down: multi-scale + flip TTA, AVG scores, NMS."""
    from detectron.core import test_wsl
    from detectron.datasets import synthetic
    c = cfgmod
    c.merge_cfg_from_list(['TEST.SCALE', 64, 'TEST.MAX_SIZE', 200, 'TEST.BBOX_AUG.ENABLED', True,
                           'TEST.BBOX_AUG.SCALES', '(48, 80)', 'TEST.BBOX_AUG.MAX_SIZE', 200,
                           'TEST.DETECTIONS_PER_IM', 20])
    blobs = synthetic.init_blobs(20, seed=3)
    _m, ex = _build(dev, False, False, blobs)
    e = synthetic.make_roidb(1, 40, 20, 64, 96, seed=9)[0]
    e['boxes'][1] = e['boxes'][0]                       # a duplicate proposal: dedup + scatter back
    im = (synthetic.make_image(e).transpose(1, 2, 0) + synthetic.PIXEL_MEANS_BGR).astype(np.float32)
    scores, boxes = test_wsl.im_detect_bbox_aug(ex, im, e['boxes'], e['obn_scores'])
    assert scores.shape == (40, 21) and np.isfinite(scores).all()
    s1, _ = test_wsl.im_detect_bbox(ex, im, 64, 200, e['boxes'], e['obn_scores'])
    # (boxes come back class-tiled, [n, 4K], as the reference's im_detect_bbox returns them)
    assert np.array_equal(boxes, np.tile(e['boxes'], (1, 21))) and not np.allclose(scores, s1)   # 6 passes averaged
    cls_boxes = test_wsl.im_detect_all(ex, im, e['boxes'], e['obn_scores'])
    assert len(cls_boxes) == 21 and sum(len(b) for b in cls_boxes[1:]) <= 20
    # the image blob prepared on the GPU (default) == prepared on the host, plain and mirrored
    assert c.cfg.NAWS.DEVICE_PREP
    c.cfg.NAWS.DEVICE_PREP = False
    s_host, _ = test_wsl.im_detect_bbox(ex, im, 64, 200, e['boxes'], e['obn_scores'])
    f_host, _ = test_wsl.im_detect_bbox_hflip(ex, im, 80, 200, e['boxes'], e['obn_scores'])
    c.cfg.NAWS.DEVICE_PREP = True
    f_dev, _ = test_wsl.im_detect_bbox_hflip(ex, im, 80, 200, e['boxes'], e['obn_scores'])
    assert np.array_equal(s1, s_host) and np.array_equal(f_dev, f_host)
    # a scale's plain + mirrored pass as one batch of two images == the two separate passes,
    # and the whole TTA result is unchanged by the pairing
    p0, p1 = test_wsl.im_detect_bbox_pair(ex, im, 80, 200, e['boxes'], e['obn_scores'])
    q0, _ = test_wsl.im_detect_bbox(ex, im, 80, 200, e['boxes'], e['obn_scores'])
    np.testing.assert_allclose(p0, q0, rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(p1, f_dev, rtol=1e-5, atol=1e-9)
    assert c.cfg.NAWS.TTA_PAIR_FLIPS
    c.cfg.NAWS.TTA_PAIR_FLIPS = False
    scores_seq, _ = test_wsl.im_detect_bbox_aug(ex, im, e['boxes'], e['obn_scores'])
    c.cfg.NAWS.TTA_PAIR_FLIPS = True
    np.testing.assert_allclose(scores, scores_seq, rtol=1e-5, atol=1e-9)
    # dedup hash: rows 0 and 1 collapse to one roi in the forward pass
    rois = np.hstack((np.zeros((40, 1), np.float32), e['boxes'])).astype(np.float32)
    u, idx, inv = test_wsl.dedup_rois(rois, 0.125)
    # (the 1/8-px hash grid also merges proposals that differ by a few pixels)
    assert u.shape[0] <= 39 and np.array_equal(np.round(u[inv] * 0.125), np.round(rois * 0.125))
    assert test_wsl.nms(np.array([[0, 0, 10, 10, .9], [1, 1, 10, 10, .8], [20, 20, 30, 30, .7]],
                                 np.float32), 0.5) == [0, 2]


def _load_tool(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        name, os.path.join(ROOT, 'na-fwebsod_amd', 'tools', name + '.py'))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    return tool


def test_cli_on_a_dataset_on_disk(dev, cfgmod, tmp_path, capsys, monkeypatch):
    """The whole caller chain on real files: COCO json + MCG pickle -> roidb -> loader threads
    (PNG decode) -> device-side image prep with the yaml's HSV distortion / crop / flip ->
    training iterations -> checkpoint -> the POST-TRAINING TEST of the final checkpoint
    (reference tools/train_net_wsl.py:118-160: test_model -> run_inference; dataset -> multi-scale
    + flip TTA -> GPU NMS -> detections.pkl in the reference's schema);  then test_net_wsl alone,
    in one process and with --multi-gpu-testing as a parent of two fresh children that share
    cuda:0 (HIP_VISIBLE_DEVICES=0,0): the same detections every way."""
    from test_datasets import _toy_dataset
    from detectron.datasets import dataset_catalog
    from detectron.utils.net_wsl import load_object
    cfgmod.reset_cfg()
    imdir, annf, pf, sizes, _pb, _ps = _toy_dataset(tmp_path)
    dataset_catalog.register('toy_train', imdir, annf)
    dataset_catalog.register('toy_test', imdir, annf)
    common = ['OUTPUT_DIR', str(tmp_path), 'MODEL.NUM_CLASSES', '3', 'DATA_LOADER.NUM_THREADS', '1',
              'TRAIN.CROWD_FILTER_THRESH', '0.0']
    test_opts = ['TEST.DATASETS', "('toy_test',)", 'TEST.PROPOSAL_FILES', "('%s',)" % pf,
                 'TEST.SCALE', '64', 'TEST.MAX_SIZE', '120', 'TEST.BBOX_AUG.ENABLED', 'True',
                 'TEST.BBOX_AUG.SCALES', '(48, 64)', 'TEST.BBOX_AUG.MAX_SIZE', '120',
                 'TEST.BBOX_AUG.H_FLIP', 'True', 'TEST.BBOX_AUG.SCALE_H_FLIP', 'True']
    res = _load_tool('train_net_wsl').main(
        ['--cfg', YAML, '--max-iter', '3'] + common + test_opts +
        ['TRAIN.DATASETS', "('toy_train',)", 'TRAIN.PROPOSAL_FILES', "('%s',)" % pf,
         'TRAIN.SCALES', '(64, 80)', 'TRAIN.MAX_SIZE', '120', 'SOLVER.BASE_LR', '1e-5',
         'NAWS.DEVICE_PREP', 'True'])
    out = capsys.readouterr().out
    import json as _json
    stats = [_json.loads(l.split('json_stats: ', 1)[1]) for l in out.splitlines() if 'json_stats: {' in l]
    assert stats and all(np.isfinite(float(st['loss'])) for st in stats)
    # (the Stat lines print nan for "bg" when an image carries every class: AI/AL = 0/0, as the
    # reference's stat_op.cu:68-74 does)
    wts = os.path.join(str(tmp_path), 'train', 'toy_train', 'generalized_wsl', 'model_final.pkl')
    assert os.path.exists(wts)
    det_file = os.path.join(str(tmp_path), 'test', 'toy_test', 'generalized_wsl', 'detections.pkl')
    assert 'reprint snapshot name for the result:' in out and os.path.exists(det_file)
    assert res['toy_test']['box']['num_images'] == len(sizes)
    post = load_object(det_file)
    assert set(post) == {'all_boxes', 'all_segms', 'all_keyps', 'cfg'}
    assert post['cfg'].startswith('!!python/object/new:detectron.utils.collections.AttrDict')
    os.remove(det_file)
    cfgmod.reset_cfg()
    all_boxes = _load_tool('test_net_wsl').main(['--cfg', YAML] + common + test_opts + ['TEST.WEIGHTS', wts])
    det = load_object(det_file)
    assert len(det['all_boxes']) == 3 and len(det['all_boxes'][1]) == len(sizes)
    n = 0
    for j in (1, 2):
        for i in range(len(sizes)):
            d = np.asarray(det['all_boxes'][j][i])
            if d.size == 0 and d.ndim == 1:
                # an image without proposals keeps the empty lists of empty_results, as in the
                # reference (test_engine_wsl.py:234-235 `continue`s before extend_results)
                assert det['all_boxes'][j][i] == [] and post['all_boxes'][j][i] == []
                continue
            assert d.ndim == 2 and d.shape[1] == 5 and np.isfinite(d).all()
            assert np.array_equal(d, np.asarray(all_boxes[j][i]))
            assert np.array_equal(d, np.asarray(post['all_boxes'][j][i])), (j, i)
            assert det['all_segms'][j][i] == [] and det['all_keyps'][j][i] == []
            n += d.shape[0]
    assert n > 0
    # --multi-gpu-testing: this process only starts the children and collates their range files
    os.remove(det_file)
    cfgmod.reset_cfg()
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0,0')
    multi = _load_tool('test_net_wsl').main(
        ['--cfg', YAML, '--multi-gpu-testing'] + common + test_opts + ['TEST.WEIGHTS', wts, 'NUM_GPUS', '2'])
    monkeypatch.delenv('HIP_VISIBLE_DEVICES')
    out_dir = os.path.dirname(det_file)
    ranges = sorted(f for f in os.listdir(out_dir) if f.startswith('detection_range_') and f.endswith('.pkl'))
    assert len(ranges) == 2, ranges
    det2 = load_object(det_file)
    for j in (1, 2):
        assert len(det2['all_boxes'][j]) == len(sizes)
        for i in range(len(sizes)):
            assert np.array_equal(np.asarray(det2['all_boxes'][j][i]), np.asarray(det['all_boxes'][j][i])), (j, i)
            assert np.array_equal(np.asarray(multi[j][i]), np.asarray(det['all_boxes'][j][i]))


def test_min_entropy_loss_op_and_graph(dev, cfgmod):
    """cfg.WSL.MIN_ENTROPY_LOSS: the HIP op against the oracle's
    restatement of min_entropy_loss_op.cc, and inside the graph (op-by-op plan): loss_entropy of
    the run equals the oracle on the fetched rois_pred, and the extra gradient that reaches the
    fc8c / fc8d logits equals autograd of 0.1 * loss_entropy through the two softmaxes."""
    from naws_hip import ops
    from oracle import oracle
    from detectron.datasets import synthetic
    rng = np.random.default_rng(61)
    x = rng.uniform(0, 1, (300, 20)).astype(np.float32) ** 6
    x[0, :3] = [0.0, 1.0, 1e-30]
    lab = np.zeros((1, 20), np.float32)
    lab[0, [0, 1, 2, 7]] = [1.0, 0.7, 0.5, 0.49]               # mixup-style fractional labels
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(lab).to(dev)
    y = float(ops.min_entropy_loss(xd, ld))
    assert abs(y - float(oracle.min_entropy_loss(x, lab))) <= 1e-5 * abs(y)
    g = ops.min_entropy_loss_grad(xd, ld, torch.tensor([0.1], device=dev)).cpu().numpy()
    np.testing.assert_allclose(g, oracle.min_entropy_loss_grad(x, lab, 0.1), rtol=1e-5, atol=1e-9)
    # ---- in the graph
    blobs = synthetic.init_blobs(20, seed=3)
    t = _inputs(dev)
    res = {}
    for flag in (False, True):
        cfgmod.cfg.WSL.MIN_ENTROPY_LOSS = flag
        m, ex = _build(dev, True, True, blobs)
        assert ex.plan == 'interpreted' and (('loss_entropy' in m.losses) == flag)
        m.UpdateWorkspaceLr(0, 1e-3)
        ex.feed(t)
        ex.run()
        res[flag] = {k: ex.fetch(k).clone() for k in ('fc8c_grad', 'fc8d_grad', 'rois_pred',
                                                        'fc8c', 'fc8d', 'noisy_fc8c_grad')}
        if flag:
            le = float(ex.fetch('loss_entropy').reshape(-1)[0])
    cfgmod.cfg.WSL.MIN_ENTROPY_LOSS = False
    rp = res[True]['rois_pred'].cpu().numpy()
    lab_oh = t['labels_oh'].cpu().numpy()
    assert abs(le - float(oracle.min_entropy_loss(rp, lab_oh))) <= 1e-5 * abs(le)
    # autograd (float64, CPU) of 0.1 * H(rois_pred) w.r.t. the clean logits
    a = res[True]['fc8c'].double().cpu().requires_grad_(True)
    b = res[True]['fc8d'].double().cpu().requires_grad_(True)
    p = torch.softmax(a, 1) * torch.softmax(b, 0)
    sel = torch.from_numpy(lab_oh[0] >= 0.5)
    pe = p[:, sel].clamp_min(1e-20)
    loss = 0.1 * (-(pe * pe.log()).sum() / pe.numel())
    da, db = torch.autograd.grad(loss, [a, b])
    for k, d in (('fc8c_grad', da), ('fc8d_grad', db)):
        extra = (res[True][k] - res[False][k]).double().cpu()
        assert float((extra - d).abs().max()) <= 1e-4 * float(d.abs().max()) + 1e-9, k
    # the noisy branch's own logits do not see the entropy term
    assert torch.equal(res[True]['noisy_fc8c_grad'], res[False]['noisy_fc8c_grad'])


def test_op_by_op_plan_reproduces_the_reference_momentum_correction(dev, cfgmod):
    """NetExecutor.update_lr of the op-by-op plan against the sequence captured from the imported
    reference (tests/golden/make_golden_lr_update.py; detector.py:509-559): same lr fed, momentum
    of every trainable parameter scaled - or not - by the same FLOAT32 quotient, bit for bit (the
    fused plan's form of this test is test_set_lr_reproduces_the_reference_momentum_correction)."""
    import json
    gold = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'reference_lr_update.json')))
    model, ex = _build(dev, True, True)
    assert ex.plan == 'interpreted'
    name = 'fc8c_w'
    g = torch.Generator(device='cpu').manual_seed(5)
    ref = torch.randn(ex.ws[name + '_momentum'].shape, generator=g)
    ex.ws[name + '_momentum'].copy_(ref.to(dev))
    want = ref.numpy().copy()
    for case in gold['cases']:
        assert np.float32(float(ex.lr.item())) == np.float32(case['cur_lr'])
        got = ex.update_lr(0, case['new_lr'])
        assert np.float32(got) == np.float32(case['returned'])
        if case['correction'] is not None:
            want = want * np.float32(case['correction'])
        assert np.array_equal(ex.ws[name + '_momentum'].cpu().numpy(), want), case
