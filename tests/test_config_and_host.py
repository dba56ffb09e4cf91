"""cfg plugin surface, lr schedule, loader host logic — against fixtures captured from the
imported reference Python (tests/golden/reference_python.json)."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
YAML = os.path.join(ROOT, 'na-fwebsod_amd', 'configs', 'flickr_voc', 'na_wsddn_V-16-C5_1x.yaml')
GOLD = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'reference_python.json')))


@pytest.fixture
def cfgmod():
    from detectron.core import config as c
    c.reset_cfg()
    yield c
    c.reset_cfg()


def test_yaml_merge_matches_reference_cfg(cfgmod):
    c = cfgmod
    c.merge_cfg_from_file(YAML)
    c.merge_cfg_from_list(['NUM_GPUS', 4])

    def walk(gold, node, path):
        for k, v in gold.items():
            if k not in node:
                continue              # key outside the hot path: not declared here
            if isinstance(v, dict):
                walk(v, node[k], path + [k])
            else:
                mine = node[k]
                if isinstance(mine, np.ndarray):
                    mine = mine.tolist()
                if isinstance(mine, tuple):
                    mine = list(mine)
                if '.'.join(path + [k]) in ('OUTPUT_DIR',):
                    continue
                assert mine == v, ('.'.join(path + [k]), mine, v)
    walk(GOLD['cfg'], c.cfg, [])
    # every key of the yaml is accepted; the alias file is identical
    alias = YAML.replace('na_wsddn', 'webly_wsddn')
    assert open(alias).read() == open(YAML).read()


def test_cfg_behaviour(cfgmod):
    c = cfgmod
    with pytest.raises(KeyError):
        c.merge_cfg_from_cfg({'MODEL': {'NO_SUCH_KEY': 1}})
    with pytest.raises(ValueError):
        c.merge_cfg_from_list(['SOLVER.BASE_LR', 'fast'])
    with pytest.raises(AssertionError):
        c.merge_cfg_from_list(['SOLVER.NOPE', 1])
    c.merge_cfg_from_list(['TRAIN.SCALES', '(480, 600)', 'SOLVER.BASE_LR', 1.0, 'MODEL.TYPE', 'x'])
    assert c.cfg.TRAIN.SCALES == (480, 600) and c.cfg.SOLVER.BASE_LR == 1.0
    with pytest.raises(ValueError):          # an int is not a float in the reference's type check
        c.merge_cfg_from_list(['SOLVER.BASE_LR', 1])
    c.assert_and_infer_cfg()
    with pytest.raises(AttributeError):
        c.cfg.NUM_GPUS = 8
    with pytest.raises(AttributeError):
        c.cfg.TRAIN.SCALES = (1,)
    c.cfg.immutable(False)
    c.cfg.WSL.CSC = True
    with pytest.raises(NotImplementedError):
        c.assert_and_infer_cfg()


def test_lr_schedule_bits(cfgmod):
    c = cfgmod
    c.merge_cfg_from_file(YAML)
    from detectron.utils import lr_policy
    for it, bits in GOLD['lr_bits'].items():
        lr = lr_policy.get_lr_at_iter(int(it))
        assert lr.dtype == np.float32 and int(lr.view(np.uint32)) == bits, it
    c.cfg.SOLVER.WARM_UP_ITERS = 500
    assert abs(float(lr_policy.get_lr_at_iter(0)) - 1e-3 / 3) < 1e-9
    c.cfg.SOLVER.LR_POLICY = 'nope'
    with pytest.raises(NotImplementedError):
        lr_policy.get_lr_at_iter(0)


def test_blob_order_and_minibatch(cfgmod):
    c = cfgmod
    c.merge_cfg_from_file(YAML)
    c.merge_cfg_from_list(['WSL.USE_DISTORTION', False, 'TRAIN.SCALES', '(600,)',
                           'TRAIN.MAX_SIZE', 1000])
    from detectron.datasets import synthetic
    from detectron.roi_data import minibatch_wsl
    assert minibatch_wsl.get_minibatch_blob_names() == [
        'data', 'data_ids', 'rois', 'obn_scores', 'labels_int32', 'labels_oh']
    roidb = synthetic.make_roidb(1, 50, 20, 120, 200, seed=5)
    np.random.seed(3)
    blobs, valid = minibatch_wsl.get_minibatch(roidb, raw=False)
    assert valid and blobs['data'].dtype == np.float32 and blobs['data'].shape[1] == 3
    # short side scaled to 600 (cap 1000): 120x200 crop 0.9 -> 108x180 -> scale 5.555 -> 600x1000
    assert blobs['data'].shape[2:] == (600, 1000)
    assert blobs['rois'].shape == (50, 5) and blobs['rois'].dtype == np.float32
    assert blobs['obn_scores'].min() >= 1.0 and blobs['labels_oh'].sum() == 1
    assert blobs['rois'][:, 1:].min() >= 0 and blobs['rois'][:, 3].max() <= 1000
    c.cfg.WSL.USE_DISTORTION = True                 # the yaml's own setting: HSV jitter applied
    np.random.seed(3)
    b2, _ = minibatch_wsl.get_minibatch(roidb, raw=False)
    assert b2['data'].shape[1] == 3 and not np.array_equal(b2['data'].shape, ()) 


def test_rank_sharding_and_collate(cfgmod):
    from detectron.roi_data import loader_wsl as lw
    perms = [lw.epoch_permutation(101, 11, 3) for _ in range(2)]
    assert np.array_equal(perms[0], perms[1])                 # every rank derives the same order
    assert not np.array_equal(perms[0], lw.epoch_permutation(101, 11, 4))
    world = 4
    shards = [lw.rank_shard(perms[0], r, world) for r in range(world)]
    flat = sorted(int(i) for s in shards for g in s for i in g)
    assert flat == sorted(perms[0].tolist())                  # disjoint and complete
    assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
    a = dict(data=np.ones((1, 3, 4, 6), np.float32), rois=np.zeros((3, 5), np.float32),
             data_ids=np.zeros((1, 1), np.int32), obn_scores=np.ones((3, 1), np.float32),
             labels_int32=np.zeros((1,), np.int32), labels_oh=np.zeros((1, 20), np.float32))
    b = dict(a, data=np.ones((1, 3, 5, 5), np.float32), rois=np.zeros((2, 5), np.float32),
             obn_scores=np.ones((2, 1), np.float32))
    out = lw.collate([a, b])
    assert out['data'].shape == (2, 3, 5, 6) and out['data'][0, 0, 4, 0] == 0
    assert out['rois'][:, 0].tolist() == [0, 0, 0, 1, 1]
    mixed = lw.mixup_blobs(dict(a, data=np.stack([a['data'][0], 3 * a['data'][0]]),
                                labels_oh=np.eye(20, dtype=np.float32)[:2],
                                rois=np.array([[0, 1, 1, 2, 2], [1, 1, 1, 2, 2]], np.float32)), 0.25)
    assert np.allclose(mixed['data'], 0.25 + 0.75 * 3) and mixed['rois'][:, 0].tolist() == [0, 0]
    assert mixed['labels_oh'][0, :2].tolist() == [0.25, 0.75]


def test_loader_threads(cfgmod):
    c = cfgmod
    c.merge_cfg_from_file(YAML)
    c.merge_cfg_from_list(['WSL.USE_DISTORTION', False, 'WSL.USE_CROP', False,
                           'TRAIN.SCALES', '(64,)', 'TRAIN.MAX_SIZE', 96])
    from detectron.datasets import synthetic
    from detectron.roi_data.loader_wsl import RoIDataLoader
    roidb = synthetic.make_roidb(6, 12, 20, 64, 96, seed=2)
    ld = RoIDataLoader(roidb, num_loaders=2, minibatch_queue_size=4, rank=1, world_size=2,
                       ims_per_batch=2)
    ld.start()
    try:
        batch = ld.next_host_batch()
        assert batch['data'].shape[0] == 2 and set(batch['rois'][:, 0]) <= {0.0, 1.0}
        assert list(batch) and not ld.has_stopped()
    finally:
        ld.shutdown()


def test_host_image_prep_matches_oracle_and_raw_mode(cfgmod):
    """prep_im_for_blob (cv2.resize INTER_LINEAR restated) == the oracle's restatement; the
    device-prep ("raw") minibatch carries the same parameters the host path applies."""
    c = cfgmod
    c.merge_cfg_from_file(YAML)
    c.merge_cfg_from_list(['TRAIN.SCALES', '(96,)', 'TRAIN.MAX_SIZE', 140])
    from detectron.datasets import synthetic
    from detectron.roi_data import minibatch_wsl, loader_wsl
    from oracle import oracle
    # known answers of cv2.resize on a ramp (2x up: edge taps clamp; 0.5x down: cvRound(1.5) = 2)
    ramp = np.arange(12, dtype=np.float32).reshape(3, 4, 1)
    assert minibatch_wsl.resize_linear(ramp, 2.0)[0, :, 0].tolist() == [0, .25, .75, 1.25, 1.75, 2.25, 2.75, 3]
    assert minibatch_wsl.resize_linear(ramp, 0.5)[:, :, 0].tolist() == [[2.5, 4.5], [8.5, 10.5]]
    rng = np.random.default_rng(7)
    im = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    for target, cap in ((96, 140), (20, 1000), (37, 1000)):
        got, sc = minibatch_wsl.prep_im_for_blob(im, c.cfg.PIXEL_MEANS, target, cap)
        ref = oracle.prep_image(im, sc, means=c.cfg.PIXEL_MEANS.reshape(-1))
        assert got.shape == ref.shape and np.array_equal(got, ref)
    # HSV jitter: known cv2 conversions, and the host restatement == the oracle's
    px = np.array([[[0, 0, 255], [0, 255, 0], [255, 0, 0], [255, 255, 255], [128, 128, 128],
                    [10, 20, 30], [200, 100, 50]]], np.uint8)
    assert oracle.bgr2hsv_u8(px).reshape(-1, 3).tolist() == [
        [0, 255, 255], [60, 255, 255], [120, 255, 255], [0, 0, 255], [0, 0, 128], [15, 170, 30],
        [110, 191, 200]]
    assert np.array_equal(oracle.hsv2bgr_u8(oracle.bgr2hsv_u8(px)), px)
    for s0, s1 in ((1.0, 1.0), (1.37, 0.71), (1 / 1.5, 1.5)):
        assert np.array_equal(minibatch_wsl.distort_hsv(im, s0, s1), oracle.distort_hsv(im, s0, s1))
    c.cfg.WSL.USE_DISTORTION = True
    roidb = synthetic.make_roidb(2, 10, 20, 40, 64, seed=5)
    roidb[1]['flipped'] = True
    np.random.seed(3)
    host, _ = minibatch_wsl.get_minibatch(roidb[1:], raw=False)
    np.random.seed(3)
    raw, _ = minibatch_wsl.get_minibatch(roidb[1:], raw=True)
    assert np.array_equal(raw['rois'], host['rois']) and raw['data'].shape == (1, 3, 1, 1)
    r = raw['_raw'][0]
    assert r['flip'] and r['out_hw'] == host['data'].shape[2:]
    assert r['distort'] is not None
    ref = oracle.prep_image(r['im'], r['scale'], flip=True, crop=r['crop'],
                            means=c.cfg.PIXEL_MEANS.reshape(-1), distort=r['distort'])
    assert np.array_equal(ref.transpose(2, 0, 1), host['data'][0])
    # collate / mixup keep the raw images and the blend factor for the device side
    parts = [dict(raw), dict(loader_wsl.mixup_blobs(dict(
        raw, _raw=[r, r], data=np.zeros((2, 3, 1, 1), np.float32),
        labels_oh=np.eye(20, dtype=np.float32)[:2], labels_int32=np.zeros((2,), np.int32),
        data_ids=np.zeros((2, 1), np.int32)), 0.3))]
    out = loader_wsl.collate(parts)
    assert [len(g) for g in out['_raw']] == [1, 2] and out['_mix'] == [None, 0.3]


def test_graph_trace_fixture_shape():
    ops_ = GOLD['trace_train']['ops']
    assert len(ops_) == 103 and ops_[32][0] == 'RoIPoolF' and ops_[33][0] == 'RoIFeatureBoost'
    assert GOLD['trace_train']['losses'] == ['loss_cls', 'loss_cls_noise']
    assert len(GOLD['trace_test']['ops']) == 61


def test_graph_builders_reproduce_reference_trace(cfgmod):
    """The op lists the builders emit == the trace recorded from the reference builders
    (Conv / FC are recorded by the reference helper without their parameter blobs), in train and
    test mode, and with cfg.WSL.MIN_ENTROPY_LOSS (the f-4 adjacent loss)."""
    c = cfgmod
    c.merge_cfg_from_file(YAML)
    c.merge_cfg_from_list(['NUM_GPUS', 1])
    import detectron.modeling.model_builder_wsl as mb

    def norm(m):
        out = []
        for o in m.net.ops:
            ins = list(o.inputs)
            outs = list(o.outputs)
            if o.type in ('Conv', 'FC'):
                ins = ins[:1]
            if o.type == 'Dropout':          # the Caffe2 op's mask output is implicit in the helper call
                outs = outs[:1]
            out.append([o.type, ins, outs])
        return out
    for train, key in ((True, 'trace_train'), (False, 'trace_test')):
        m = mb.create(c.cfg.MODEL.TYPE, train=train)
        assert norm(m) == [[o[0], o[1], o[2]] for o in GOLD[key]['ops']]
    c.cfg.WSL.MIN_ENTROPY_LOSS = True
    m = mb.create(c.cfg.MODEL.TYPE, train=True)
    tail = GOLD['trace_train_min_entropy_tail']
    got = [[o.type, list(o.inputs), list(o.outputs), dict(o.args)] for o in m.net.ops[103:]]
    assert got == tail['ops'] and m.losses == tail['losses']
    assert m.grad_ops[0].type == 'MinEntropyLossGradient'      # its gradient joins rois_pred_grad
    assert 'rois_pred_grad' in m.grad_ops[0].outputs


def test_engine_host_scheduling_helpers():
    """Host-side arithmetic of the engine that needs no GPU: the fc6-wgrad column cut (whole
    waves of 256x256 tiles + a small-tile remainder), the all-reduce row chunks it composes with,
    and views of a split operand."""
    import torch
    from naws_hip import ops
    from naws_hip.engine import WsddnEngine
    from naws_hip.reducer import row_chunks

    class E(object):
        k6 = 25088
    cut = lambda rows, cus=256: WsddnEngine._wgrad_column_cut(E(), rows, cus)
    # 32 x 98 tiles = 12.25 waves -> 96 column tiles (12 waves) + 512 columns
    assert cut(8192) == 24576 and cut(4096) == 24576 and cut(2048) == 24576
    for rows in (8192, 4096, 2048):
        assert ((rows // 256) * (cut(rows) // 256)) % 256 == 0
    E.k6 = 8192                      # 32 x 32 tiles = 4 whole waves: no cut
    assert cut(8192) == 0
    E.k6 = 25088
    assert cut(8192, cus=7) == 0     # tail not a whole number of tile columns / more than half full
    for n in (1, 2, 4):
        ch = row_chunks(8192, n)
        assert ch[0][0] == 0 and ch[-1][1] == 8192 and all(a[1] == b[0] for a, b in zip(ch, ch[1:]))
        assert all((r1 - r0) % 128 == 0 for r0, r1 in ch)
    op = ops.F16x2(torch.zeros((2, 2, 4, 96, 16), dtype=torch.float16), torch.zeros((2, 2, 96)))
    sub = op.rows(32, 64)
    assert sub.planes.shape == (2, 2, 4, 32, 16) and sub.scales.shape == (2, 2, 32)
    assert sub.planes.data_ptr() == op.planes[..., 32:64, :].data_ptr()
    assert op.batches(1).planes.shape == (2, 1, 4, 96, 16) and op.batches(1).inv_scale.shape == (1, 96)
    assert op.inv_scale.data_ptr() == op.scales[1].data_ptr()


def test_checkpoint_preserved_blobs_round_trip(tmp_path, cfgmod):
    """Blobs of the weights file that the model does not use are carried through load -> save
    under their UNSCOPED names (reference net_wsl.py:129-137 keeps them as '__preserve__/<name>'
    in the workspace and :170-178 writes UnscopeName() of that), stable over resume cycles."""
    import torch
    from detectron.utils import net_wsl

    class Model(object):
        params = ['fc6_w']
        param_shapes = {'fc6_w': (2, 3)}

        def TrainableParams(self):
            return ['fc6_w']

    class Executor(object):
        def __init__(self):
            self.b = {'fc6_w': torch.zeros(2, 3), 'fc6_w_momentum': torch.zeros(2, 3)}

        def blobs(self, with_momentum=True):
            return {k: v for k, v in self.b.items()
                    if with_momentum or not k.endswith('_momentum')}

        def load_blobs(self, blobs):
            self.b.update(blobs)

        def broadcast_parameters(self):
            pass

    src = str(tmp_path / 'vgg.pkl')
    w6 = np.arange(6, dtype=np.float32).reshape(2, 3)
    w1000 = np.arange(8, dtype=np.float32).reshape(4, 2)
    net_wsl.save_object({'blobs': {'fc6_w': w6, 'fc1000_w': w1000}}, src)
    path = src
    for cycle in range(2):
        model, ex = Model(), Executor()
        net_wsl.initialize_from_weights_file(model, path, ex)
        assert list(model.preserved_blobs) == ['__preserve__/fc1000_w']
        path = str(tmp_path / ('model_iter%d.pkl' % cycle))
        net_wsl.save_model_to_weights_file(path, model, ex)
        saved = net_wsl.load_object(path)['blobs']
        assert sorted(saved) == ['fc1000_w', 'fc6_w', 'fc6_w_momentum'], sorted(saved)
        assert np.array_equal(saved['fc1000_w'], w1000) and np.array_equal(saved['fc6_w'], w6)


HOST = np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_host_paths.npz'))


def test_dedup_hash_and_scatter_back_match_reference(cfgmod):
    """Pairs captured from the reference's own im_detect_bbox (core/test_wsl.py:102-178) run with a
    recording workspace: the rois it feeds after projection (float64 product) + dedup hash, and the
    scores it returns after the scatter-back."""
    from detectron.core import test_wsl
    boxes, obn = HOST['dedup_boxes'], HOST['dedup_obn']
    rois = test_wsl.project_rois(boxes, float(HOST['dedup_im_scale']))
    uniq, index, inv = test_wsl.dedup_rois(rois, float(HOST['dedup_factor']))
    assert uniq.dtype == np.float32 and np.array_equal(uniq, HOST['dedup_fed_rois'])
    assert np.array_equal((obn + 1.0).astype(np.float32)[index], HOST['dedup_fed_obn'])
    assert uniq.shape[0] < rois.shape[0]
    # the fixture's "network": scores are a function of the fed rows only
    k = HOST['dedup_scores'].shape[1]
    base = (uniq[:, 1:5].sum(1, keepdims=True) * 0.001 + HOST['dedup_fed_obn']).astype(np.float32)
    scores = (base + np.arange(k, dtype=np.float32)[None, :] * 0.01).astype(np.float32)
    assert np.array_equal(scores[inv], HOST['dedup_scores'])
    assert np.array_equal(np.tile(boxes, (1, k)), HOST['dedup_pred_boxes'])


def test_image_id_blob_matches_reference():
    from detectron.roi_data import minibatch_wsl
    blob = minibatch_wsl._get_image_id_blob([{'image': str(s)} for s in HOST['imgid_names']])
    assert blob.dtype == np.int32 and np.array_equal(blob, HOST['imgid_blob'])


def test_bagging_mixup_blend_matches_reference(cfgmod):
    """The blended minibatch RoIDataLoader.get_next_minibatch produced in the reference
    (loader_wsl.py:130-168) from a captured two-image minibatch and its seeded lambda."""
    from detectron.roi_data import loader_wsl
    two = {k[len('mixup_in_'):]: HOST[k] for k in HOST.files if k.startswith('mixup_in_')}
    want = {k[len('mixup_out_'):]: HOST[k] for k in HOST.files if k.startswith('mixup_out_')}
    got = loader_wsl.mixup_blobs(two, float(HOST['mixup_lam']))
    assert set(want) <= set(got)
    for k, v in want.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape, k
        assert np.array_equal(got[k], v), k         # float32 products, float32 sum: bit for bit
    # and the stream: the same numpy draws give the same branch decision and lambda
    np.random.seed(int(HOST['mixup_seed']))
    assert np.random.random() > 0.8
    assert np.random.beta(float(HOST['mixup_alpha']), float(HOST['mixup_alpha'])) == float(HOST['mixup_lam'])


OICR = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'reference_oicr.json')))


def _norm_ops(ops_):
    out = []
    for o in ops_:
        ins, outs, args = list(o.inputs), list(o.outputs), dict(o.args)
        if o.type in ('Conv', 'FC'):
            ins = ins[:1]
        if o.type == 'Dropout':
            outs = outs[:1]
        if 'uuid' in args:
            args['uuid'] = 0             # random per build upstream (uuid4)
        out.append([o.type, ins, outs, args])
    return out


def test_oicr_builders_reproduce_reference_trace(cfgmod):
    """WSL.OICR on the plain (non-webly) generalized_wsl model - the op lists of
    wsl_heads.add_wsl_outputs / add_wsl_oicr_outputs / add_wsl_losses / add_oicr_losses == the
    trace recorded from the reference builders (tests/golden/make_golden_oicr.py), train and test
    mode, incl. the RoILabel / SoftmaxWithLossN arguments; and the backward plan reaches the three
    refinement classifiers."""
    c = cfgmod
    c.merge_cfg_from_file(YAML)
    c.merge_cfg_from_list(['NUM_GPUS', 4, 'WEBLY.WEBLY_ON', False, 'WSL.OICR', True,
                           'FAST_RCNN.ROI_BOX_HEAD', 'wsl_heads.add_VGG16_roi_2fc_head'])
    c.assert_and_infer_cfg(make_immutable=False)       # WSL.OICR is an accepted switch now
    import detectron.modeling.model_builder_wsl as mb
    for train, key in ((True, 'wsl_oicr_train'), (False, 'wsl_oicr_test')):
        m = mb.create(c.cfg.MODEL.TYPE, train=train)
        want = OICR[key]
        got = _norm_ops(m.net.ops)
        assert [g[:3] for g in got] == [w[:3] for w in want['ops']]
        for g, w in zip(got, want['ops']):
            if g[0] in ('RoILabel', 'SoftmaxWithLossN', 'Mean', 'Split', 'Concat'):
                assert g[3] == w[3], g
        assert m.losses == want['losses'] and m.metrics == want['metrics']
    m = mb.create(c.cfg.MODEL.TYPE, train=True)
    for k in (1, 2, 3):
        assert m.param_shapes['cls_score%d_w' % k] == (21, 4096)
        assert m.param_to_grad['cls_score%d_w' % k] == 'cls_score%d_w_grad' % k
    gtypes = [o.type for o in m.grad_ops]
    assert gtypes.count('SoftmaxWithLossNGradient') == 3
    # drop7 feeds fc8c, fc8d and the three cls_score FCs: its gradient accumulates five times
    acc = [o for o in m.grad_ops if o.type == 'FCGradient' and o.inputs[0] == 'drop7']
    assert len(acc) == 5 and sum(o.args['_accumulate'][0] for o in acc) == 4
    # WEBLY.MINING is dead code upstream (add_webly_mining does not exist): same failure here
    c.merge_cfg_from_list(['WEBLY.WEBLY_ON', True, 'WEBLY.MINING', True, 'WSL.OICR', False,
                           'FAST_RCNN.ROI_BOX_HEAD', 'webly_heads.add_VGG16_roi_2fc_noise_head'])
    with pytest.raises(AttributeError):
        mb.create(c.cfg.MODEL.TYPE, train=True)


def test_entropy_weight_builder_reproduces_reference_trace(cfgmod):
    c = cfgmod
    c.merge_cfg_from_file(YAML)
    c.merge_cfg_from_list(['NUM_GPUS', 4])
    from detectron.modeling.detector import DetectionModelHelper
    from detectron.modeling import webly_heads
    m = DetectionModelHelper(name='t', train=True, num_classes=c.cfg.MODEL.NUM_CLASSES)
    w = webly_heads.add_entropy_weight(m, 'rois_pred', 'rois')
    assert w == OICR['entropy_weight']['weight']
    assert _norm_ops(m.net.ops) == OICR['entropy_weight']['ops']


def _ckpt_fakes():
    import torch

    class Model(object):
        params = ['fc6_w', 'fc6_b', '_[noisy]_fc6_w', '_[noisy]_fc6_b', 'fc8c_w', '_[noisy]_fc7_w']
        param_shapes = {'fc6_w': (4, 6), 'fc6_b': (4,), '_[noisy]_fc6_w': (4, 6),
                        '_[noisy]_fc6_b': (4,), 'fc8c_w': (3, 4), '_[noisy]_fc7_w': (2, 2)}

        def TrainableParams(self):
            return list(self.params)

    class Executor(object):
        def __init__(self):
            self.b = {}
            for n, s in Model.param_shapes.items():
                self.b[n] = torch.full(s, 7.0)
                self.b[n + '_momentum'] = torch.zeros(s)
            self.loaded = None

        def blobs(self, with_momentum=True):
            return {k: v for k, v in self.b.items() if with_momentum or not k.endswith('_momentum')}

        def load_blobs(self, blobs):
            self.loaded = dict(blobs)
            self.b.update(blobs)

        def broadcast_parameters(self):
            pass
    return Model, Executor


def test_weights_file_load_matches_what_the_reference_feeds(tmp_path, cfgmod):
    """`initialize_gpu_from_weights_file` of the imported reference (net_wsl.py:51-137), run on a
    pretrained-style file with a recording workspace (tests/golden/make_golden_checkpoint.py): the
    same blobs arrive here - the '_[noisy]_foo' <- 'foo' alias incl. its momentum, float64
    sources as float32, momentum only where the file has it, and the '__preserve__/' set (which
    keeps a blob that only served an alias, and drops None entries and stray momenta)."""
    from detectron.utils import net_wsl
    ref = np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_checkpoint_load.npz'))
    src = {k[4:]: ref[k] for k in ref.files if k.startswith('src/')}
    src['unused_none'] = None
    path = str(tmp_path / 'pretrained.pkl')
    net_wsl.save_object({'blobs': src}, path)
    Model, Executor = _ckpt_fakes()
    model, ex = Model(), Executor()
    net_wsl.initialize_from_weights_file(model, path, ex)
    fed = {k[4:]: ref[k] for k in ref.files if k.startswith('fed/')}
    want_params = {k[len('gpu_0/'):]: v for k, v in fed.items() if k.startswith('gpu_0/')}
    # parameters the file does not hold keep their initialisation here and are simply not fed there
    got = {k: v.numpy() for k, v in ex.loaded.items()
           if k in want_params or not (v == 7.0).all()}
    assert sorted(got) == sorted(want_params)
    for k, v in want_params.items():
        assert got[k].dtype == np.float32 == v.dtype and np.array_equal(got[k], v), k
    want_keep = {k: v for k, v in fed.items() if k.startswith('__preserve__/')}
    assert sorted(model.preserved_blobs) == sorted(want_keep)
    for k, v in want_keep.items():
        assert np.array_equal(model.preserved_blobs[k], v)


def test_weights_file_written_by_the_reference_loads_and_resaves(tmp_path, cfgmod):
    """tests/golden/reference_checkpoint.pkl is the pickle the reference's own
    `save_model_to_weights_file` (net_wsl.py:140-180) wrote for a small state: it loads here, and
    saving that state again gives the same blob names and arrays and a cfg entry the reference's
    loader would accept (a yaml mapping with the same top-level keys)."""
    import yaml
    from detectron.utils import net_wsl
    path = os.path.join(ROOT, 'tests', 'golden', 'reference_checkpoint.pkl')
    theirs = net_wsl.load_object(path)
    assert sorted(theirs) == ['blobs', 'cfg']
    Model, Executor = _ckpt_fakes()
    model, ex = Model(), Executor()
    net_wsl.initialize_from_weights_file(model, path, ex)
    for n in Model.params:
        assert np.array_equal(ex.b[n].numpy(), theirs['blobs'][n]), n
        assert np.array_equal(ex.b[n + '_momentum'].numpy(), theirs['blobs'][n + '_momentum']), n
    assert list(model.preserved_blobs) == ['__preserve__/fc1000_w']
    out = str(tmp_path / 'again.pkl')
    net_wsl.save_model_to_weights_file(out, model, ex)
    ours = net_wsl.load_object(out)
    assert sorted(ours['blobs']) == sorted(theirs['blobs'])
    for k, v in theirs['blobs'].items():
        assert ours['blobs'][k].dtype == v.dtype and np.array_equal(ours['blobs'][k], v), k
    # the reference dumps its AttrDict tree with python-object tags (AttrDict, numpy scalars): only
    # its own process can rebuild all of that, so compare top-level section names.  Ours carries the
    # SAME AttrDict tag on every mapping (the reference reads the string back with its unsafe
    # loader and uses the result as an AttrDict: net_wsl.py:64-66, :277; checked against the
    # imported reference in tests/test_test_engine_cpu.py) and plain scalars / lists below them
    import re
    from detectron.core.config import load_cfg
    assert ours['cfg'].startswith('!!python/object/new:detectron.utils.collections.AttrDict')
    a = load_cfg(ours['cfg'])
    b = set(re.findall(r'^  ([A-Z][A-Z0-9_]*):', theirs['cfg'], re.M))
    # (the mirror carries the sections of the hot path only, plus its own NAWS block)
    assert b and set(a) - {'NAWS'} <= b, sorted(set(a) - b)
    assert 'NUM_CLASSES' in a['MODEL'] and 'BBOX_REG_WEIGHTS' in a['MODEL']   # what the reference's loader reads


def test_training_stats_reproduce_the_reference_log_lines(cfgmod, monkeypatch):
    """tests/golden/reference_training_stats.json: the reference's own TrainingStats +
    log_json_stats driven for 400 iterations (NUM_GPUS = 8) on a seeded series with a
    deterministic clock.  The mirror, fed the same per-iteration values - averaged the way
    net_wsl.average_multi_gpu_blob averages them -, prints the same `json_stats:` lines character
    for character at the same iterations (window averages, rounded window-averaged queue size,
    global-average timer with its one reset, '%.6f' strings)."""
    import json
    import os
    import numpy as np
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden',
                                       'reference_training_stats.json')))
    c = cfgmod
    c.merge_cfg_from_list(['NUM_GPUS', gold['NUM_GPUS'], 'SOLVER.MAX_ITER', gold['MAX_ITER']])
    import detectron.utils.training_stats_wsl as ts

    class _Model(object):
        losses, metrics = gold['losses'], gold['metrics']

    clock = [1000.0]
    monkeypatch.setattr(ts.time, 'time', lambda: clock[0])
    lines = []
    stats = ts.TrainingStats(_Model(), printer=lambda s: lines.append([it, s]))
    assert stats.LOG_PERIOD == gold['LOG_PERIOD'] and stats.WIN_SZ == gold['WIN_SZ']
    for it in range(len(gold['lr'])):
        stats.IterTic()
        clock[0] += gold['dt'][it]
        stats.IterToc()
        vals = {}
        for k, series in gold['values'].items():
            tot = 0
            for v in series[it]:                       # sum_multi_gpu_blob: val += float(blob)
                tot += float(np.float32(v))
            vals[k] = tot / gold['NUM_GPUS']
        stats.UpdateIterStats(vals, gold['qsize'][it])
        stats.LogIterStats(it, np.float32(gold['lr'][it]), gold['mem_bytes'])
        if it == stats.LOG_PERIOD:
            stats.ResetIterTimer()
    assert lines == gold['lines']
    assert len(lines) == 4 and lines[1][0] == 160


def _describe_cfg_value(v):
    import numpy as np
    if isinstance(v, np.ndarray):
        return {'type': 'ndarray', 'dtype': str(v.dtype), 'value': v.tolist()}
    if isinstance(v, tuple):
        return {'type': 'tuple', 'value': [_describe_cfg_value(x) for x in v]}
    if isinstance(v, list):
        return {'type': 'list', 'value': [_describe_cfg_value(x) for x in v]}
    if isinstance(v, dict):
        return {'type': 'dict', 'keys': sorted(v.keys())}
    return {'type': type(v).__name__, 'value': v}


def test_cfg_merge_behaviour_matches_the_reference(cfgmod):
    """tests/golden/reference_cfg_behaviour.json: 45 `merge_cfg_from_list` and 15
    `merge_cfg_from_cfg` cases run on the imported reference (value decoding, the type check with
    its three conversions - str(anything), tuple <-> list, ndarray -, deprecated keys ignored,
    renamed keys refused with their pointer, unknown keys, immutability).  The mirror ends with the
    same value and type, or raises the same exception class with the same message."""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden',
                                       'reference_cfg_behaviour.json')))
    c = cfgmod

    def lookup(full_key):
        node = c.cfg
        for p in full_key.split('.'):
            node = node[p]
        return node

    def first_leaf(d, stack=()):
        for k, v in d.items():
            if isinstance(v, dict):
                return first_leaf(v, stack + (k,))
            return '.'.join(stack + (k,))

    for group, fn in (('list', c.merge_cfg_from_list), ('dict', c.merge_cfg_from_cfg)):
        for rec in gold[group]:
            c.reset_cfg()
            args = rec['args']
            key = args[0] if group == 'list' else first_leaf(args)
            try:
                fn(list(args) if group == 'list' else args)
                try:
                    got = _describe_cfg_value(lookup(key))
                except KeyError:
                    got = {'type': 'absent'}
                assert 'result' in rec, (args, 'the reference raised', rec.get('error'))
                assert got == rec['result'], (args, got, rec['result'])
            except AssertionError as e:
                if 'error' in rec and rec['error'][0] == 'AssertionError':
                    assert str(e) == rec['error'][1], (args, str(e))
                else:
                    raise
            except (KeyError, ValueError) as e:
                assert 'error' in rec, (args, 'the reference accepted it', rec.get('result'), repr(e))
                assert type(e).__name__ == rec['error'][0], (args, repr(e), rec['error'])
                if 'AttrDict' not in rec['error'][1]:        # (a message that spells the class name)
                    assert str(e) == rec['error'][1], (args, str(e), rec['error'][1])
    c.reset_cfg()
    c.assert_and_infer_cfg()
    for stmt, err in gold['immutable']:
        try:
            exec(stmt, {'cfg': c.cfg})
            raised = None
        except BaseException as e:          # noqa: B902
            raised = type(e).__name__
        assert raised == err, (stmt, raised, err)
    c.cfg.immutable(False)


# ---- a whole minibatch against the imported reference's get_minibatch ----------------------------

def _minibatch_roidb():
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        'minibatch_inputs', os.path.join(os.path.dirname(__file__), 'golden', 'minibatch_inputs.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize('tag,over,raw', [
    ('default', [], True),
    ('crop_no_distortion', ['WSL.USE_CROP', True, 'WSL.USE_DISTORTION', False], True),
    ('single_scale', ['TRAIN.SCALES', (600,), 'TRAIN.MAX_SIZE', 1000, 'WSL.USE_CROP', False, 'WSL.USE_DISTORTION', False], False),
    ('single_scale', ['TRAIN.SCALES', (600,), 'TRAIN.MAX_SIZE', 1000, 'WSL.USE_CROP', False, 'WSL.USE_DISTORTION', False], True),
])
def test_whole_minibatch_matches_the_reference_capture(tag, over, raw, monkeypatch, cfgmod):
    """`get_minibatch` on a seeded three-image roidb against the imported reference's call
    (tests/golden/make_golden_minibatch.py): random scale / jitter / crop draws in the same order
    and number (the RNG's next draw afterwards is the same), the same per-image scale and resized
    size, proposals projected into the same cropped / flipped / scaled frame, same labels."""
    C = cfgmod
    from detectron.roi_data import minibatch_wsl as mbw
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'reference_minibatch.npz'))
    gen = _minibatch_roidb()
    C.merge_cfg_from_file(YAML)
    C.merge_cfg_from_list(['NUM_GPUS', 1, 'TRAIN.BATCH_SIZE_PER_IM', 30] + list(over))
    cfg = C.cfg
    assert [str(cfg.TRAIN.SCALES), str(cfg.TRAIN.MAX_SIZE), str(cfg.WSL.USE_CROP), str(cfg.WSL.USE_DISTORTION),
            str(cfg.WSL.CROP), str(cfg.WSL.SATURATION), str(cfg.WSL.EXPOSURE)] == list(g[tag + '__cfg'])
    monkeypatch.setattr(mbw, '_read_image', lambda entry: gen.fake_image(entry['image']))
    np.random.seed(77)
    blobs, valid = mbw.get_minibatch(gen.make_roidb(), raw=raw)
    nxt = np.random.random()
    assert bool(valid) == bool(g[tag + '__valid'])
    assert nxt == float(g[tag + '__next_draw'])
    for k in ('data_ids', 'rois', 'obn_scores', 'labels_int32', 'labels_oh'):
        want = g[tag + '__' + k]
        got = np.asarray(blobs[k])
        assert got.shape == want.shape and got.dtype == want.dtype, (k, got.shape, want.shape, got.dtype, want.dtype)
        assert np.array_equal(got, want), k
    to = g[tag + '__resized_to']
    if raw:
        assert [list(r['out_hw']) for r in blobs['_raw']] == to.tolist()
        assert np.array_equal(np.array([r['scale'] for r in blobs['_raw']]), g[tag + '__im_scales'])
        frm = g[tag + '__resized_from']
        assert [[r['crop'][2] - r['crop'][0] + 1, r['crop'][3] - r['crop'][1] + 1]
                for r in blobs['_raw']] == frm.tolist()
    else:
        assert list(blobs['data'].shape) == g[tag + '__data_shape'].tolist()
