"""ORACLE parity on the shapes the loader really emits, not only at the
bench's 600 x 1000 / R = 2000:

  * the ends of TRAIN.SCALES / TEST.BBOX_AUG.SCALES: 480 x 800 and 1200 x 2000 single images
    (reference: roi_data/minibatch_wsl.py:111-171 - a random scale per image, the long side capped);
  * an odd crop, 563 x 877 (WSL.USE_CROP: minibatch_wsl.py:129-150) - no dimension a multiple of
    any tile, pool or K-slab size;
  * a mixup batch (roi_data/loader_wsl.py:136-168): two images blended into one, their proposals
    concatenated on batch index 0 to R = 4096, fractional labels_oh.

Every case runs the oracle (oracle/: torch-CPU conv / fc + the C restatement of the custom ops)
ONCE and compares each fp32 arithmetic plan of the HIP path against it - never one HIP plan
against another:

  conv5_3                       <= 1e-4 of max|conv5_3|
  RoI bins / argmax / values    bit-exact on identical input (the oracle's conv5_3)
  fc8 logits                    <= 1e-4 of max|logit|
  loss_cls, loss_cls_noise      <= 1e-4 relative
  class weights, cls_prob       rtol 1e-4 / the softmax bound of the full-size test
  d_logits                      normwise <= 1e-3 against the fp32 oracle (the float64 arbiter of
                                the 600 x 1000 test says both sit ~1e-4 from the truth)
"""
import numpy as np
import pytest
import torch

from test_gpu_fullsize_oracle import LOGIT_KEYS, _dense_roi_feat, _engine, _masks, _relmax

pytestmark = pytest.mark.gpu
SEED = 11
C = 20


def _batch(name):
    from detectron.datasets import synthetic
    from detectron.roi_data import loader_wsl
    if name == 'mixup':
        roidb = synthetic.make_roidb(2, 2048, C, 600, 1000, seed=SEED + 3)
        roidb[1]['gt_classes'][roidb[1]['gt_classes'] > 0] = 1 + (int(roidb[0]['gt_classes'].max()) % C)
        mb2 = synthetic.make_minibatch(roidb, C, max_rois=2048)
        mb = loader_wsl.mixup_blobs(mb2, 0.3)
        assert mb['data'].shape[0] == 1 and mb['rois'].shape[0] == 4096
        assert (mb['rois'][:, 0] == 0).all() and sorted(set(mb['labels_oh'][0]) - {0.0}) == \
            pytest.approx([0.3, 0.7])
        return mb
    h, w, r = {'scale480': (480, 800, 1500), 'scale1200': (1200, 2000, 600),
               'crop563x877': (563, 877, 777)}[name]
    mb = synthetic.make_minibatch(synthetic.make_roidb(1, r, C, h, w, seed=SEED + 1), C, max_rois=4096)
    assert mb['rois'].shape[0] == r and mb['data'].shape[2:] == (h, w)
    return mb


@pytest.fixture(scope='module')
def blobs():
    from detectron.datasets import synthetic
    return synthetic.init_blobs(C, seed=SEED)


@pytest.mark.parametrize('name', ['scale480', 'scale1200', 'crop563x877', 'mixup'])
def test_loader_shapes_match_the_oracle(dev, blobs, name):
    from naws_hip import ops
    from oracle import oracle
    mb = _batch(name)
    rt = mb['rois'].shape[0]
    eng = _engine(dev, C, blobs, 'fp32')
    masks = _masks(eng, rt, dev)           # the counter-based masks every plan draws at step 0
    del eng
    ref = oracle.full_forward_backward(blobs, mb, masks, C)
    t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
    # ---- RoI index assignment on identical input: bins, argmax, values bit-exact
    x = torch.from_numpy(ref['conv5_3']).to(dev)
    boost = t['obn_scores'].reshape(-1)
    y, am = ops.roi_pool_f(x, t['rois'], 7, 7, 0.125, boost=boost, layout='NCHW', with_argmax=True)
    assert np.array_equal(am.cpu().numpy(), ref['roi_argmax'])
    assert np.array_equal(y.cpu().numpy().reshape(rt, -1), ref['roi_feat'].reshape(rt, -1))
    y2 = ops.roi_pool_f(x.permute(0, 2, 3, 1).contiguous(), t['rois'], 7, 7, 0.125, boost=boost,
                        layout='NHWC', hier=True)
    assert torch.equal(y2.reshape(rt, -1), y.reshape(rt, -1))
    del x, y, y2, am
    act, tl = ref['act'], ref['tails'][0]
    logits = np.concatenate([act[k] for k in LOGIT_KEYS], 1)
    lmax = float(np.abs(logits).max())
    dl32 = np.concatenate([ref['d_logits'][k] for k in LOGIT_KEYS], 1).astype(np.float64)
    for mode in ('fp16x2', 'fp32x3', 'fp32'):
        eng = _engine(dev, C, blobs, mode)
        conv5 = eng.conv_body(t['data'])
        m = {'conv5_3': _relmax(conv5.permute(0, 3, 1, 2), ref['conv5_3'])}
        xf = eng._roi_features(conv5, t['rois'], t['obn_scores'])
        m['roi_feat'] = _relmax(_dense_roi_feat(xf), ref['roi_feat'].reshape(rt, -1))
        del conv5, xf
        out = eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'])
        m['logits'] = _relmax(out['logits'], logits)
        for k in ('loss_cls', 'loss_cls_noise'):
            a, b = float(out[k][0]), float(tl[k])
            m[k] = abs(a - b) / abs(b)
        dlg = out['d_logits'].cpu().numpy().astype(np.float64)
        m['d_logits'] = float(np.linalg.norm(dlg - dl32) / np.linalg.norm(dl32))
        print('\n[%s, %s] vs oracle: %s; max|logit| %.1f' % (
            name, mode, ', '.join('%s %.1e' % kv for kv in m.items()), lmax))
        for k in ('conv5_3', 'roi_feat', 'logits', 'loss_cls', 'loss_cls_noise'):
            assert m[k] <= 1e-4, (name, mode, k, m[k])
        assert m['d_logits'] <= 1e-3, (name, mode, m['d_logits'])
        ptol = 2e-4 * max(1.0, lmax)
        for k in ('cls_prob', 'cls_prob_noise'):
            got, want = out[k][0].cpu().numpy(), tl[k][0]
            assert float(np.abs(got / want - 1).max()) <= ptol, (name, mode, k)
        np.testing.assert_allclose(out['class_weight'][0].cpu().numpy(), tl['class_weight'][0],
                                   rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(out['class_weight_noise'][0].cpu().numpy(),
                                   tl['class_weight_noise'][0], rtol=1e-4, atol=1e-6)
        # the 16 parameter gradients: direction and size against the fp32 oracle
        for gname, g32 in ref['grads'].items():
            if gname.endswith('fc8d_b'):
                continue                   # identically zero in exact arithmetic (rounding residues)
            got = eng.grad_blob(gname).cpu().numpy().astype(np.float64)
            g32 = g32.astype(np.float64)
            e = np.linalg.norm(got - g32) / np.linalg.norm(g32)
            assert e <= 2e-3, (name, mode, gname, e)
        del eng, out
        torch.cuda.empty_cache()
