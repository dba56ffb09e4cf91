"""Full-size ORACLE parity on the headline configuration (BASELINE.json configs[1]: a 600x1000
image with 2000 proposals, C = 20, training mode with the dropout masks replayed) for all three
fp32 arithmetic plans, and configs[3]'s arithmetic (C = 80, bf16 MFMA operands, fp32 loss) at its
own tolerance.

The oracle (oracle/: torch-CPU conv / fc + the C restatement of the custom operators) runs the
whole image once per module - conv body 464 GFLOP, RoIPoolF over 2000 x 512 x 49 bins, head
forward + backward 2 TFLOP - and every plan is compared against the same reference:

  conv5_3                      <= 1e-4 of max|conv5_3|            (north_star: fp32, 1e-4 rel)
  RoI bins / argmax / values   bit-exact on identical input       (north_star: bit-exact)
  roi_feat, drop6, drop7       <= 1e-4 of the blob's max
  fc8 logits                   <= 1e-4 of max|logit|
  class weights                rtol 1e-4
  loss_cls, loss_cls_noise     <= 1e-4 relative
  cls_prob                     relative error <= 2e-4 x max|logit| (what a logit error does to
                               a softmax output), measured 1e-4..2e-4 at max|logit| = 58
  d_logits, 16 gradients       against the float64 arbiter (oracle.head_float64), normwise:
                               d_logits <= 1e-4, parameter gradients <= 5e-4 (the fp32 oracle's
                               own distance from the arbiter is printed beside each)
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H, W, R, SEED = 600, 1000, 2000, 11
LOGIT_KEYS = ('fc8c', 'fc8d', 'noisy_fc8c', 'noisy_fc8d')


def _masks(eng, rt, dev):
    from naws_hip import ops
    m6 = ops.dropout_mask(eng._seed(6), eng.dropout, rt * 8192, dev).view(rt, 8192).cpu().numpy()
    m7 = ops.dropout_mask(eng._seed(7), eng.dropout, 2 * rt * 4096, dev).view(2, rt, 4096).cpu().numpy()
    return {'drop6': m6[:, :4096], '_[noisy]_drop6': m6[:, 4096:], 'drop7': m7[0],
            '_[noisy]_drop7': m7[1]}


def _engine(dev, c, blobs, mode):
    from naws_hip.engine import WsddnEngine
    eng = WsddnEngine(c + 1, dev, dropout=0.5, gpu_num=1, seed=SEED, mfma_dtype=mode)
    eng.set_conv_blobs(blobs)
    eng.set_head_blobs(blobs)
    return eng


def _reference(dev, stats):
    """Oracle forward + backward of one full-size image, C = 20, plus the float64 arbiter.
    stats = 'kaiming': seeded Kaiming-normal weights, zero biases, uniform-random pixels (the
    bench workload); 'skewed': synthetic.skew_blobs / skew_images on top of that - per-channel
    weight scales log-uniform over 2^+-6 in every conv and fc layer, non-zero biases, the left
    third of the image at 2^-12 of the rest plus a smooth low-frequency component (VERDICT r3
    weak #2: the split plans' error depends on the operand statistics)."""
    from detectron.datasets import synthetic
    from oracle import oracle
    torch.set_num_threads(max(1, torch.get_num_threads()))
    c = 20
    blobs = synthetic.init_blobs(c, seed=SEED)
    mb = synthetic.make_minibatch(synthetic.make_roidb(1, R, c, H, W, seed=SEED), c)
    assert mb['rois'].shape[0] == R
    if stats == 'skewed':
        blobs = synthetic.skew_blobs(blobs, seed=SEED)
        mb['data'] = synthetic.skew_images(mb['data'])
    eng = _engine(dev, c, blobs, 'fp32')
    masks = _masks(eng, R, dev)           # the counter-based masks every plan draws at step 0
    del eng
    ref = oracle.full_forward_backward(blobs, mb, masks, c)
    cw = [(t['class_weight'], t['class_weight_noise']) for t in ref['tails']]
    arb = oracle.head_float64(ref['roi_feat'], mb['rois'], mb['labels_oh'], blobs, masks, cw)
    return dict(c=c, blobs=blobs, mb=mb, masks=masks, ref=ref, arb=arb)


@pytest.fixture(scope='module')
def ref20(dev):
    return _reference(dev, 'kaiming')


@pytest.fixture(scope='module')
def ref20_skewed(dev):
    return _reference(dev, 'skewed')


def _relmax(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return float(np.abs(a.astype(np.float64) - b).max() / (np.abs(b).max() + 1e-30))


def _dense_roi_feat(x):
    """The engine's fc6 input as a dense fp32 [R, K] matrix (fp16x2 plan: hi + lo planes times
    the per-roi inverse scale)."""
    from naws_hip import ops
    if isinstance(x, ops.F16x2):
        p = x.planes.float()                                   # [2, K/16, R, 16]
        d = (p[0] + p[1]).permute(1, 0, 2).reshape(p.shape[2], -1)
        return d * x.inv_scale[:, None]
    return x


def test_roi_pool_bitexact_at_full_size(dev, ref20):
    """RoIPoolF + boost on the ORACLE's conv5_3 (identical input): bins, argmax and values
    bit-exact over all 2000 x 512 x 49 outputs, NCHW and NHWC layouts."""
    from naws_hip import ops
    ref, mb = ref20['ref'], ref20['mb']
    x = torch.from_numpy(ref['conv5_3']).to(dev)
    rois = torch.from_numpy(mb['rois']).to(dev)
    boost = torch.from_numpy(mb['obn_scores'].reshape(-1)).to(dev)
    y, am = ops.roi_pool_f(x, rois, 7, 7, 0.125, boost=boost, layout='NCHW', with_argmax=True)
    assert np.array_equal(am.cpu().numpy(), ref['roi_argmax'])
    assert np.array_equal(y.cpu().numpy().reshape(R, -1), ref['roi_feat'].reshape(R, -1))
    y2 = ops.roi_pool_f(x.permute(0, 2, 3, 1).contiguous(), rois, 7, 7, 0.125, boost=boost,
                        layout='NHWC')
    assert torch.equal(y2, y)


def _region_relmax(got_nchw, want_nchw, channel_rms):
    """Relative measures that a tensor-wide maximum hides: (a) per channel, max error / the RMS
    of the channel's pre-activation (channels 12 octaves below the loudest one must still be
    right), (b) max error / max value over the columns that see only the dark third of the image."""
    g = got_nchw.detach().cpu().numpy().astype(np.float64)
    w = want_nchw.astype(np.float64)
    err = np.abs(g - w)
    live = channel_rms > 0
    per_channel = float((err.max(axis=(0, 2, 3))[live] / channel_rms[live]).max())
    wd = w.shape[3] // 3 - 4                                   # conv5_3 columns inside the dark third
    dark = float(err[..., :wd].max() / max(np.abs(w[..., :wd]).max(), 1e-300))
    return per_channel, dark


@pytest.mark.parametrize('mode', ['fp16x2', 'fp32x3', 'fp32'])
@pytest.mark.parametrize('stats', ['kaiming', 'skewed'])
def test_full_size_training_step_matches_oracle(dev, request, stats, mode):
    r20 = request.getfixturevalue('ref20' if stats == 'kaiming' else 'ref20_skewed')
    c, blobs, mb, ref, arb = (r20[k] for k in ('c', 'blobs', 'mb', 'ref', 'arb'))
    mode_tag = '%s, %s' % (mode, stats)
    t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
    eng = _engine(dev, c, blobs, mode)
    # ---- stage by stage (the same kernels forward_backward launches)
    conv5 = eng.conv_body(t['data'])
    m = {'conv5_3': _relmax(conv5.permute(0, 3, 1, 2), ref['conv5_3'])}
    m['conv5_3/channel'], m['conv5_3/dark third'] = _region_relmax(
        conv5.permute(0, 3, 1, 2), ref['conv5_3'], ref['conv_stats']['conv5_3_rms'])
    x = eng._roi_features(conv5, t['rois'], t['obn_scores'])
    m['roi_feat'] = _relmax(_dense_roi_feat(x), ref['roi_feat'].reshape(R, -1))
    h6, h7, lg = eng.head_forward(x, train=True)
    act = ref['act']
    for got, names in ((h6, ('drop6', '_[noisy]_drop6')), (h7, ('drop7', '_[noisy]_drop7'))):
        want = np.concatenate([act[n] for n in names], 1)
        m[names[0]] = _relmax(got, want)
    # ---- the head's GEMMs alone, per output UNIT.  Yardstick: the RMS over the proposals of the
    # unit's own pre-activation (float64) - the scale of its dot product, which a unit 12 octaves
    # below the loudest one keeps once the next layer's weights undo the factor (a unit's maximum
    # would not do: a nearly dead unit has a tiny maximum but the full rounding error of its
    # 25088-term sum).  Input: the oracle's roi_feat, so that the float64 arbiter sees the same
    # operand.  Every fp32 plan must stay below 1e-4 of that scale; the fp32 oracle's own distance
    # is printed beside it.
    xo = torch.from_numpy(ref['roi_feat'].reshape(R, -1)).to(dev)
    h6o, h7o, _lgo = eng.head_forward(xo, train=True)
    unit = {}
    for got, names, rms in ((h6o, ('drop6', '_[noisy]_drop6'), ('fc6_rms', '_[noisy]_fc6_rms')),
                            (h7o, ('drop7', '_[noisy]_drop7'), ('fc7_rms', '_[noisy]_fc7_rms'))):
        w64 = np.concatenate([arb['act'][n] for n in names], 1)
        w32 = np.concatenate([act[n] for n in names], 1).astype(np.float64)
        yard = 2.0 * np.concatenate([arb['act'][n] for n in rms])       # (Dropout scale 2)
        live = yard > 0
        e_hip = np.abs(got.cpu().numpy().astype(np.float64) - w64).max(axis=0)[live] / yard[live]
        e_orc = np.abs(w32 - w64).max(axis=0)[live] / yard[live]
        unit[names[0]] = (float(e_hip.max()), float(e_orc.max()),
                          float(np.median(e_hip)), float(np.median(e_orc)))
    print('[%s] per-unit max error / unit pre-activation RMS vs float64 (HIP max, fp32 oracle max, '
          'HIP median, oracle median): %s' % (mode_tag, ', '.join(
              '%s %.1e/%.1e/%.1e/%.1e' % ((k,) + v) for k, v in unit.items())))
    for k, (hmax, omax, hmed, omed) in unit.items():
        assert hmax <= 1e-4 and hmed <= 1e-5, (k, unit[k])
    del xo, h6o, h7o, _lgo
    ld8 = eng.ld8
    cols = [0, c, ld8, ld8 + c]
    logits = np.concatenate([act[k] for k in LOGIT_KEYS], 1)
    got_logits = torch.cat([lg[:, o:o + c] for o in cols], 1)
    m['logits'] = _relmax(got_logits, logits)
    lmax = float(np.abs(logits).max())
    print('\n[%s] max error / max|blob| vs oracle: %s; max|logit| %.2f' % (
        mode_tag, ', '.join('%s %.1e' % kv for kv in m.items()), lmax))
    for k, v in m.items():
        assert v < 1e-4, (k, v)
    del conv5, x, h6, h7, lg
    # ---- the whole step
    out = eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'])
    tl = ref['tails'][0]
    for k in ('loss_cls', 'loss_cls_noise'):
        a, b = float(out[k][0]), float(tl[k])
        assert abs(a - b) <= 1e-4 * abs(b), (k, a, b)
    # image-level probabilities span 8 orders of magnitude (1e-11 .. 1e-3 on random weights): a
    # logit error dz moves a probability by the factor exp(dz), so "logits within 1e-4 relative"
    # bounds the RELATIVE probability error by 1e-4 * max|logit| per softmax (two of them)
    # (measured: logits agree to ~1e-6 of max|logit| ~ 30, probabilities to ~1e-4 relative - and
    # the fp32 oracle is as far from the float64 evaluation as the HIP path is)
    ptol = 2e-4 * max(1.0, lmax)
    e64 = float(np.abs(got_logits.cpu().numpy() - arb['logits']).max() / np.abs(arb['logits']).max())
    o64 = float(np.abs(logits - arb['logits']).max() / np.abs(arb['logits']).max())
    print('[%s] logits vs float64: HIP %.1e, fp32 oracle %.1e' % (mode_tag, e64, o64))
    assert e64 <= 1e-5          # north_star asks 1e-4; the oracle's head alone is at 4e-7
    for k in ('cls_prob', 'cls_prob_noise'):
        got, want, p64 = out[k][0].cpu().numpy(), tl[k][0], arb[k][0]
        rel = float(np.abs(got / want - 1).max())
        r64, o64 = float(np.abs(got / p64 - 1).max()), float(np.abs(want / p64 - 1).max())
        print('[%s] %s max relative error: vs fp32 oracle %.1e; vs float64: HIP %.1e, fp32 '
              'oracle %.1e (bound %.1e)' % (mode_tag, k, rel, r64, o64, ptol))
        assert rel <= ptol and r64 <= ptol, (k, rel, r64)
    np.testing.assert_allclose(out['class_weight'][0].cpu().numpy(), tl['class_weight'][0],
                               rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(out['hatE_sum_norm'][0].cpu().numpy(), tl['hatE_sum_norm'][0],
                               rtol=1e-4, atol=1e-6)
    # ---- backward: both fp32 evaluations against the float64 arbiter
    dl64 = arb['d_logits']
    dl32 = np.concatenate([ref['d_logits'][k] for k in LOGIT_KEYS], 1).astype(np.float64)
    dlg = out['d_logits'].cpu().numpy().astype(np.float64)
    n64 = np.linalg.norm(dl64)
    e_hip, e_orc = np.linalg.norm(dlg - dl64) / n64, np.linalg.norm(dl32 - dl64) / n64
    print('[%s] d_logits normwise error vs float64: HIP %.1e, fp32 oracle %.1e' % (mode_tag, e_hip, e_orc))
    assert e_hip <= 1e-4, ('d_logits', e_hip, e_orc)
    report = {}
    for name, g64 in arb['grads'].items():
        got = eng.grad_blob(name).cpu().numpy().astype(np.float64)
        g32 = ref['grads'][name].astype(np.float64)
        if name.endswith('fc8d_b'):
            # identically zero in exact arithmetic (a softmax-over-proposals gradient sums to 0
            # down each column): both sides are sums of R rounding residues
            bound = 1e-5 * float(np.abs(dl64).max()) * np.sqrt(R)
            assert np.abs(got).max() <= bound and np.abs(g32).max() <= bound, name
            continue
        n = np.linalg.norm(g64)
        e_hip, e_orc = np.linalg.norm(got - g64) / n, np.linalg.norm(g32 - g64) / n
        report[name] = (e_hip, e_orc)
    print('\n[%s] normwise gradient error vs float64 (HIP, fp32 oracle): %s' % (
        mode_tag, ', '.join('%s %.1e/%.1e' % (k, a, b) for k, (a, b) in report.items())))
    # A gradient entry is (probability) x (activation): its relative error is the ABSOLUTE logit
    # error (here 3e-6 x max|logit| 58 = 1.7e-4, of which the conv body's 2e-6 - not seen by the
    # arbiter, which starts from the oracle's roi_feat - is the larger part).  Measured 1e-5
    # (fc6/fc7, clean fc8) .. 2e-4 (the noise branch's fc8, whose loss is 100x smaller); round 1
    # allowed 5e-3.
    # (skewed statistics: the noise branch's fc8d gradient is ill-conditioned enough that the fp32
    # oracle itself sits at 1.8e-3 from float64; there the HIP path must be no further than it)
    for name, (e_hip, e_orc) in report.items():
        bound = 5e-4
        if stats == 'skewed' and name == 'noisy_fc8d_w':
            bound = max(5e-4, e_orc)       # that one blob, that one case; 5e-4 elsewhere
        assert e_hip <= bound, (name, e_hip, e_orc)


def test_full_size_bf16_c80_matches_oracle(dev, ref20):
    """BASELINE configs[3] arithmetic (80 classes, bf16 MFMA conv / fc6 / fc7 operands, fp32
    storage, fc8, softmaxes and loss) on the full-size image: the conv body and the fc6 / fc7
    activations of the oracle are class-count independent (same seeded weights), so its C = 80
    run re-uses conv5_3 / roi_feat and adds the head forward + loss tails.  bf16 operands carry
    8 significant bits: conv5_3 / activations to 1e-2 normwise, logits to 3e-2 of max, losses to
    5e-2, weight-gradient direction cosine >= 0.99 (the tight check of the bf16 kernels is
    tests/test_gpu_bf16.py)."""
    from detectron.datasets import synthetic
    from oracle import oracle
    c = 80
    blobs = synthetic.init_blobs(c, seed=SEED)
    for k in ('conv5_3_w', 'fc6_w', '_[noisy]_fc7_w'):      # same stream -> same frozen weights
        assert torch.equal(blobs[k], ref20['blobs'][k])
    mb = dict(ref20['mb'])
    cls = 37
    mb['labels_oh'] = np.zeros((1, c), np.float32)
    mb['labels_oh'][0, cls] = 1
    mb['labels_int32'] = np.array([cls], np.int32)
    ref = oracle.full_forward_backward(blobs, mb, None, c, train=False, backward=True,
                                       conv5=ref20['ref']['conv5_3'],
                                       roi_feat=ref20['ref']['roi_feat'])
    t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
    eng = _engine(dev, c, blobs, 'bf16')
    eng.dropout = 0.0
    conv5 = eng.conv_body(t['data']).permute(0, 3, 1, 2).cpu().numpy()
    assert np.linalg.norm(conv5 - ref['conv5_3']) <= 1e-2 * np.linalg.norm(ref['conv5_3'])
    out = eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'])
    tl = ref['tails'][0]
    for k in ('loss_cls', 'loss_cls_noise'):
        a, b = float(out[k][0]), float(tl[k])
        assert np.isfinite(a) and abs(a - b) <= 5e-2 * abs(b), (k, a, b)
    cp, cpr = out['cls_prob'][0].cpu().numpy(), tl['cls_prob'][0]
    assert np.abs(cp - cpr).max() <= 5e-2 * cpr.max()
    # backward at size: every gradient finite, and the direction of each
    # weight gradient against the fp32 oracle's (dropout off on both sides; bf16 operands leave
    # ~1e-2 of noise per GEMM, so the measure is the cosine, as in test_engine_bf16_mode)
    assert torch.isfinite(eng.grads).all() and float(eng.grads.abs().max()) > 0
    cosines = {}
    for name in ('fc6_w', '_[noisy]_fc6_w', 'fc7_w', '_[noisy]_fc7_w', 'fc8c_w', 'fc8d_w',
                 'noisy_fc8c_w', 'noisy_fc8d_w', 'fc6_b', 'fc7_b'):
        g = eng.grad_blob(name).cpu().numpy().astype(np.float64).reshape(-1)
        r = ref['grads'][name].astype(np.float64).reshape(-1)
        cosines[name] = float(g.dot(r) / (np.linalg.norm(g) * np.linalg.norm(r) + 1e-300))
        ratio = np.linalg.norm(g) / (np.linalg.norm(r) + 1e-300)
        assert 0.9 <= ratio <= 1.1, (name, ratio)
    print('\n[bf16, C=80] gradient cosine vs fp32 oracle: ' +
          ', '.join('%s %.4f' % kv for kv in cosines.items()))
    for name, cs in cosines.items():
        assert cs >= 0.99, (name, cs)
