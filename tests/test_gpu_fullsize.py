"""Full-size (BASELINE.json configs[1]: 2 images 600x1000, 2000 proposals each) checks through
size-independent properties — the oracle is too slow at this size."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def setup(dev):
    from detectron.datasets import synthetic
    from naws_hip.engine import WsddnEngine
    c = 20
    blobs = synthetic.init_blobs(c, seed=11)
    mb = synthetic.make_minibatch(synthetic.make_roidb(2, 2000, c, 600, 1000, seed=11), c)
    t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}

    def make(gpu_num=2):
        eng = WsddnEngine(c + 1, dev, gpu_num=gpu_num, seed=11)
        eng.set_conv_blobs(blobs)
        eng.set_head_blobs(blobs)
        eng.set_lr(1e-5)
        return eng
    return make, t, mb


def test_probabilities_and_zero_sum_gradients(setup):
    make, t, mb = setup
    eng = make()
    out = eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'])
    cp = out['cls_prob'].cpu().numpy()
    assert cp.shape == (2, 20) and (cp >= 0).all() and (cp <= 1 + 1e-5).all()
    rp = out['rois_pred'].double()
    # image-level score = sum of its proposals' scores
    np.testing.assert_allclose(rp[:2000].sum(0).cpu().numpy(), cp[0], rtol=1e-5)
    np.testing.assert_allclose(rp[2000:].sum(0).cpu().numpy(), cp[1], rtol=1e-5)
    # sum_c sum_r rois_pred = sum_c cls_prob <= number of classes, and each alpha is a softmax
    dl = out['d_logits'].double()
    scale = float(dl.abs().max())
    # softmax gradients are zero-sum: over classes per row (fc8c parts), over the image's
    # proposals per class (fc8d parts)
    for col0 in (0, 40):
        assert float(dl[:, col0:col0 + 20].sum(1).abs().max()) <= 1e-5 * scale
    for col0 in (20, 60):
        for sl in (slice(0, 2000), slice(2000, 4000)):
            assert float(dl[sl, col0:col0 + 20].sum(0).abs().max()) <= 1e-4 * scale
    for k in ('loss_cls', 'loss_cls_noise'):
        assert torch.isfinite(out[k]).all() and (out[k] > 0).all()
    w = out['class_weight'].cpu().numpy()
    lab = mb['labels_oh']
    assert ((w >= 0) & (w <= 1)).all() and (w[lab == 1] == 1).all()   # labelled class never gated


def test_run_to_run_bitwise_reproducible(setup):
    make, t, _mb = setup
    res = []
    for _ in range(2):
        eng = make()
        out = eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'])
        eng.sgd_step()
        eng.flush()
        res.append((out['loss_cls'].clone(), eng.grads.clone(), eng.params.clone()))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])


def test_batch_of_two_equals_two_single_image_ranks(setup):
    """B = 2 images in one process == the reference's two GPUs with one image each: per-image
    losses identical, gradients = sum of the per-image gradients."""
    make, t, _mb = setup
    both = make()
    both.dropout = 0.0          # dropout streams are indexed by row: compare without it
    out = both.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'])
    gsum = torch.zeros_like(both.grads)
    for i in range(2):
        one = make()
        one.dropout = 0.0
        sel = t['rois'][:, 0] == i
        rois = t['rois'][sel].clone()
        rois[:, 0] = 0
        o = one.forward_backward(t['data'][i:i + 1].contiguous(), rois, t['obn_scores'][sel],
                                 t['labels_oh'][i:i + 1].contiguous())
        for k in ('loss_cls', 'loss_cls_noise'):
            a, b = float(out[k][i]), float(o[k][0])
            assert abs(a - b) <= 1e-5 * abs(b), (k, i, a, b)
        np.testing.assert_allclose(out['cls_prob'][i].cpu().numpy(), o['cls_prob'][0].cpu().numpy(),
                                   rtol=1e-5)
        gsum += one.grads
    err = float((both.grads - gsum).norm() / gsum.norm())
    assert err < 1e-5, err


def test_roi_pool_full_image_roi_is_global_max(dev):
    """A proposal covering the whole image with a 1x1 output is the per-channel global max."""
    from naws_hip import ops
    feat = torch.rand((1, 74, 124, 512), device=dev)
    rois = torch.tensor([[0, 0, 0, 991, 591]], device=dev, dtype=torch.float32)
    y = ops.roi_pool_f(feat, rois, 1, 1, 0.125, layout='NHWC')
    assert torch.equal(y.reshape(-1), feat.reshape(-1, 512).max(0).values)


def test_sgd_identity_properties(dev):
    from naws_hip import ops
    n = 1 << 22
    p = torch.randn(n, device=dev); p0 = p.clone()
    m = torch.zeros(n, device=dev); g = torch.zeros(n, device=dev)
    lr = torch.tensor([1e-3], device=dev)
    ends = torch.tensor([n], dtype=torch.int64, device=dev)
    one = torch.ones(1, device=dev); zero = torch.zeros(1, device=dev)
    ops.acm_sgd_update(g, m, lr, p, None, ends, one, zero, 0.9, 0, 1, 8, 0)
    assert torch.equal(p, p0) and float(m.abs().max()) == 0.0     # zero grad, no decay: no-op
    g.fill_(8.0)
    ops.acm_sgd_update(g, m, lr, p, None, ends, one, zero, 0.0, 0, 1, 8, 1)
    assert torch.allclose(p, p0 - 1e-3, rtol=0, atol=1e-6)       # grad / gpu_num * lr


def test_config5_inference_1200x2000_4000_rois(dev):
    """BASELINE.json configs[4]: the largest TTA scale (1200x2000) with 4000 proposals, forward
    only.  Size-independent checks: (1) rois_pred is a product of two softmaxes (>= 0, column
    sums over proposals <= 1, background column = column 0 of the foreground block);
    (2) the softmax over proposals is permutation-equivariant; (3) the two fp32 plans - two
    independent kernel families - agree to the fp32 parity tolerance; (4) per-class NMS of the
    result on the GPU equals the host restatement."""
    from detectron.datasets import synthetic
    from detectron.core import test_wsl
    from naws_hip import ops
    from naws_hip.engine import WsddnEngine
    c = 20
    blobs = synthetic.init_blobs(c, seed=11)
    mb = synthetic.make_minibatch(synthetic.make_roidb(1, 4000, c, 1200, 2000, seed=13), c,
                                  max_rois=4000)
    t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
    assert t['rois'].shape[0] == 4000
    res = {}
    for mode in ('fp16x2', 'fp32x3', 'fp32'):
        eng = WsddnEngine(c + 1, dev, gpu_num=1, seed=11, mfma_dtype=mode)
        eng.set_conv_blobs(blobs)
        eng.set_head_blobs(blobs)
        res[mode] = eng.infer(t['data'], t['rois'], t['obn_scores'])
        if mode == 'fp16x2':
            perm = torch.randperm(4000, generator=torch.Generator().manual_seed(1)).to(dev)
            p2 = eng.infer(t['data'], t['rois'][perm], t['obn_scores'][perm])
        del eng
    p = res['fp16x2']
    assert p.shape == (4000, c + 1) and torch.isfinite(p).all() and (p >= 0).all()
    assert torch.equal(p[:, 0], p[:, 1])
    assert float(p[:, 1:].sum(0).max()) <= 1.0 + 1e-5
    scale = float(p.max())
    assert float((p2 - p[perm]).abs().max()) <= 1e-5 * scale
    assert float((res['fp32'] - p).abs().max()) <= 1e-4 * scale
    assert float((res['fp32x3'] - p).abs().max()) <= 1e-4 * scale
    # NMS on the real score matrix: GPU == host restatement, class by class
    keep = ops.nms_per_class(t['rois'][:, 1:5].contiguous(), p[:, 1:].contiguous(), 0.0, 0.5)
    pn, bn = p.cpu().numpy(), mb['rois'][:, 1:5]
    for j in (1, 7, 20):
        inds = np.where(pn[:, j] > 0.0)[0]
        dets = np.hstack([bn[inds], pn[inds, j:j + 1]]).astype(np.float32)
        assert np.array_equal(np.where(keep[j - 1].cpu().numpy())[0], inds[test_wsl.nms(dets, 0.5)])


@pytest.mark.parametrize('h,w,rois', [(203, 317, (517, 301)), (480, 640, (999,)), (97, 131, (5, 1, 64))])
def test_ragged_shapes_plans_agree(dev, h, w, rois):
    """Odd image sizes, ragged proposal counts (not multiples of any tile / K-slab), 1-3 images
    per process: the fp16x2, fp32x3 and fp32 plans - disjoint GEMM / conv kernels - agree on losses,
    probabilities and parameter gradients to the fp32 parity tolerance, and the bf16 plan to
    its own."""
    from detectron.datasets import synthetic
    from naws_hip.engine import WsddnEngine
    c = 20
    b = len(rois)
    blobs = synthetic.init_blobs(c, seed=5)
    roidb = synthetic.make_roidb(b, max(rois), c, h, w, seed=9)
    for e, r in zip(roidb, rois):
        for k in ('boxes', 'obn_scores', 'gt_classes'):
            e[k] = e[k][:r]
        e['gt_classes'][0] = max(int(e['gt_classes'][0]), 1)
    mb = synthetic.make_minibatch(roidb, c, max_rois=4000)
    t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
    assert t['rois'].shape[0] == sum(rois)
    res = {}
    for mode in ('fp32', 'fp32x3', 'fp16x2', 'bf16'):
        eng = WsddnEngine(c + 1, dev, dropout=0.5, gpu_num=b, seed=3, mfma_dtype=mode)
        eng.set_conv_blobs(blobs)
        eng.set_head_blobs(blobs)
        out = eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'])
        torch.cuda.synchronize()
        res[mode] = (out, eng.grads.clone())
        del eng
    o32, g32 = res['fp32']
    for mode, tol, gtol in (('fp32x3', 1e-4, 2e-3), ('fp16x2', 1e-4, 2e-3), ('bf16', 5e-2, 0.2)):
        o, g = res[mode]
        lscale = float(o32['loss_cls'].abs().max())       # (the noise loss is ~100x smaller)
        for k in ('loss_cls', 'loss_cls_noise'):
            a, r = o[k].cpu().numpy(), o32[k].cpu().numpy()
            bound = tol * (np.abs(r).max() if mode != 'bf16' else lscale)
            assert np.isfinite(a).all() and np.abs(a - r).max() <= bound, (mode, k)
        pa, pr = o['cls_prob'].cpu().numpy(), o32['cls_prob'].cpu().numpy()
        if mode == 'bf16':
            # bf16 (no blanket percentage): the logits carry the operands' rounding
            # through 13 conv layers + fc6 / fc7 and are held to the plan's own logit tolerance
            # (3e-2 of max|logit|, as in test_full_size_bf16_c80_matches_oracle); GIVEN the
            # logits, cls_prob = sum_r softmax_c x softmax_d is a sum of positive terms each
            # moved by at most exp(2 dz) per softmax, so its relative error is bounded by
            # exp(4 dz) - 1 with dz the largest logit difference actually present
            la, lr_ = o['logits'].double().cpu().numpy(), o32['logits'].double().cpu().numpy()
            dz = float(np.abs(la - lr_).max())
            assert dz <= 3e-2 * float(np.abs(lr_).max()), (mode, dz)
            live = pr > 0
            assert np.abs(pa[live] / pr[live] - 1).max() <= np.expm1(4 * dz) * (1 + 1e-3) + 1e-5, mode
        else:
            assert np.abs(pa - pr).max() <= tol * pr.max(), mode
        ga, gr = g.double(), g32.double()
        assert float((ga - gr).norm()) <= gtol * float(gr.norm()), mode
