"""GPU parity of the fused execution plan (naws_hip.engine) against the CPU oracle on a
tiny end-to-end case: 2 synthetic 64x96 images, 24 / 16 rois, random weights -> conv5_3,
roi_feat, logits, loss_cls, loss_cls_noise, all 16 parameter gradients, 3 SGD steps."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def _setup(dev, c=20, dropout=0.5, seed=11, mfma_dtype='fp32'):
    from detectron.datasets import synthetic
    from naws_hip.engine import WsddnEngine
    roidb = synthetic.make_roidb(2, 24, c, 64, 96, seed=seed)
    roidb[1]['boxes'] = roidb[1]['boxes'][:16]
    roidb[1]['obn_scores'] = roidb[1]['obn_scores'][:16]
    roidb[1]['gt_classes'] = roidb[1]['gt_classes'][:16]
    mb = synthetic.make_minibatch(roidb, c)
    blobs = synthetic.init_blobs(c, seed=seed)
    for k in list(blobs):     # non-zero biases so the bias paths are exercised
        if k.endswith('_b'):
            blobs[k] = torch.randn(blobs[k].shape, generator=torch.Generator().manual_seed(1)) * 0.05
    eng = WsddnEngine(c + 1, dev, dropout=dropout, gpu_num=2, seed=seed, mfma_dtype=mfma_dtype)
    eng.set_conv_blobs(blobs)
    eng.set_head_blobs(blobs)
    return eng, mb, blobs


def _masks(eng, rt, dev):
    from naws_hip import ops
    if eng.dropout <= 0:
        return None
    m6 = ops.dropout_mask(eng._seed(6), eng.dropout, rt * 8192, dev).view(rt, 8192).cpu().numpy()
    m7 = ops.dropout_mask(eng._seed(7), eng.dropout, 2 * rt * 4096, dev).view(2, rt, 4096).cpu().numpy()
    return {'drop6': m6[:, :4096], '_[noisy]_drop6': m6[:, 4096:],
            'drop7': m7[0], '_[noisy]_drop7': m7[1]}


@pytest.mark.parametrize('mode', ['fp32', 'fp32x3', 'fp16x2'])
@pytest.mark.parametrize('dropout,c', [(0.5, 20), (0.0, 20), (0.5, 80), (0.5, 2), (0.5, 21)])
def test_engine_matches_oracle(dev, dropout, c, mode):
    """Both fp32 plans (fp32 MFMA everywhere / fc6+fc7 as exact 3xbf16 splits on the bf16 MFMA)
    are held to the same fp32 tolerances."""
    from oracle import oracle
    eng, mb, blobs = _setup(dev, c=c, dropout=dropout, mfma_dtype=mode)
    t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
    rt = mb['rois'].shape[0]
    masks = _masks(eng, rt, dev)
    out = eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'])
    ref = oracle.full_forward_backward(blobs, mb, masks, c, train=dropout > 0)
    conv5 = eng.conv_body(t['data']).permute(0, 3, 1, 2)
    assert _rel(conv5, ref['conv5_3']) < 1e-4
    for i in range(2):
        tl = ref['tails'][i]
        assert abs(float(out['loss_cls'][i]) - tl['loss_cls']) <= 1e-4 * abs(tl['loss_cls'])
        assert abs(float(out['loss_cls_noise'][i]) - tl['loss_cls_noise']) <= \
            1e-4 * abs(tl['loss_cls_noise'])
        np.testing.assert_allclose(out['cls_prob'][i].cpu().numpy(), tl['cls_prob'][0], rtol=1e-4,
                                       atol=1e-9)      # probabilities down to 1e-12 exist
        np.testing.assert_allclose(out['class_weight'][i].cpu().numpy(), tl['class_weight'][0],
                                   rtol=1e-4, atol=1e-6)
    dl = np.concatenate([ref['d_logits'][k] for k in ('fc8c', 'fc8d', 'noisy_fc8c', 'noisy_fc8d')], 1)
    # logits agree to ~1e-6 relative (GEMM summation order); where one proposal dominates a
    # class, alpha_cls - cls_prob cancels to ~1e-3 of its operands and that 1e-6 becomes
    # ~1e-3 of the gradient entry (checked against a float64 evaluation): 2e-3 of the max
    assert _rel(out['d_logits'], dl) < 2e-3
    # fc8d_b's gradient is identically zero in exact arithmetic (a softmax-over-proposals
    # gradient sums to 0 down each column): compare against the rounding floor instead
    floor = 1e-6 * float(np.abs(dl).max()) * rt
    for name, g in ref['grads'].items():
        got = eng.grad_blob(name).cpu().numpy()
        if name.endswith('fc8d_b'):
            # exactly 0 in exact arithmetic: both sides are sums of rt rounding residues
            bound = 2e-3 * float(np.abs(dl).max()) * np.sqrt(rt)
            assert np.abs(got).max() <= bound and np.abs(g).max() <= bound, name
            continue
        # A pre-activation within rounding of 0 can land on opposite sides of the ReLU gate in
        # two fp32 evaluations (seen: 1 of 160k units), which changes one row/column of a
        # weight gradient by O(1): bound the Frobenius error and the share of outliers
        # instead of the max (measured: d_logits agree to 1e-5).
        err = np.abs(got - g)
        tol = 2e-3 * np.abs(g).max() + floor
        assert np.linalg.norm(err) <= 5e-3 * np.linalg.norm(g) + floor * np.sqrt(g.size), name
        assert (err > tol).mean() <= 2e-3, name


@pytest.mark.parametrize('c', [20, 80])
def test_engine_bf16_mode(dev, c):
    """BASELINE configs[3]: bf16 MFMA conv/fc with fp32 storage and loss.  Operands are rounded
    to 8 significant bits, so conv5_3 / logits / gradients agree with the fp32 oracle to a few
    1e-3 .. 1e-2 (norm-wise), the dropout-free loss to 5e-2, gradients to 15% / cosine 0.99
    (the tight check of the bf16 kernels themselves is tests/test_gpu_bf16.py: 5e-6 against a
    float64 product of the same rounded operands)."""
    from oracle import oracle
    eng, mb, blobs = _setup(dev, c=c, dropout=0.0, mfma_dtype='bf16')
    t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
    out = eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'])
    ref = oracle.full_forward_backward(blobs, mb, None, c, train=False)
    conv5 = eng.conv_body(t['data']).permute(0, 3, 1, 2).cpu().numpy()
    assert np.linalg.norm(conv5 - ref['conv5_3']) <= 1e-2 * np.linalg.norm(ref['conv5_3'])
    for i in range(2):
        tl = ref['tails'][i]
        for k in ('loss_cls', 'loss_cls_noise'):
            got, want = float(out[k][i]), float(tl[k])
            assert abs(got - want) <= 5e-2 * abs(want), (k, i, got, want)
    for name, g in ref['grads'].items():
        if name.endswith('fc8d_b'):
            continue          # identically zero in exact arithmetic (see the fp32 test)
        # the logits carry ~1e-2 relative error into exp(): the softmax-over-proposals gradient
        # moves by several percent (measured 7% on fc6_w), direction preserved
        got = eng.grad_blob(name).cpu().numpy().astype(np.float64).ravel()
        gr = g.astype(np.float64).ravel()
        assert np.linalg.norm(got - gr) <= 0.15 * np.linalg.norm(gr) + 1e-7, name
        assert got @ gr >= 0.99 * np.linalg.norm(got) * np.linalg.norm(gr), name
    # the bf16 plan is deterministic too
    out2 = eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'])
    assert torch.equal(out['loss_cls'], out2['loss_cls'])


@pytest.mark.parametrize('mode', ['fp32', 'fp32x3', 'fp16x2'])
def test_engine_sgd_steps(dev, mode):
    """3 iterations of fwd+bwd+SGD vs the oracle (dropout masks replayed)."""
    from oracle import oracle
    eng, mb, blobs = _setup(dev, mfma_dtype=mode)
    t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
    rt = mb['rois'].shape[0]
    cur = {k: v.clone() for k, v in blobs.items()}
    state = {}
    lr = np.float32(1e-3)
    eng.set_lr(lr)
    for it in range(3):
        masks = _masks(eng, rt, dev)
        eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'])
        eng.sgd_step()
        ref = oracle.full_forward_backward(cur, mb, masks, 20)
        for name, g in ref['grads'].items():
            p = cur[name].numpy().reshape(-1).copy()
            st = state.setdefault(name, dict(m=np.ones_like(p), a=np.ones_like(p), it=0))
            bias = name.endswith('_b')
            st['it'] = oracle.acm_sgd(np.ascontiguousarray(g.reshape(-1)), st['m'],
                                      np.array([lr], np.float32), p, st['a'], 0.9, 0,
                                      0.0 if bias else 5e-4, 1, 2, 2.0 if bias else 1.0, st['it'])
            cur[name] = torch.from_numpy(p.reshape(cur[name].shape))
    for name in ref['grads']:
        assert _rel(eng.blob(name), cur[name].numpy()) < 1e-5, name
        m_ref = state[name]['m'].reshape(cur[name].shape)
        m_got = eng.momentum_blob(name).cpu().numpy()
        # fc8d_b's gradient is pure rounding noise (see test above): absolute floor
        assert np.abs(m_got - m_ref).max() <= 2e-3 * np.abs(m_ref).max() + 1e-9, name


@pytest.mark.parametrize('mode', ['fp32', 'fp32x3', 'fp16x2'])
def test_engine_infer(dev, mode):
    from oracle import oracle
    eng, mb, blobs = _setup(dev, dropout=0.5, mfma_dtype=mode)
    t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
    one = {k: v[:1] if k in ('data', 'labels_oh') else v for k, v in mb.items()}
    sel = mb['rois'][:, 0] == 0
    cls_prob = eng.infer(t['data'][:1], t['rois'][sel], t['obn_scores'][sel])
    one['rois'], one['obn_scores'] = mb['rois'][sel], mb['obn_scores'][sel]
    ref = oracle.full_forward_backward(blobs, one, None, 20, train=False)
    rp = ref['tails'][0]['rois_pred']
    np.testing.assert_allclose(cls_prob.cpu().numpy(), np.concatenate([rp[:, :1], rp], 1),
                               rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize('mode', ['fp32', 'fp32x3', 'fp16x2'])
def test_deferred_update_is_equivalent(dev, mode):
    """The N>1 schedule (all-reduce launched after backward, SGD applied after the NEXT
    iteration's conv body) gives bit-identical parameters to the immediate update (fp32x3: the
    weight planes are re-split on the update stream, the head must see the new ones)."""
    res = []
    for defer in (False, True):
        eng, mb, _blobs = _setup(dev, mfma_dtype=mode)
        eng.defer_update = defer
        t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
        eng.set_lr(1e-3)
        for it in range(3):
            if it == 2:
                eng.set_lr(1e-4)           # lr change must not leak into the pending update
            eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'])
            eng.sgd_step()
        eng.flush()
        res.append((eng.params.clone(), eng.momentum_buf.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def test_rccl_chunked_allreduce_single_rank(dev):
    """The distributed schedule on one GPU: a 1-rank RCCL group forced active, fc6_w's gradient
    exchanged in four row chunks behind the wgrad GEMMs, the small gradients last - identical
    gradients and parameters to the plain single-process run."""
    import os
    import torch.distributed as dist
    from detectron.datasets import synthetic
    from naws_hip.engine import WsddnEngine
    c = 20
    blobs = synthetic.init_blobs(c, seed=5)
    mb = synthetic.make_minibatch(synthetic.make_roidb(2, 300, c, 160, 224, seed=9), c, max_rois=300)
    t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
    created = False
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29541')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        created = True
    try:
        res = []
        # plain run; RCCL with the update queued by sgd_step(); RCCL with train_step(), which queues
        # the piece-by-piece update's parts from inside backward (round 5) - real NCCL work handles
        # waited for on the update stream while backward is still being enqueued
        for forced, train_step in ((False, False), (True, False), (True, True)):
            eng = WsddnEngine(c + 1, dev, dropout=0.5, gpu_num=2, seed=3,
                              process_group=dist.group.WORLD if forced else None, world_size=1,
                              allreduce_chunks=4)
            eng.reducer.force = forced
            eng.set_conv_blobs(blobs)
            eng.set_head_blobs(blobs)
            eng.set_lr(1e-4)
            for _ in range(3):
                if train_step:
                    eng.train_step(t['data'], t['rois'], t['obn_scores'], t['labels_oh'])
                else:
                    eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'])
                    eng.sgd_step()
            eng.flush()
            torch.cuda.synchronize()
            res.append((eng.grads.clone(), eng.params.clone(), eng.momentum_buf.clone()))
            assert eng.reducer.active == forced and eng._pipelined() == forced
            del eng
        for other in res[1:]:
            for a, b in zip(res[0], other):
                assert torch.equal(a, b)
    finally:
        if created:
            dist.destroy_process_group()


def test_weight_planes_after_fused_update_match_fresh_split(dev):
    """fp16x2.  (1) fused_planes off: the SGD kernel reports the updated rows' maxima and the
    re-split reads the weights once - the planes equal a from-scratch split bit for bit.
    (2) fused_planes on (default): the SGD kernel writes the planes itself, scaled from twice the
    row maximum before the update - scales may be one power of two smaller than a fresh split's,
    the planes reconstruct the same fp32 weights to 2^-22 relative, and a training run gives the
    same losses and parameters as (1) to fp32-accumulation level."""
    from naws_hip import ops
    runs = {}
    for fused_planes in (False, True):
        eng, mb, _blobs = _setup(dev, mfma_dtype='fp16x2')
        eng.fused_planes = fused_planes
        t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
        eng.set_lr(1e-2)
        losses = []
        for _ in range(3):
            out = eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'])
            losses.append(float(out['loss_cls'].sum() + out['loss_cls_noise'].sum()))
            eng.sgd_step()
        eng.flush()
        torch.cuda.synchronize()
        assert eng._rm_table is not None and not eng._planes_dirty
        assert (eng._sgd_regions is not None) and int(eng._wovf.item()) == 0
        w6, w7 = eng._weight_views()
        for key, fresh in (('w6', ops.split_f16x2(w6)), ('w7', ops.split_f16x2(w7)),
                           ('w7t', ops.split_f16x2(w7, transpose=True))):
            got = eng._wplanes[key]
            if not fused_planes or key == 'w7t':
                assert torch.equal(got.inv_scale, fresh.inv_scale), key
                assert torch.equal(got.planes.view(torch.int16), fresh.planes.view(torch.int16)), key
                continue
            ratio = got.inv_scale / fresh.inv_scale
            assert bool(((ratio == 1) | (ratio == 2)).all()), key
            w = (w6 if key == 'w6' else w7).reshape(-1, got.planes.shape[-3] * 16).double()
            p = got.planes.double()
            d = (p[0] + p[1])
            d = d.unsqueeze(0) if d.dim() == 3 else d
            dense = d.permute(0, 2, 1, 3).reshape(w.shape) * got.inv_scale.reshape(-1).double()[:, None]
            rowmax = w.abs().amax(dim=1, keepdim=True)
            assert bool(((dense - w).abs() <= torch.maximum(w.abs() * 2.0 ** -22,
                                                            rowmax * 2.0 ** -36)).all()), key
        runs[fused_planes] = (losses, eng.params.clone())
    for a, b in zip(runs[False][0], runs[True][0]):
        assert abs(a - b) <= 1e-5 * abs(a), (runs[False][0], runs[True][0])
    pa, pb = runs[False][1], runs[True][1]
    assert float((pa - pb).abs().max()) <= 1e-5 * float(pa.abs().max())


@pytest.mark.parametrize('mode', ['fp16x2', 'bf16'])
def test_where_the_deferred_update_is_queued_does_not_change_the_result(dev, mode):
    """engine.UPDATE_AFTER (how many layers of each image's conv chain run before the deferred
    update is queued beside them: 1 = conv1_1, 2 = + conv1_2 / pool1, 5 = + conv2_x / pool2) only
    moves launches between streams: parameters after three steps are bit-identical - and, in the
    plans without operand scales, identical to one conv chain over the whole batch too."""
    runs = {}
    for ua, streams in ((1, True), (2, True), (5, True), (1, False)):
        eng, mb, _blobs = _setup(dev, mfma_dtype=mode)
        eng.UPDATE_AFTER, eng.conv_streams = ua, streams
        t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
        eng.set_lr(1e-3)
        for _ in range(3):
            out = eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'])
            eng.sgd_step()
        eng.flush()
        torch.cuda.synchronize()
        runs[(ua, streams)] = (eng.params.clone(), out['loss_cls'].clone())
    ref = runs[(1, True)]
    for key in ((2, True), (5, True)) + (((1, False),) if mode == 'bf16' else ()):
        assert torch.equal(runs[key][0], ref[0]) and torch.equal(runs[key][1], ref[1]), key


@pytest.mark.parametrize('mode', ['fp32x3', 'bf16'])
def test_weight_planes_written_by_the_update_equal_a_fresh_split(dev, mode):
    """fp32x3 / bf16 plans: the SGD kernel also writes the fc6_w / fc7_w operand planes (exact
    3 x bf16 split / one rounded bf16 plane: no scales involved), so after training steps they
    must equal a from-scratch split of the updated parameters bit for bit - and the run must be
    bit-identical to the one that re-splits after every update."""
    from naws_hip import ops
    cv = ops.split_bf16x3 if mode == 'fp32x3' else ops.to_bf16_slab
    runs = {}
    for fused in (True, False):
        eng, mb, _blobs = _setup(dev, mfma_dtype=mode)
        eng.fused_planes = fused
        t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
        eng.set_lr(1e-2)
        for _ in range(3):
            eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'])
            eng.sgd_step()
        eng.flush()
        torch.cuda.synchronize()
        assert not eng._planes_dirty and (eng._sgd_regions is not None)
        w6, w7 = eng._weight_views()
        for key, fresh in (('w6', cv(w6)), ('w7', cv(w7)), ('w7t', cv(w7, transpose=True))):
            assert torch.equal(eng._wplanes[key].view(torch.int16), fresh.view(torch.int16)), (fused, key)
        runs[fused] = (eng.params.clone(), eng.momentum_buf.clone())
    assert torch.equal(runs[True][0], runs[False][0]) and torch.equal(runs[True][1], runs[False][1])


def test_train_step_updates_fc6_in_the_wgrad_epilogue_bit_identically(dev):
    """train_step(): with no gradient exchange fc6_w's update runs inside its weight-gradient
    GEMM (no gradient blob, engine.train_step).  Parameters, momentum and all three operand plane
    sets after four iterations (an lr change in between) equal forward_backward() + sgd_step()
    bit for bit; the fused route really ran (its gradient slot stays untouched)."""
    res = []
    for fused in (False, True):
        eng, mb, _blobs = _setup(dev, mfma_dtype='fp16x2')
        eng.fuse_wgrad_update = fused
        t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
        eng.set_lr(1e-3)
        gw6 = eng.arena.span(eng.grads, 'fc6_w', '_[noisy]_fc6_w')
        for it in range(4):
            if it == 2:
                eng.set_lr(1e-4)
            if it == 3:
                eng.flush()              # (the pending update of iteration 2 still reads the gradients)
                gw6.fill_(123.0)
            out = eng.train_step(t['data'], t['rois'], t['obn_scores'], t['labels_oh'])
            assert eng._can_fuse_wgrad_update() == fused
        eng.flush()
        torch.cuda.synchronize()
        assert bool((gw6 == 123.0).all()) == fused
        wp = eng._wplanes
        res.append((eng.params.clone(), eng.momentum_buf.clone(),
                    [wp[k].planes.view(torch.int16).clone() for k in ('w6', 'w7', 'w7t')],
                    [wp[k].scales.clone() for k in ('w6', 'w7')], out['loss_cls'].clone(),
                    int(eng._wovf.item())))
    a, b = res
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert all(torch.equal(x, y) for x, y in zip(a[2], b[2]))
    assert all(torch.equal(x, y) for x, y in zip(a[3], b[3]))
    assert torch.equal(a[4], b[4]) and a[5] == b[5]


def test_train_step_bf16_plan_updates_fc6_in_the_wgrad_epilogue_bit_identically(dev):
    """The bf16 plan's form of the same route (ops.gemm_bf16_slab_nt_sgd): parameters, momentum and
    the rounded operand planes after four iterations equal forward_backward() + sgd_step() bit for
    bit, and the fused route really ran."""
    res = []
    for fused in (False, True):
        eng, mb, _blobs = _setup(dev, mfma_dtype='bf16')
        eng.fuse_wgrad_update = fused
        t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
        eng.set_lr(1e-3)
        gw6 = eng.arena.span(eng.grads, 'fc6_w', '_[noisy]_fc6_w')
        for it in range(4):
            if it == 2:
                eng.set_lr(1e-4)
            if it == 3:
                eng.flush()
                gw6.fill_(123.0)
            out = eng.train_step(t['data'], t['rois'], t['obn_scores'], t['labels_oh'])
            assert eng._can_fuse_wgrad_update() == fused
        eng.flush()
        torch.cuda.synchronize()
        assert bool((gw6 == 123.0).all()) == fused
        wp = eng._wplanes
        res.append((eng.params.clone(), eng.momentum_buf.clone(),
                    [wp[k].view(torch.int16).clone() for k in ('w6', 'w7', 'w7t')],
                    out['loss_cls'].clone()))
    a, b = res
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert all(torch.equal(x, y) for x, y in zip(a[2], b[2]))
    assert torch.equal(a[3], b[3])
    w6 = eng.arena.span(eng.params, 'fc6_w', '_[noisy]_fc6_w').view(2 * 4096, -1)
    from naws_hip import ops
    assert torch.equal(ops.to_bf16_slab(w6).view(torch.int16), eng._wplanes['w6'].view(torch.int16))


def test_train_step_with_an_exchange_or_other_plans_is_the_two_call_form(dev):
    """train_step() must not fuse when gradients are exchanged (reducer active) or in a plan
    without fp16x2 planes: fc6_w's gradient is written as before."""
    for kw, force in ((dict(mfma_dtype='fp32x3'), False), (dict(mfma_dtype='fp16x2'), True)):
        eng, mb, _blobs = _setup(dev, **kw)
        if force:
            class _Red(object):          # an "active" reducer that reduces over one rank
                active = True
                world_size = 1
                def reduce_async(self, _t): pass
                def wait(self): pass
            eng.reducer = _Red()
        t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
        eng.set_lr(1e-3)
        gw6 = eng.arena.span(eng.grads, 'fc6_w', '_[noisy]_fc6_w')
        gw6.fill_(123.0)
        eng.train_step(t['data'], t['rois'], t['obn_scores'], t['labels_oh'])
        eng.flush()
        assert not eng._can_fuse_wgrad_update()
        assert not bool((gw6 == 123.0).all())


def test_checkpoint_save_mid_run_leaves_the_trajectory_bit_identical(dev):
    """export_blobs() (every checkpoint save) must not mark the operand planes dirty -
    the step after a save would drop the wgrad-epilogue route and re-split with exact instead of
    bound-derived scales, i.e. the fp16x2 trajectory would depend on the checkpoint cadence."""
    res = []
    for save in (False, True):
        eng, mb, _blobs = _setup(dev, mfma_dtype='fp16x2')
        t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
        eng.set_lr(1e-3)
        for it in range(4):
            eng.train_step(t['data'], t['rois'], t['obn_scores'], t['labels_oh'])
            if save and it in (1, 2):
                snap = {k: v.cpu() for k, v in eng.export_blobs().items()}
                assert 'fc6_w' in snap and 'fc6_w_momentum' in snap
                assert not eng._planes_dirty and eng._can_fuse_wgrad_update()
        eng.flush()
        torch.cuda.synchronize()
        wp = eng._wplanes
        res.append((eng.params.clone(), eng.momentum_buf.clone(),
                    [wp[k].planes.view(torch.int16).clone() for k in ('w6', 'w7', 'w7t')]))
    a, b = res
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert all(torch.equal(x, y) for x, y in zip(a[2], b[2]))


@pytest.mark.parametrize('mode', ['fp16x2', 'fp32x3'])
def test_plane_writing_update_needs_one_hyper_parameter_run_per_region(dev, mode):
    """the plane-writing SGD kernel applies ONE (lr_mult, weight decay) per region.  A
    configuration whose clean and noisy fc6_w differ in either must fall back to the element-wise
    kernel (+ re-split): same parameters as the oracle's per-blob update, no silent use of the
    clean branch's settings."""
    eng, mb, _blobs = _setup(dev, mfma_dtype=mode)
    t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
    o, n, _shape = eng.arena.offsets['_[noisy]_fc6_w']
    ends, lrm, wd = eng._seg_host
    assert eng._one_hyper_run(eng.arena.offsets['fc6_w'][0], 2 * n)
    # split the first run at the noisy branch: it gets lr_mult 10 (a '_lrm10_' blob would)
    i = next(k for k, e in enumerate(ends) if e > o)
    ends2 = ends[:i] + [o, o + n] + ([ends[i]] if ends[i] > o + n else []) + ends[i + 1:]
    lrm2 = lrm[:i] + [lrm[i], 10.0] + ([lrm[i]] if ends[i] > o + n else []) + lrm[i + 1:]
    wd2 = wd[:i] + [wd[i], wd[i]] + ([wd[i]] if ends[i] > o + n else []) + wd[i + 1:]
    eng._seg_host = (ends2, lrm2, wd2)
    eng.seg_end = torch.tensor(ends2, dtype=torch.int64, device=dev)
    eng.seg_lr_mult = torch.tensor(lrm2, dtype=torch.float32, device=dev)
    eng.seg_wd = torch.tensor(wd2, dtype=torch.float32, device=dev)
    assert not eng._one_hyper_run(eng.arena.offsets['fc6_w'][0], 2 * n)
    eng.set_lr(1e-3)
    p0 = eng.params.clone()
    eng.train_step(t['data'], t['rois'], t['obn_scores'], t['labels_oh'])
    eng.flush()
    assert eng._sgd_regions is None and not eng._can_fuse_wgrad_update()
    g = eng.grads
    # first step, zero momentum: dp = lr * lr_mult * (g / gpu_num + wd * p)
    for name, mult in (('fc6_w', 1.0), ('_[noisy]_fc6_w', 10.0)):
        gp = eng.arena.view(g, name).double() / eng.gpu_num + 5e-4 * eng.arena.view(p0, name).double()
        want = eng.arena.view(p0, name).double() - 1e-3 * mult * gp
        got = eng.arena.view(eng.params, name).double()
        assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max()), name


def test_set_lr_reproduces_the_reference_momentum_correction(dev):
    """UpdateWorkspaceLr / _SetNewLr / _CorrectMomentum (detector.py:509-559) captured from the
    imported reference with a recording workspace (tests/golden/make_golden_lr_update.py): for
    every lr of the sequence the engine holds the same lr and has scaled - or not scaled - the
    momentum arena by the same float32 factor, bit for bit."""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden',
                                       'reference_lr_update.json')))
    eng, _mb, _blobs = _setup(dev, mfma_dtype='fp32')
    assert eng.scale_momentum == gold['SCALE_MOMENTUM']
    assert eng.scale_momentum_threshold == gold['SCALE_MOMENTUM_THRESHOLD']
    g = torch.Generator(device='cpu').manual_seed(5)
    ref = torch.randn((4096,), generator=g)
    eng.momentum_buf[:4096].copy_(ref.to(dev))
    want = ref.numpy().copy()
    for case in gold['cases']:
        assert np.float32(eng._lr_host) == np.float32(case['cur_lr'])
        eng.set_lr(case['new_lr'])
        assert np.float32(eng._lr_host) == np.float32(case['returned'])
        assert float(eng.lr.item()) == (case['fed'][-1] if case['fed'] else float(np.float32(case['cur_lr'])))
        if case['correction'] is not None:
            want = want * np.float32(case['correction'])
        assert np.array_equal(eng.momentum_buf[:4096].cpu().numpy(), want), case


def test_cross_plan_soak_short(dev):
    """tools/soak_crossplan.py at a reduced size: 40 training steps of the 2 x f16 split plan,
    the fp32-MFMA plan and the exact 3 x bf16 split in lockstep; the headline plan must track
    the fp32-MFMA plan's loss trajectory (1e-3) for at least 0.6 of the steps the other fp32 ordering
    does (horizons of a chaotic map: see tools/soak_crossplan.report)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    'na-fwebsod_amd', 'tools'))
    import soak_crossplan as sc
    traj = sc.run(['fp16x2', 'fp32', 'fp32x3'], steps=40, lr=1e-4, rois=300, height=256, width=384,
                  device=dev)
    lines, ok = sc.report(traj, 'fp16x2', 'fp32', 'fp32x3')
    print('\n' + '\n'.join(lines))
    assert ok
    assert sc.horizon(traj['fp16x2'], traj['fp32'], 1e-3)[0] >= 8


def test_per_image_pooling_on_the_conv_streams_is_bit_identical(dev):
    """engine.conv_body(roi_job=...) pools each image's proposals at the tail of that image's conv
    chain (naws_roi_pool_f_f16x2_mapped_range_fwd on the image's stream): fc6's operand planes and
    scales, the losses and every gradient equal the one-launch-after-the-join route bit for bit;
    ragged proposal counts, one image without proposals of its own in the middle of the batch."""
    from detectron.datasets import synthetic
    from naws_hip.engine import WsddnEngine
    c = 20
    blobs = synthetic.init_blobs(c, seed=3)
    roidb = synthetic.make_roidb(3, 40, c, 64, 96, seed=4)
    for e, r in zip(roidb, (37, 5, 40)):
        for k in ('boxes', 'obn_scores', 'gt_classes'):
            e[k] = e[k][:r]
    mb = synthetic.make_minibatch(roidb, c)
    t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
    seg = [0, 37, 42, 82]
    res = []
    for on in (True, False):
        eng = WsddnEngine(c + 1, dev, dropout=0.5, gpu_num=3, seed=3, mfma_dtype='fp16x2')
        eng.ROI_POOL_ON_CHAINS = on
        eng.set_conv_blobs(blobs)
        eng.set_head_blobs(blobs)
        conv5 = eng.conv_body(t['data'], roi_job=(t['rois'], t['obn_scores'], seg))
        assert (eng._roi_operand is not None) == on
        x = eng._roi_features(conv5, t['rois'], t['obn_scores'])
        planes, scales = x.planes.clone(), x.inv_scale.clone()
        out = eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'], seg=seg)
        torch.cuda.synchronize()
        res.append((planes, scales, out['loss_cls'].clone(), out['loss_cls_noise'].clone(),
                    eng.grads.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert torch.isfinite(res[0][2]).all()


def test_default_plan_runs_winograd4_on_the_512_channel_layers_and_f2_agrees(dev):
    """The default plan's conv forms (engine.set_conv_blobs): conv4_1 .. conv5_3 carry the 36-frequency
    planes of Winograd F(4x4,3x3) (csrc/winograd4.hip), conv3_x the F(2x2) planes + the direct kernel's
    (chosen per input size), conv1_2 .. conv2_2 the direct kernel's; WINO_F4_MIN_CIN = 0 brings the
    F(2x2) plan of rounds 1-5 back, and the two conv bodies agree to 1e-5 of max|conv5_3|."""
    from naws_hip import ops
    from naws_hip.engine import WsddnEngine
    eng, mb, blobs = _setup(dev, mfma_dtype='fp16x2')
    for name in ('conv4_1', 'conv4_2', 'conv4_3', 'conv5_1', 'conv5_2', 'conv5_3'):
        wp = eng.conv[name][0]
        assert isinstance(wp, ops.F16x2) and wp.planes.dim() == 5 and wp.planes.shape[1] == 36, name
        assert name not in eng.conv_direct_h2
    for name in ('conv3_2', 'conv3_3'):
        assert eng.conv[name][0].planes.shape[1] == 16 and name in eng.conv_direct_h2
    for name in ('conv1_2', 'conv2_1', 'conv2_2'):
        assert eng.conv[name][0].planes.dim() == 4
    data = torch.from_numpy(mb['data']).to(dev)
    y4 = eng.conv_body(data).clone()
    eng2 = WsddnEngine(21, dev, dropout=0.5, gpu_num=2, seed=11, mfma_dtype='fp16x2')
    eng2.WINO_F4_MIN_CIN = 0
    eng2.set_conv_blobs(blobs)
    assert eng2.conv['conv4_2'][0].planes.shape[1] == 16
    y2 = eng2.conv_body(data)
    assert float((y4 - y2).abs().max()) <= 1e-5 * float(y2.abs().max())
