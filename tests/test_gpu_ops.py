"""GPU parity: every HIP operator (through the C ABI) against the CPU oracle.

Bit-exact for index work (RoI bins/argmax, IoU ints); fp32 results within 1e-4
relative (BASELINE.json north_star), usually far tighter.
"""
import numpy as np
import pytest
import torch

from helpers import make_rois, seg_offsets

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _close(a, b, rtol=RTOL, atol=1e-6):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, equal_nan=True)


# ---------------------------------------------------------------- RoIPoolF --
@pytest.mark.parametrize('layout,c', [('NCHW', 300), ('NHWC', 300), ('NHWC', 128), ('NHWC', 30)])
@pytest.mark.parametrize('with_boost', [False, True])
def test_roi_pool_bitexact(dev, layout, c, with_boost):
    from naws_hip import ops
    from oracle import oracle
    rng = np.random.default_rng(11)
    n, h, w = 2, 20, 30   # c = 300: float4 kernel with a ragged tile; 128: XCD-sliced; 30: scalar
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    x[:, :, 3, 4] = x[:, :, 3, 5]  # ties -> first index must win
    rois = make_rois(rng, n, 40, h * 8, w * 8)
    y_ref, am_ref = oracle.roi_pool_f(x, rois, 7, 7, 0.125)
    boost = (rng.uniform(0, 1, rois.shape[0]) + 1).astype(np.float32)
    if with_boost:
        y_ref = oracle.roi_feature_boost(y_ref, boost)
    xd = _t(x, dev)
    if layout == 'NHWC':
        xd = xd.permute(0, 2, 3, 1).contiguous()
    y, am = ops.roi_pool_f(xd, _t(rois, dev), 7, 7, 0.125, boost=_t(boost, dev) if with_boost
                           else None, layout=layout, with_argmax=True)
    assert np.array_equal(am.cpu().numpy(), am_ref)        # bit-exact index assignment
    assert np.array_equal(y.cpu().numpy(), y_ref)           # max is exact, boost is one fp32 mul
    y2 = ops.roi_pool_f(xd, _t(rois, dev), 7, 7, 0.125, boost=_t(boost, dev) if with_boost
                        else None, layout=layout)
    assert np.array_equal(y2.cpu().numpy(), y_ref)


@pytest.mark.parametrize('h,w,nroi', [(20, 30, 40), (74, 124, 300), (5, 7, 24)])
def test_roi_pool_hierarchical_bitexact(dev, h, w, nroi):
    """RoIPoolF over the precomputed 2x2 / 4x4 block maxima (naws_roi_pool_f_nhwc_hier_fwd,
    naws_roi_pool_f_f16x2_hier_fwd) == the oracle's pixel loop, value for value: windows of every
    size incl. 1-px, malformed, outside-the-image and whole-image rois, negative data and NaNs
    (the reference's strict '>' never lets a NaN win)."""
    from naws_hip import ops
    from oracle import oracle
    rng = np.random.default_rng(h * w)
    n, c = 2, 128
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    x[0, :, h // 2, w // 3] = np.nan
    x[1, 5, :, :] = -3.0                         # a constant negative channel
    rois = make_rois(rng, n, nroi, h * 8, w * 8)
    boost = (rng.uniform(0, 1, rois.shape[0]) + 1).astype(np.float32)
    y_ref = oracle.roi_feature_boost(oracle.roi_pool_f(x, rois, 7, 7, 0.125)[0], boost)
    xd = _t(x, dev).permute(0, 2, 3, 1).contiguous()
    y = ops.roi_pool_f(xd, _t(rois, dev), 7, 7, 0.125, boost=_t(boost, dev), layout='NHWC', hier=True)
    assert np.array_equal(y.cpu().numpy(), y_ref, equal_nan=True)
    # the operand-plane form: identical planes and scales to the direct kernel's
    amax = torch.full((n,), 0, device=dev, dtype=torch.int32)
    amax.copy_(torch.tensor([np.float32(8.0).view(np.int32)] * n))
    xz = torch.nan_to_num(xd)
    a = ops.roi_pool_f_f16x2(xz, _t(rois, dev), amax, 7, 7, 0.125, boost=_t(boost, dev), hier=True)
    b = ops.roi_pool_f_f16x2(xz, _t(rois, dev), amax, 7, 7, 0.125, boost=_t(boost, dev), hier=False)
    assert torch.equal(a.planes.view(torch.int16), b.planes.view(torch.int16))
    assert torch.equal(a.inv_scale, b.inv_scale)
    # the two halves as separate calls, the maps of each image built on their own (the engine
    # does that on the image's conv stream): identical planes
    m2, m4 = torch.empty_like(xz), torch.empty_like(xz)
    for i in range(n):
        ops.roi_maxmaps(xz[i:i + 1], m2[i:i + 1], m4[i:i + 1])
    c2 = ops.roi_pool_f_f16x2(xz, _t(rois, dev), amax, 7, 7, 0.125, boost=_t(boost, dev), maps=(m2, m4))
    assert torch.equal(c2.planes.view(torch.int16), b.planes.view(torch.int16))
    assert torch.equal(c2.inv_scale, b.inv_scale)


def test_emulate_exchange_copies_at_the_requested_pace(dev):
    """naws_emulate_exchange (bench.py's N-rank projection aid): copies exactly `bytes` and lasts
    about bytes / rate; argument checks."""
    from naws_hip import ops, lib
    n = 8 << 20
    src = torch.arange(n, device=dev, dtype=torch.float32)
    dst = torch.zeros_like(src)
    nbytes = 4 * n - 64
    t = []
    ops.emulate_exchange(src, dst, nbytes, 32, 1000.0)           # (first launch: code object load)
    for gbps in (600.0, 100.0):
        dst.zero_()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.emulate_exchange(src, dst, nbytes, 32, gbps)
        e.record()
        torch.cuda.synchronize()
        t.append(s.elapsed_time(e))
        k = nbytes // 4
        assert torch.equal(dst[:k], src[:k]) and not dst[k:].any()
        want = nbytes / (gbps * 1e9) * 1e3
        assert 0.9 * want <= t[-1] <= 1.6 * want + 0.2, (gbps, t[-1], want)
    with pytest.raises(lib.NawsError):
        lib.call('naws_emulate_exchange', src.data_ptr(), dst.data_ptr(), 1024, 0, 100.0, 0)
    with pytest.raises(ValueError):
        ops.emulate_exchange(src, dst, 4 * n + 16, 32, 100.0)


def test_roi_pool_empty_and_errors(dev):
    from naws_hip import ops, lib
    x = torch.zeros((1, 4, 5, 5), device=dev)
    y = ops.roi_pool_f(x, torch.zeros((0, 5), device=dev))
    assert y.shape == (0, 4, 7, 7)
    with pytest.raises(lib.NawsError):
        ops.roi_pool_f(x, torch.zeros((3, 4), device=dev))
    with pytest.raises(TypeError):
        ops.roi_pool_f(x.cpu(), torch.zeros((3, 5)))


def test_roi_feature_boost(dev):
    from naws_hip import ops, lib
    from oracle import oracle
    rng = np.random.default_rng(1)
    for f in (25088, 37):
        x = rng.standard_normal((17, f)).astype(np.float32)
        s = rng.uniform(1, 2, (17, 1)).astype(np.float32)
        ref = oracle.roi_feature_boost(x, s)
        assert np.array_equal(ops.roi_feature_boost(_t(x, dev), _t(s, dev)).cpu().numpy(), ref)
        assert np.array_equal(ops.roi_feature_boost_grad(_t(x, dev), _t(s, dev)).cpu().numpy(), ref)
    with pytest.raises(lib.NawsError):
        ops.roi_feature_boost(_t(x, dev), _t(np.ones((16, 1), np.float32), dev))


def test_roi_iou_bitexact(dev):
    from naws_hip import ops
    from oracle import oracle
    rng = np.random.default_rng(2)
    rois = make_rois(rng, 1, 300, 600, 1000)
    j = ops.roi_iou(_t(rois, dev)).cpu().numpy()
    assert np.array_equal(j, oracle.roi_iou(rois), equal_nan=True)


# ------------------------------------------------------------ WSDDN outputs --
def _logits(rng, rt, c):
    return [(rng.standard_normal((rt, c)) * s).astype(np.float32) for s in (2.0, 3.0, 0.5, 0.5)]


@pytest.mark.parametrize('c', [20, 80])
def test_wsddn_outputs_fwd_bwd(dev, c):
    from naws_hip import ops
    from oracle import oracle
    rng = np.random.default_rng(3)
    lens = [257, 64, 301]
    seg = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    rt = int(seg[-1])
    fc8c, fc8d, nc, nd = _logits(rng, rt, c)
    buf = _t(np.concatenate([fc8c, fc8d, nc, nd], 1), dev)      # [Rt, 4C] row-strided views
    v = [buf[:, i * c:(i + 1) * c] for i in range(4)]
    ac, ad, rp, cp = ops.wsddn_outputs(v[0], v[1], v[2], v[3], _t(seg, dev))
    g = rng.standard_normal((2, len(lens), c)).astype(np.float32)
    dl = ops.wsddn_outputs_grad(ac, ad, rp, cp, _t(g, dev), _t(seg, dev)).cpu().numpy()
    for s, (lo, hi) in enumerate(zip(seg[:-1], seg[1:])):
        sl = slice(lo, hi)
        r0 = oracle.wsddn_outputs(fc8c[sl], fc8d[sl])
        r1 = oracle.wsddn_outputs(fc8c[sl], fc8d[sl], nc[sl], nd[sl])
        for b, ref in enumerate((r0, r1)):
            _close(ac[b, sl], ref[0]); _close(ad[b, sl], ref[1], atol=1e-9)
            _close(rp[b, sl], ref[2], atol=1e-10); _close(cp[b, s], ref[3][0])
        dzc0, dzd0 = oracle.wsddn_outputs_grad(r0[0], r0[1], g[0, s])
        dzc1, dzd1 = oracle.wsddn_outputs_grad(r1[0], r1[1], g[1, s])
        scale = np.abs(g).max()
        _close(dl[sl, 0:c], dzc0 + dzc1, atol=2e-6 * scale)
        _close(dl[sl, c:2 * c], dzd0 + dzd1, atol=2e-6 * scale)
        _close(dl[sl, 2 * c:3 * c], dzc1, atol=2e-6 * scale)
        _close(dl[sl, 3 * c:4 * c], dzd1, atol=2e-6 * scale)


# ------------------------------------------------------------- entropy gate --
@pytest.mark.parametrize('c', [20, 27, 80, 93])     # one pass of 20 / 40 / 80 classes; two passes
def test_entropy_gate(dev, c):
    from naws_hip import ops
    from oracle import oracle
    rng = np.random.default_rng(4)
    rois = make_rois(rng, 2, 333, 600, 1000)
    seg = seg_offsets(rois)
    rt = rois.shape[0]
    fc8c, fc8d, _, _ = _logits(rng, rt, c)
    labels = np.zeros((2, c), np.float32)
    labels[0, 3] = 1; labels[1, 5] = 1; labels[1, 7] = 0.4   # mixup-style fractional label
    rp = np.empty((rt, c), np.float32); cp = np.empty((2, c), np.float32)
    for s in range(2):
        sl = slice(seg[s], seg[s + 1])
        _, _, rp[sl], cps = oracle.wsddn_outputs(fc8c[sl], fc8d[sl])
        cp[s] = cps[0]
    rp[5, 2] = 0.0           # p = 0 -> 0*log 0 = NaN -> ReplaceNaN -> 0
    outs = ops.entropy_gate(_t(rois, dev), _t(rp, dev), _t(cp, dev), _t(labels, dev),
                            _t(seg, dev), int(np.diff(seg).max()))
    for s in range(2):
        sl = slice(seg[s], seg[s + 1])
        ref = oracle.entropy_gate(rois[sl], rp[sl], cp[s], labels[s])
        for o, r in zip(outs, ref):
            _close(o[s], r[0], rtol=1e-4, atol=1e-7)


# ------------------------------------------------------------------ WCE ------
@pytest.mark.parametrize('weighted', [True, False])
@pytest.mark.parametrize('is_mean', [True, False])
def test_weighted_ce(dev, weighted, is_mean):
    from naws_hip import ops
    from oracle import oracle
    rng = np.random.default_rng(5)
    for c in (20, 80):
        x = rng.uniform(0, 1, (3, 1, c)).astype(np.float32)
        x[0, 0, :4] = [0.0, 1.0, 1e-30, 0.9999999]
        l = (rng.uniform(0, 1, (3, 1, c)) > 0.8).astype(np.float32)
        l[1, 0, 2] = 0.37
        w = rng.uniform(0, 1, (3, 1, c)).astype(np.float32) if weighted else None
        dy = np.array([1.0, 0.5, 2.0], np.float32)
        y = ops.weighted_ce(_t(x, dev), _t(l, dev), _t(w, dev) if weighted else None, is_mean, 3)
        dx = ops.weighted_ce_grad(_t(x, dev), _t(l, dev), _t(w, dev) if weighted else None,
                                  _t(dy, dev), is_mean, 3)
        for p in range(3):
            wp = w[p] if weighted else None
            _close(y[p], oracle.weighted_ce(x[p], l[p], wp, is_mean), rtol=1e-6)
            _close(dx[p], oracle.weighted_ce_grad(x[p], l[p], wp, dy[p:p + 1], is_mean), rtol=1e-6)


def test_wce_known_answer(dev):
    """The one reference output on record (reference .cc compiled by the survey)."""
    from naws_hip import ops
    x = np.array([[.9, .05, .3, 0.]], np.float32)
    l = np.array([[1, 0, .4, 0]], np.float32)
    w = np.array([[1, .5, 1, .25]], np.float32)
    y = ops.weighted_ce(_t(x, dev), _t(l, dev), _t(w, dev), True)
    assert abs(float(y[0]) - 0.206650317) < 1e-6


# ------------------------------------------------------------------ SGD ------
@pytest.mark.parametrize('iter_size,gpu_num,nesterov', [(1, 1, 0), (1, 8, 0), (2, 4, 0), (1, 2, 1)])
def test_acm_sgd(dev, iter_size, gpu_num, nesterov):
    from naws_hip import ops
    from oracle import oracle
    rng = np.random.default_rng(6)
    sizes = [4096 * 8, 4096, 20 * 4096, 20]           # weight, bias, weight, bias
    lr_mult = [1.0, 2.0, 1.0, 2.0]
    wd = [5e-4, 0.0, 5e-4, 0.0]
    total = sum(sizes)
    ends = np.cumsum(sizes).astype(np.int64)
    p = rng.standard_normal(total).astype(np.float32)
    m = rng.standard_normal(total).astype(np.float32)     # garbage: first call must zero it
    a = rng.standard_normal(total).astype(np.float32)
    lr = np.array([1e-3], np.float32)
    pd, md, ad = _t(p, dev), _t(m, dev), _t(a, dev)
    use_acm = iter_size != 1
    pr, mr, ar = p.copy(), m.copy(), a.copy()
    it_ref = [0] * len(sizes)
    for it in range(4):
        g = rng.standard_normal(total).astype(np.float32)
        gd = _t(g, dev)
        ops.acm_sgd_update(gd, md, _t(lr, dev), pd, ad if use_acm else None, _t(ends, dev),
                           _t(np.array(lr_mult, np.float32), dev), _t(np.array(wd, np.float32), dev),
                           0.9, nesterov, iter_size, gpu_num, it)
        assert np.array_equal(gd.cpu().numpy(), g)        # grad blob is read-only
        lo = 0
        for k, sz in enumerate(sizes):
            sl = slice(lo, lo + sz)
            gk, mk, pk, ak = g[sl].copy(), mr[sl].copy(), pr[sl].copy(), ar[sl].copy()
            it_ref[k] = oracle.acm_sgd(gk, mk, lr, pk, ak, 0.9, nesterov, wd[k], iter_size,
                                       gpu_num, lr_mult[k], it_ref[k])
            mr[sl], pr[sl], ar[sl] = mk, pk, ak
            lo += sz
        # every product and sum of the kernel is rounded on its own (no FMA contraction), as in
        # the oracle's -ffp-contract=off restatement of the scalar CPU operator: bit for bit
        assert np.array_equal(pd.cpu().numpy(), pr), it
        assert np.array_equal(md.cpu().numpy(), mr), it
        if use_acm:
            assert np.array_equal(ad.cpu().numpy(), ar), it


def _planes_to_dense(f16x2, rows_per_batch):
    """F16x2 planes [2, (b,) K/16, rows, 16] -> the fp32 matrix they represent, [b*rows, K]."""
    p = f16x2.planes.double()
    d = p[0] + p[1]
    if d.dim() == 3:
        d = d.unsqueeze(0)
    d = d.permute(0, 2, 1, 3).reshape(d.shape[0] * d.shape[2], -1)       # [b*rows, K]
    return d * f16x2.inv_scale.reshape(-1).double()[:, None]


@pytest.fixture
def sgd_walk(request):
    """knob "sgd_wgs": 0 = the launch rule, -3 = three resident workgroups walk all tiles."""
    from naws_hip import lib as L
    L.set_variant('sgd_wgs', request.param)
    yield request.param
    L.set_variant('sgd_wgs', 0)


@pytest.mark.parametrize('sgd_walk', [0, -3, -1], indirect=True)
@pytest.mark.parametrize('nesterov', [0, 1])
def test_acm_sgd_update_f16x2_writes_the_operand_planes(dev, nesterov, sgd_walk):
    """naws_acm_sgd_update_f16x2: parameters and momentum bit-identical to the
    plain fused update (which is held to the oracle in test_acm_sgd); rowmax = the exact row
    maxima of the updated weights; the planes it writes, times the 1/scale it reports, reproduce
    every updated weight to the documented 2^-22 relative / 2^-38 of twice the old row maximum;
    the bias / fc8 elements between and behind the matrices are updated too."""
    from naws_hip import ops
    rng = np.random.default_rng(8 + nesterov)
    # [W6-like 64 x 512 | bias 64 | W7-like 2 x (32 x 256) | bias 64 | tail 40]
    r6, c6, r7, c7 = 64, 512, 32, 256
    sizes = [r6 * c6, 64, 2 * r7 * c7, 64, 40]
    lr_mult = [1.0, 2.0, 1.0, 2.0, 1.0]
    wd = [5e-4, 0.0, 5e-4, 0.0, 5e-4]
    total = sum(sizes)
    o7 = sizes[0] + sizes[1]
    ends = _t(np.cumsum(sizes).astype(np.int64), dev)
    lm, wdd = _t(np.array(lr_mult, np.float32), dev), _t(np.array(wd, np.float32), dev)
    p0 = rng.standard_normal(total).astype(np.float32)
    p0[:r6 * c6].reshape(r6, c6)[5] *= 1e-6             # rows of very different magnitude
    p0[:r6 * c6].reshape(r6, c6)[9] *= 3e4
    lr = _t(np.array([1e-2], np.float32), dev)
    pa, pb = _t(p0, dev), _t(p0, dev)
    ma = _t(rng.standard_normal(total).astype(np.float32), dev)    # garbage: the first call zeroes it
    mb = ma.clone()
    w6 = pb[:r6 * c6].view(r6, c6)
    w7 = pb[o7:o7 + 2 * r7 * c7].view(2, r7, c7)
    q6, q7 = ops.split_f16x2(w6), ops.split_f16x2(w7)                     # initial planes + maxima
    bound = torch.zeros((r6 + 2 * r7,), device=dev, dtype=torch.int32)
    ovf = torch.zeros((1,), device=dev, dtype=torch.int32)
    mx6, mx7 = q6.scales[0].view(torch.int32), q7.scales[0].view(torch.int32).reshape(-1)
    cm7 = torch.zeros((2, c7), device=dev, dtype=torch.int32)      # fc7-like: column maxima per batch item
    regs = ops.SgdPlaneRegions([(0, r6, c6, r6, q6.planes, bound[:r6], mx6, q6.scales[1]),
                                (o7, 2 * r7, c7, r7, q7.planes, bound[r6:], mx7,
                                 q7.scales[1].reshape(-1), cm7)])
    rowscale = np.ones((total,), np.float32)            # gradients in proportion to their rows
    rowscale[:r6 * c6].reshape(r6, c6)[5] = 1e-6
    rowscale[:r6 * c6].reshape(r6, c6)[9] = 3e4
    for it in range(3):
        g = _t(rng.standard_normal(total).astype(np.float32) * rowscale, dev)
        ops.acm_sgd_update(g, ma, lr, pa, None, ends, lm, wdd, 0.9, nesterov, 1, 2, it)
        bound[:r6].copy_(mx6); bound[r6:].copy_(mx7)
        old = bound.clone()
        mx6.zero_(); mx7.zero_(); cm7.zero_()
        ops.acm_sgd_update_f16x2(g, mb, lr, pb, ends, lm, wdd, 0.9, nesterov, 2, it, regs, ovf, it + 1)
        assert torch.equal(pa, pb) and torch.equal(ma, mb), it
        assert torch.equal(cm7.view(torch.float32), w7.abs().amax(dim=1)), it
        assert int(ovf.item()) == 0
        for w, q, mx, b in ((w6, q6, mx6, old[:r6]), (w7, q7, mx7, old[r6:])):
            w2 = w.reshape(-1, w.shape[-1])
            assert torch.equal(mx.view(torch.float32), w2.abs().amax(dim=1)), it
            dense = _planes_to_dense(q, w.shape[-2])
            err = (dense - w2.double()).abs()
            floor = 2.0 * b.view(torch.float32).double() * 2.0 ** -38
            tol = torch.maximum(w2.double().abs() * 2.0 ** -22, floor[:, None])
            assert bool((err <= tol).all()), (it, float((err / tol).max()))
            # 1/scale is the power of two that goes with twice the old maximum
            # (exact powers of two: evaluated on the host, a device pow is not exact)
            e_old = np.floor(np.log2(b.view(torch.float32).cpu().numpy().astype(np.float64)))
            want = np.ldexp(1.0, (e_old + 1 - 14).astype(np.int64))
            assert np.array_equal(q.inv_scale.reshape(-1).cpu().numpy().astype(np.float64), want), it
    # ---- a row that outgrows twice its old maximum raises the flag, and the conditional re-split
    # then leaves exactly the planes of a from-scratch split
    g = torch.zeros((total,), device=dev)
    g[3 * c6:4 * c6] = -5e3                     # row 3 of W6 jumps by lr * 5e3 / gpu_num = 25
    bound[:r6].copy_(mx6); bound[r6:].copy_(mx7)
    mx6.zero_(); mx7.zero_()
    ops.acm_sgd_update_f16x2(g, mb, lr, pb, ends, lm, wdd, 0.9, nesterov, 2, 3, regs, ovf, 77)
    assert int(ovf.item()) == 77
    ops.split_f16x2_rows_if(w6, mx6, q6, ovf, 76)                         # wrong tag: nothing happens
    stale = q6.planes.clone()
    torch.cuda.synchronize()
    assert torch.equal(stale, q6.planes)
    ops.split_f16x2_rows_if(w6, mx6, q6, ovf, 77)
    ops.split_f16x2_rows_if(w7, mx7, q7, ovf, 77)
    for w, q in ((w6, q6), (w7, q7)):
        fresh = ops.split_f16x2(w)
        assert torch.equal(q.inv_scale, fresh.inv_scale)
        assert torch.equal(q.planes.view(torch.int16), fresh.planes.view(torch.int16))
    # argument checks (the ENFORCE-style codes)
    from naws_hip import lib
    odd = torch.empty((2, 20, r6, 16), device=dev, dtype=torch.float16)       # 320 columns: not x 256
    bad = ops.SgdPlaneRegions([(0, r6, 320, r6, odd, bound[:r6], mx6, q6.scales[1])])
    with pytest.raises(lib.NawsError):
        ops.acm_sgd_update_f16x2(g, mb, lr, pb, ends, lm, wdd, 0.9, 0, 2, 4, bad, ovf, 1)


def test_stat(dev):
    from naws_hip import ops
    from oracle import oracle
    rng = np.random.default_rng(7)
    ai = np.full(20, 7.0, np.float32); al = np.full(20, 3.0, np.float32)
    aid, ald = _t(ai, dev), _t(al, dev)
    for it in range(3):
        i = rng.uniform(0, 1, 20).astype(np.float32)
        l = (rng.uniform(0, 1, 20) > 0.5).astype(np.float32)
        ops.stat_accumulate(_t(i, dev), _t(l, dev), aid, ald, it == 0)
        oracle.stat(i, l, ai, al, it == 0)
    assert np.array_equal(aid.cpu().numpy(), ai) and np.array_equal(ald.cpu().numpy(), al)


# ------------------------------------------------------------------ GEMM -----
def _ref_mm(a, b):
    return (a.astype(np.float64) @ b.astype(np.float64))


@pytest.mark.parametrize('ta,tb', [(0, 1), (0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize('m,n,k', [(300, 200, 96), (128, 128, 32), (1000, 40, 4096),
                                   (520, 8192, 64), (4000, 4096, 100), (40, 4096, 1000)])
def test_gemm_layouts(dev, ta, tb, m, n, k):
    from naws_hip import ops
    rng = np.random.default_rng(8)
    # operand dims that end up as vector-load (contiguous) dims must be multiples of 4
    a = rng.uniform(-1, 1, (k, m) if ta else (m, k)).astype(np.float32)
    b = rng.uniform(-1, 1, (n, k) if tb else (k, n)).astype(np.float32)
    ref = _ref_mm(a.T if ta else a, b.T if tb else b)
    c = ops.gemm(_t(a, dev), _t(b, dev), bool(ta), bool(tb)).cpu().numpy()
    err = np.abs(c - ref).max() / (np.abs(ref).max() + 1e-30)
    assert err < 5e-6, err


def test_gemm_asymmetric_identity(dev):
    """A = I with an asymmetric B catches a transposed C/D register map."""
    from naws_hip import ops
    n = 256
    b = np.arange(n * n, dtype=np.float32).reshape(n, n) % 1013
    c = ops.gemm(_t(np.eye(n, dtype=np.float32), dev), _t(b, dev)).cpu().numpy()
    assert np.array_equal(c, b)


def test_gemm_epilogues(dev):
    from naws_hip import ops, lib
    rng = np.random.default_rng(9)
    m, n, k = 260, 384, 160
    a = rng.uniform(-1, 1, (m, k)).astype(np.float32)
    w = rng.uniform(-1, 1, (n, k)).astype(np.float32)
    bias = rng.uniform(-1, 1, n).astype(np.float32)
    z = _ref_mm(a, w.T) + bias
    ad, wd, bd = _t(a, dev), _t(w, dev), _t(bias, dev)
    _close(ops.gemm(ad, wd, False, True, epilogue=lib.EPI_BIAS, bias=bd), z, atol=1e-4)
    _close(ops.gemm(ad, wd, False, True, epilogue=lib.EPI_BIAS_RELU, bias=bd), np.maximum(z, 0),
           atol=1e-4)
    y = ops.gemm(ad, wd, False, True, epilogue=lib.EPI_BIAS_RELU_DROP, bias=bd, drop_ratio=0.5,
                 seed=1234)
    mask = ops.dropout_mask(1234, 0.5, m * n, dev).reshape(m, n).cpu().numpy()
    assert 0.45 < mask.mean() < 0.55
    _close(y, np.maximum(z, 0) * mask * 2.0, atol=2e-4)
    # gate epilogue (FC dgrad through ReLU + Dropout) and accumulate
    aux = rng.standard_normal((m, n)).astype(np.float32)
    g = ops.gemm(ad, wd, False, True, epilogue=lib.EPI_GATE_POS, aux=_t(aux, dev), alpha=2.0)
    _close(g, np.where(aux > 0, _ref_mm(a, w.T) * 2.0, 0.0), atol=2e-4)
    c0 = rng.standard_normal((m, n)).astype(np.float32)
    cd = _t(c0, dev)
    ops.gemm(ad, wd, False, True, out=cd, accumulate=True)
    _close(cd, c0 + _ref_mm(a, w.T), atol=2e-4)


def test_gemm_batched_and_errors(dev):
    from naws_hip import ops, lib
    rng = np.random.default_rng(10)
    a = rng.uniform(-1, 1, (2, 130, 64)).astype(np.float32)
    w = rng.uniform(-1, 1, (2, 72, 64)).astype(np.float32)
    b = rng.uniform(-1, 1, (2, 72)).astype(np.float32)
    y = ops.gemm(_t(a, dev), _t(w, dev), False, True, epilogue=lib.EPI_BIAS, bias=_t(b, dev))
    for i in range(2):
        _close(y[i], _ref_mm(a[i], w[i].T) + b[i], atol=1e-4)
    with pytest.raises(lib.NawsError):          # K not a multiple of 4
        ops.gemm(torch.zeros((8, 6), device=dev), torch.zeros((8, 6), device=dev), False, True)
    with pytest.raises(lib.NawsError):          # inner dims differ
        ops.gemm(torch.zeros((8, 8), device=dev), torch.zeros((8, 12), device=dev), False, True)


def test_colsum(dev):
    from naws_hip import ops
    rng = np.random.default_rng(12)
    x = rng.standard_normal((1000, 200)).astype(np.float32)
    _close(ops.colsum(_t(x, dev)), x.astype(np.float64).sum(0), rtol=1e-5, atol=1e-4)


# ------------------------------------------------------------------ conv -----
@pytest.mark.parametrize('cin,cout,dil,h,w', [(64, 64, 1, 37, 53), (128, 256, 1, 19, 23),
                                              (512, 512, 2, 20, 31), (64, 128, 1, 75, 125)])
def test_conv3x3_nhwc(dev, cin, cout, dil, h, w):
    from naws_hip import ops
    import torch.nn.functional as F
    rng = np.random.default_rng(13)
    n = 2
    x = rng.uniform(-1, 1, (n, cin, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, cout).astype(np.float32)
    ref = F.relu(F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(),
                          torch.from_numpy(b).double(), padding=dil, dilation=dil)).numpy()
    xd = ops.nchw_to_nhwc(_t(x, dev))
    wp = ops.conv3x3_pack_weight(_t(wt, dev))
    y = ops.nhwc_to_nchw(ops.conv3x3_nhwc(xd, wp, _t(b, dev), dil, True)).cpu().numpy()
    assert np.abs(y - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize('cin,cout,dil,h,w', [(256, 256, 1, 37, 53), (128, 256, 1, 19, 24),
                                              (512, 512, 2, 21, 31), (512, 512, 2, 74, 124)])
def test_conv3x3_winograd(dev, cin, cout, dil, h, w):
    from naws_hip import ops
    import torch.nn.functional as F
    rng = np.random.default_rng(16)
    n = 2
    x = rng.uniform(-1, 1, (n, cin, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, cout).astype(np.float32)
    ref = F.relu(F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(),
                          torch.from_numpy(b).double(), padding=dil, dilation=dil)).numpy()
    xd = ops.nchw_to_nhwc(_t(x, dev))
    u = ops.winograd_weight_transform(_t(wt, dev))
    y = ops.nhwc_to_nchw(ops.conv3x3_winograd_nhwc(xd, u, _t(b, dev), dil, True)).cpu().numpy()
    assert np.abs(y - ref).max() < 1e-5 * max(1.0, np.abs(ref).max())
    # and against the direct implicit-GEMM kernel
    yd = ops.nhwc_to_nchw(ops.conv3x3_nhwc(xd, ops.conv3x3_pack_weight(_t(wt, dev)), _t(b, dev),
                                           dil, True)).cpu().numpy()
    assert np.abs(y - yd).max() < 1e-5 * max(1.0, np.abs(ref).max())


def test_conv1_1_and_pool(dev):
    from naws_hip import ops
    import torch.nn.functional as F
    rng = np.random.default_rng(14)
    x = rng.uniform(-120, 140, (2, 3, 41, 67)).astype(np.float32)
    wt = (rng.standard_normal((64, 3, 3, 3)) * 0.1).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, 64).astype(np.float32)
    ref = F.relu(F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(),
                          torch.from_numpy(b).double(), padding=1))
    y = ops.conv3x3_c3_nchw_to_nhwc(_t(x, dev), _t(wt, dev), _t(b, dev), True)
    _close(ops.nhwc_to_nchw(y), ref.numpy(), rtol=1e-5, atol=1e-3)
    # no bias / no ReLU, a tile-aligned shape, and the generic (Cout != 64) kernel
    x2 = rng.uniform(-1, 1, (1, 3, 16, 128)).astype(np.float32)
    for cout in (64, 32):
        w2 = (rng.standard_normal((cout, 3, 3, 3)) * 0.1).astype(np.float32)
        r2 = F.conv2d(torch.from_numpy(x2).double(), torch.from_numpy(w2).double(), None, padding=1)
        y2 = ops.conv3x3_c3_nchw_to_nhwc(_t(x2, dev), _t(w2, dev), None, False)
        _close(ops.nhwc_to_nchw(y2), r2.numpy(), rtol=1e-5, atol=1e-5)
    for stride in (2, 1):
        pr = F.max_pool2d(ref, 2, stride, 0, ceil_mode=False).float().numpy()
        yp = ops.nhwc_to_nchw(ops.maxpool2x2_nhwc(y, stride))
        assert yp.shape == pr.shape
        _close(yp, pr, rtol=1e-5, atol=1e-3)


# ------------------------------------------------------------ small built-ins
def test_small_builtins(dev):
    from naws_hip import ops, lib
    rng = np.random.default_rng(15)
    x = rng.uniform(0.01, 2, (37, 20)).astype(np.float32)
    xd = _t(x, dev)
    _close(ops.unary(lib.UN_LOG, xd), np.log(x), rtol=1e-6)
    _close(ops.unary(lib.UN_SCALE, xd, -1.0), -x)
    xn = x.copy(); xn[3, 4] = np.nan
    _close(ops.unary(lib.UN_REPLACE_NAN, _t(xn, dev), 0.0), np.nan_to_num(xn, nan=0.0))
    _close(ops.unary(lib.UN_LEAKY_RELU, _t(x - 1, dev), 0.01), np.where(x - 1 >= 0, x - 1, 0.01 * (x - 1)))
    _close(ops.unary(lib.UN_CLIP, _t(xn, dev), 0.0, 1.0), np.clip(xn, 0, 1))   # NaN stays NaN
    row = rng.uniform(1, 2, (1, 20)).astype(np.float32)
    sc = np.array([[7.0]], np.float32)
    _close(ops.binary(lib.BIN_DIV, xd, _t(row, dev)), x / row)
    _close(ops.binary(lib.BIN_SUB, _t(sc, dev), _t(row, dev)), sc - row)
    _close(ops.binary(lib.BIN_MUL, _t(row, dev), xd), row * x)
    _close(ops.binary(lib.BIN_GATE_POS, xd, _t(x - 1, dev)), np.where(x - 1 > 0, x, 0))
    e = np.exp(x - x.max(1, keepdims=True)); sm = e / e.sum(1, keepdims=True)
    y = ops.softmax_rows(xd)
    _close(y, sm, rtol=1e-5)
    dy = rng.standard_normal(x.shape).astype(np.float32)
    _close(ops.softmax_rows_grad(y, _t(dy, dev)), sm * (dy - (sm * dy).sum(1, keepdims=True)),
           rtol=1e-4, atol=1e-7)
    assert np.array_equal(ops.transpose2d(xd).cpu().numpy(), x.T)
    _close(ops.reduce_sum_axis0(xd), x.sum(0, keepdims=True), rtol=1e-5)


@pytest.mark.parametrize('ta,tb,m,n,k,batch,ksplit', [
    (False, True, 4000, 40, 4096, 2, 4),      # fc8 forward: logits = H7 W8^T + b
    (True, False, 40, 4096, 4000, 2, 4),      # fc8 wgrad: dW8 = dL^T H7
    (False, True, 70, 44, 100, 1, 8),         # more slices than 32-deep K-steps: clamped
    (False, False, 130, 36, 64, 3, 1),        # one slice: the same two passes
])
def test_gemm_splitk_matches_the_one_pass_gemm(dev, ta, tb, m, n, k, batch, ksplit):
    """naws_gemm_f32_splitk: K cut into slices, partial products summed in slice order by a second
    pass (fc8's small outputs): equal to naws_gemm_f32 up to the order of the fp32 sums, and
    bit-identical from run to run."""
    from naws_hip import ops, lib as L
    g = torch.Generator(device=dev).manual_seed(k + n)
    a = torch.randn((batch, k, m) if ta else (batch, m, k), device=dev, generator=g)
    b = torch.randn((batch, n, k) if tb else (batch, k, n), device=dev, generator=g) * 0.05
    bias = torch.randn((batch, n), device=dev, generator=g)
    for epi, bb in ((L.EPI_NONE, None), (L.EPI_BIAS, bias)):
        want = ops.gemm(a, b, ta, tb, epilogue=epi, bias=bb)
        got = ops.gemm_splitk(a, b, ta, tb, epilogue=epi, bias=bb, ksplit=ksplit)
        again = ops.gemm_splitk(a, b, ta, tb, epilogue=epi, bias=bb, ksplit=ksplit)
        ref = (a.double().transpose(1, 2) if ta else a.double()) @ \
            (b.double().transpose(1, 2) if tb else b.double())
        if bb is not None:
            ref = ref + bb.double()[:, None, :]
        scale = float(ref.abs().max())
        assert float((got.double() - ref).abs().max()) <= 2e-6 * scale
        assert float((got - want).abs().max()) <= 4e-6 * scale     # two fp32 results: both errors
        assert torch.equal(got, again)
    # strided output rows (the engine writes the two branches' logits side by side)
    wide = torch.zeros((m, batch, n + 4), device=dev)
    outv = wide.permute(1, 0, 2)[:, :, :n]
    ops.gemm_splitk(a, b, ta, tb, out=outv, ksplit=ksplit)
    assert torch.equal(outv, ops.gemm_splitk(a, b, ta, tb, ksplit=ksplit))
    assert not wide[:, :, n:].any()
    with pytest.raises(L.NawsError):
        ops.gemm_splitk(a, b, ta, tb, epilogue=L.EPI_BIAS_RELU, bias=bias, ksplit=ksplit)


# --------- head kernels against the imported reference's graph, executed in numpy ----------
@pytest.mark.parametrize('idx', range(5))
def test_head_kernels_match_the_reference_graph_numeric(dev, idx):
    """ops.wsddn_outputs (incl. cls_pred) and ops.entropy_gate against tests/golden/
    reference_graph_numeric.npz: the reference's add_webly_outputs / add_cls_pred /
    add_spatial_entropy_weight / add_webly_losses run op by op in numpy fp32 (R in {64, 300},
    C in {20, 80}, p = 0 entries; case 4: a D = 0 column -> NaN, under the fixture's stated Clip
    reading).  Same bounds as the oracle's own test of this fixture (test_graph_numeric_golden.py)."""
    from naws_hip import ops
    from test_graph_numeric_golden import CASES, close, logits_of
    info, case = CASES[idx]
    fc8c, fc8d, nc, nd = logits_of(case)
    r = fc8c.shape[0]
    seg = _t(np.array([0, r], np.int32), dev)
    ac, ad, rp, cp = ops.wsddn_outputs(_t(fc8c, dev), _t(fc8d, dev), _t(nc, dev), _t(nd, dev), seg)
    if 'out_alpha_cls' in case:
        close(ac[0].cpu().numpy(), case['out_alpha_cls'], 2e-6)
        close(ad[0].cpu().numpy(), case['out_alpha_det'], 2e-6, 1e-44)
        close(ac[1].cpu().numpy(), case['out_alpha_cls_noise'], 2e-6)
        close(ad[1].cpu().numpy(), case['out_alpha_det_noise'], 2e-6, 1e-44)
    close(rp[0].cpu().numpy(), case['out_rois_pred'], 1e-5, 1e-44)
    close(rp[1].cpu().numpy(), case['out_rois_pred_noise'], 1e-5, 1e-44)
    close(cp[0].cpu().numpy(), case['out_cls_prob'], 1e-5)
    close(cp[1].cpu().numpy(), case['out_cls_prob_noise'], 1e-5)
    assert (rp[0].cpu().numpy()[info['p0_rows'], info['p0_class']] == 0).all()
    outs = ops.entropy_gate(_t(case['in_rois'], dev), _t(case['out_rois_pred'], dev),
                            _t(case['out_cls_prob'], dev), _t(case['in_labels_oh'], dev), seg, r)
    for o, name in zip(outs, ('rois_class_weight', 'rois_class_weight_noise', 'rois_pred_hatE_sum',
                              'rois_pred_hatE_sum_norm')):
        close(o.cpu().numpy().reshape(1, -1), case['out_' + name], 2e-5)


@pytest.mark.parametrize('is_mean', [True, False])
def test_weighted_ce_shared_labels(dev, is_mean):
    """naws_weighted_ce_shared_fwd / _bwd (2 branches x nseg images scored against ONE labels_oh per
    image; gradient seeded by a constant instead of a tensor of ones) == the per-problem operator
    on expanded labels, bit for bit, and == the oracle (cross_entropy_wsl_op.cc:87-180)."""
    from naws_hip import ops
    from oracle import oracle
    rng = np.random.default_rng(15)
    for c, nseg in ((20, 3), (80, 2)):
        x = rng.uniform(0, 1, (2, nseg, c)).astype(np.float32)
        x[0, 0, :4] = [0.0, 1.0, 1e-30, 0.9999999]
        l = (rng.uniform(0, 1, (nseg, c)) > 0.8).astype(np.float32)
        l[0, 2] = 0.37
        w = rng.uniform(0, 1, (2, nseg, c)).astype(np.float32)
        xd, ld, wd = _t(x, dev), _t(l, dev), _t(w, dev)
        y = ops.weighted_ce_shared(xd, ld, wd, is_mean)
        dx = ops.weighted_ce_shared_grad(xd, ld, wd, is_mean, dy_const=1.0)
        l2 = ld.unsqueeze(0).expand(2, nseg, c).contiguous()
        ones = torch.ones((2 * nseg,), device=dev)
        assert torch.equal(y, ops.weighted_ce(xd, l2, wd, is_mean, 2 * nseg))
        assert torch.equal(dx, ops.weighted_ce_grad(xd, l2, wd, ones, is_mean, 2 * nseg))
        dy = _t(rng.uniform(0.5, 2, 2 * nseg).astype(np.float32), dev)
        assert torch.equal(ops.weighted_ce_shared_grad(xd, ld, wd, is_mean, dy=dy),
                           ops.weighted_ce_grad(xd, l2, wd, dy, is_mean, 2 * nseg))
        for b in range(2):
            for s in range(nseg):
                _close(y[b * nseg + s], oracle.weighted_ce(x[b, s:s + 1], l[s:s + 1], w[b, s:s + 1], is_mean),
                       rtol=1e-6)
                _close(dx[b, s], oracle.weighted_ce_grad(x[b, s:s + 1], l[s:s + 1], w[b, s:s + 1],
                                                         np.ones((1,), np.float32), is_mean)[0], rtol=1e-6)
