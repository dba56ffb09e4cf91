"""GPU parity for the bf16 MFMA option (BASELINE.json configs[3]).

The kernels round their fp32 operands to bf16 (round-to-nearest-even) and accumulate in
fp32, so against a float64 product of the *same rounded operands* they must agree to fp32
accumulation error (tight test); against the unrounded fp32 oracle result they agree to
bf16 operand precision (2^-8 per operand, random-sign error ~ 2^-8 * |a||b| / sqrt(K)).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _bf(a):
    """fp32 ndarray -> bf16-rounded values as float64 (torch's RNE conversion on the CPU)."""
    return torch.from_numpy(a).bfloat16().double().numpy()


@pytest.mark.parametrize('m,n,k', [(128, 128, 32), (130, 72, 64), (517, 260, 1000), (64, 4000, 264),
                                   (2100, 140, 96)])
@pytest.mark.parametrize('a16,b16', [(False, False), (True, False), (False, True), (True, True)])
def test_gemm_bf16_nt(dev, m, n, k, a16, b16):
    from naws_hip import ops
    rng = np.random.default_rng(21)
    a = rng.uniform(-1, 1, (m, k)).astype(np.float32)
    b = rng.uniform(-1, 1, (n, k)).astype(np.float32)
    ref = _bf(a) @ _bf(b).T
    ad, bd = _t(a, dev), _t(b, dev)
    if a16:
        ad = ad.bfloat16()
    if b16:
        bd = bd.bfloat16()
    c = ops.gemm_bf16_nt(ad, bd).cpu().numpy()
    err = np.abs(c - ref).max() / np.abs(ref).max()
    assert err < 5e-6, err
    # and against the unrounded product: bf16 operand precision
    full = a.astype(np.float64) @ b.astype(np.float64).T
    assert np.abs(c - full).max() / np.abs(full).max() < 2e-2


def test_gemm_bf16_identity_and_map(dev):
    """A = I with an asymmetric B (small integers are exact in bf16) catches a wrong
    operand lane map or a transposed C/D register map."""
    from naws_hip import ops
    n = 256
    b = (np.arange(n * n, dtype=np.float32).reshape(n, n) % 251)
    c = ops.gemm_bf16_nt(_t(np.eye(n, dtype=np.float32), dev), _t(b.T.copy(), dev)).cpu().numpy()
    assert np.array_equal(c, b)


def test_gemm_bf16_epilogues_batched(dev):
    from naws_hip import ops, lib
    rng = np.random.default_rng(22)
    m, n, k = 260, 384, 160
    a = rng.uniform(-1, 1, (2, m, k)).astype(np.float32)
    w = rng.uniform(-1, 1, (2, n, k)).astype(np.float32)
    bias = rng.uniform(-1, 1, (2, n)).astype(np.float32)
    z = np.stack([_bf(a[i]) @ _bf(w[i]).T + bias[i] for i in range(2)])
    ad, wd, bd = _t(a, dev), _t(w, dev), _t(bias, dev)
    y = ops.gemm_bf16_nt(ad, wd, epilogue=lib.EPI_BIAS, bias=bd).cpu().numpy()
    np.testing.assert_allclose(y, z, rtol=1e-5, atol=1e-4)
    y = ops.gemm_bf16_nt(ad, wd, epilogue=lib.EPI_BIAS_RELU, bias=bd).cpu().numpy()
    np.testing.assert_allclose(y, np.maximum(z, 0), rtol=1e-5, atol=1e-4)
    y = ops.gemm_bf16_nt(ad, wd, epilogue=lib.EPI_BIAS_RELU_DROP, bias=bd, drop_ratio=0.5,
                         seed=77).cpu().numpy()
    mask = ops.dropout_mask(77, 0.5, 2 * m * n, dev).reshape(2, m, n).cpu().numpy()
    np.testing.assert_allclose(y, np.maximum(z, 0) * mask * 2.0, rtol=1e-5, atol=2e-4)
    aux = rng.standard_normal((2, m, n)).astype(np.float32)
    g = ops.gemm_bf16_nt(ad, wd, epilogue=lib.EPI_GATE_POS, aux=_t(aux, dev), alpha=2.0)
    zz = np.stack([_bf(a[i]) @ _bf(w[i]).T for i in range(2)])
    np.testing.assert_allclose(g.cpu().numpy(), np.where(aux > 0, zz * 2.0, 0.0), rtol=1e-5,
                               atol=2e-4)
    c0 = rng.standard_normal((2, m, n)).astype(np.float32)
    cd = _t(c0, dev)
    ops.gemm_bf16_nt(ad, wd, out=cd, accumulate=True)
    np.testing.assert_allclose(cd.cpu().numpy(), c0 + zz, rtol=1e-5, atol=2e-4)
    with pytest.raises(lib.NawsError):      # K not a multiple of 8
        ops.gemm_bf16_nt(torch.zeros((8, 12), device=dev), torch.zeros((8, 12), device=dev))


def test_transpose_to_bf16(dev):
    from naws_hip import ops
    rng = np.random.default_rng(23)
    x = rng.standard_normal((2, 101, 70)).astype(np.float32)
    y = ops.transpose_to_bf16(_t(x, dev), rows_pad=104)
    assert y.shape == (2, 70, 104) and y.dtype == torch.bfloat16
    ref = torch.from_numpy(x).bfloat16().transpose(1, 2)
    assert torch.equal(y[:, :, :101].cpu(), ref)
    assert (y[:, :, 101:] == 0).all()
    # strided (column-slice) source
    xs = _t(x, dev)[0][:, 8:40]
    ys = ops.transpose_to_bf16(xs)
    assert torch.equal(ys[:, :101].cpu(), torch.from_numpy(x[0][:, 8:40]).bfloat16().t())


@pytest.mark.parametrize('cin,cout,dil,h,w', [(64, 64, 1, 37, 53), (128, 256, 1, 19, 23),
                                              (512, 512, 2, 20, 31)])
def test_conv3x3_bf16(dev, cin, cout, dil, h, w):
    from naws_hip import ops
    import torch.nn.functional as F
    rng = np.random.default_rng(24)
    n = 2
    x = rng.uniform(-1, 1, (n, cin, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, cout).astype(np.float32)
    ref = F.relu(F.conv2d(torch.from_numpy(x).bfloat16().double(),
                          torch.from_numpy(wt).bfloat16().double(),
                          torch.from_numpy(b).double(), padding=dil, dilation=dil)).numpy()
    xd = ops.nchw_to_nhwc(_t(x, dev))
    wp = ops.conv3x3_pack_weight(_t(wt, dev))
    y = ops.nhwc_to_nchw(ops.conv3x3_nhwc_bf16(xd, wp, _t(b, dev), dil, True)).cpu().numpy()
    assert np.abs(y - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize('cin,cout,h,w,dil,pool', [
    (64, 64, 37, 70, 1, True),         # conv1_2 + pool1: ragged tiles, odd height (last row dropped)
    (64, 128, 16, 33, 1, False),       # conv2_1
    (128, 256, 24, 40, 1, True),       # four channel tiles of 64
    (256, 512, 19, 45, 1, False),      # 1 x 4 waves, 128-channel tiles
    (512, 512, 21, 35, 2, False),      # conv5_x: dilation 2
    (128, 192, 9, 31, 2, False),       # Cout % 128 != 0: 64-channel tiles, dilation 2
])
def test_conv3x3_bf16_wave_private(dev, cin, cout, h, w, dil, pool):
    """The bf16 plan's conv body kernel (`naws_conv3x3_nhwc_bf16_wp_fwd`, reference op: Conv + Relu
    (+ MaxPool) of VGG16.py:10-48): operands rounded to bf16 (nearest-even), fp32 accumulation -
    against float64 on the rounded operands; the fused pool against pooling the unfused output
    (bit-identical: max of the same fp32 values); both channel-tile forms agree bit for bit."""
    from naws_hip import ops, lib as L
    import torch.nn.functional as F
    rng = np.random.default_rng(cin + cout + h)
    n = 2
    x = rng.uniform(-1, 1, (n, cin, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, cout).astype(np.float32)
    ref = F.relu(F.conv2d(torch.from_numpy(x).bfloat16().double(),
                          torch.from_numpy(wt).bfloat16().double(),
                          torch.from_numpy(b).double(), padding=dil, dilation=dil))
    xd = ops.nchw_to_nhwc(_t(x, dev))
    ws = ops.to_bf16_slab(ops.conv3x3_pack_weight(_t(wt, dev)).view(cout, -1))
    assert ws.shape == (9 * cin // 16, cout, 16)
    y = ops.conv3x3_nhwc_bf16_wp(xd, ws, _t(b, dev), dil, True)
    got = ops.nhwc_to_nchw(y).cpu().double()
    assert float((got - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    if pool:
        yp = ops.conv3x3_nhwc_bf16_wp(xd, ws, _t(b, dev), dil, True, pool2=True)
        assert yp.shape == (n, h // 2, w // 2, cout)
        assert torch.equal(yp, ops.maxpool2x2_nhwc(y, 2))
    if cout % 128 == 0:
        try:
            outs = []
            for bn in (64, 128):
                L.set_variant('conv_bn', bn)
                outs.append(ops.conv3x3_nhwc_bf16_wp(xd, ws, _t(b, dev), dil, True))
            assert torch.equal(outs[0], outs[1])
        finally:
            L.set_variant('conv_bn', 0)
    # no-ReLU / no-bias form
    y0 = ops.conv3x3_nhwc_bf16_wp(xd, ws, None, dil, False)
    ref0 = F.conv2d(torch.from_numpy(x).bfloat16().double(), torch.from_numpy(wt).bfloat16().double(),
                    None, padding=dil, dilation=dil)
    assert float((ops.nhwc_to_nchw(y0).cpu().double() - ref0).abs().max()) < 2e-5 * max(1.0, float(ref0.abs().max()))


@pytest.mark.parametrize('m,n,k', [(256, 256, 64), (130, 72, 64), (517, 260, 1000), (64, 4000, 264),
                                   (2100, 140, 96)])
def test_gemm_bf16_slab(dev, m, n, k):
    """The bf16 plan's FC GEMM on the LDS-DMA pipeline (one-plane form of gemm_x3_kernel)."""
    from naws_hip import ops
    rng = np.random.default_rng(25)
    a = rng.uniform(-1, 1, (m, k)).astype(np.float32)
    b = rng.uniform(-1, 1, (n, k)).astype(np.float32)
    ref = _bf(a) @ _bf(b).T
    a1, b1 = ops.to_bf16_slab(_t(a, dev)), ops.to_bf16_slab(_t(b, dev))
    kp = (k + 63) // 64 * 64
    assert a1.shape == (kp // 16, m, 16) and a1.dtype == torch.bfloat16
    # layout: [k/16][row][k%16], zero-filled K pad
    dense = a1.float().permute(1, 0, 2).reshape(m, kp).cpu().numpy()
    assert np.array_equal(dense[:, :k], _bf(a).astype(np.float32)) and not dense[:, k:].any()
    c = ops.gemm_bf16_slab_nt(a1, b1).cpu().numpy()
    assert np.abs(c - ref).max() / np.abs(ref).max() < 5e-6
    # transposing conversion: dW = dY^T X with K = rows
    r = 203
    dy = rng.uniform(-1, 1, (r, 96)).astype(np.float32)
    x = rng.uniform(-1, 1, (r, n)).astype(np.float32)
    c2 = ops.gemm_bf16_slab_nt(ops.to_bf16_slab(_t(dy, dev), transpose=True),
                               ops.to_bf16_slab(_t(x, dev), transpose=True)).cpu().numpy()
    ref2 = _bf(dy).T @ _bf(x)
    assert np.abs(c2 - ref2).max() / np.abs(ref2).max() < 5e-6


def test_gemm_bf16_slab_epilogues_batched(dev):
    from naws_hip import ops, lib
    rng = np.random.default_rng(26)
    m, n, k = 260, 384, 160
    a = rng.uniform(-1, 1, (2, m, k)).astype(np.float32)
    w = rng.uniform(-1, 1, (2, n, k)).astype(np.float32)
    bias = rng.uniform(-1, 1, (2, n)).astype(np.float32)
    zz = np.stack([_bf(a[i]) @ _bf(w[i]).T for i in range(2)])
    z = zz + bias[:, None, :]
    a1, w1, bd = ops.to_bf16_slab(_t(a, dev)), ops.to_bf16_slab(_t(w, dev)), _t(bias, dev)
    y = ops.gemm_bf16_slab_nt(a1, w1, epilogue=lib.EPI_BIAS_RELU_DROP, bias=bd, drop_ratio=0.5,
                              seed=77).cpu().numpy()
    mask = ops.dropout_mask(77, 0.5, 2 * m * n, dev).reshape(2, m, n).cpu().numpy()
    np.testing.assert_allclose(y, np.maximum(z, 0) * mask * 2.0, rtol=1e-5, atol=2e-4)
    aux = rng.standard_normal((2, m, n)).astype(np.float32)
    g = ops.gemm_bf16_slab_nt(a1, w1, epilogue=lib.EPI_GATE_POS, aux=_t(aux, dev), alpha=2.0)
    np.testing.assert_allclose(g.cpu().numpy(), np.where(aux > 0, zz * 2.0, 0.0), rtol=1e-5, atol=2e-4)
    out = torch.zeros((2, m, n), device=dev)
    ops.gemm_bf16_slab_nt(a1[:, :, :100], w1, out=out[:, :100])          # row-sliced operand
    np.testing.assert_allclose(out[:, :100].cpu().numpy(), zz[:, :100], rtol=1e-5, atol=2e-4)


@pytest.mark.parametrize('m,n,r,it,blocks', [(256, 512, 96, 3, False),      # one tile row
                                             (4096, 4096, 160, 0, False),    # first iteration
                                             (512, 768, 70, 2, True)])       # ragged K, two row blocks
def test_bf16_wgrad_with_the_update_in_its_epilogue_equals_gemm_then_sgd(dev, m, n, r, it, blocks):
    """naws_gemm_bf16_slab_nt_sgd (the bf16 plan's fc6_w gradient at one process) against the two
    kernels it replaces - naws_gemm_bf16_slab_nt, then naws_acm_sgd_update_planes (NAWS_PLANES_BF16):
    parameters, momentum and the rounded operand plane bit for bit (reference: FCGradient +
    detectron/ops/acm_weightdecay_momentum_sgd_op.h:72-109)."""
    from naws_hip import ops, lib as L
    g = torch.Generator(device=dev).manual_seed(5 + m)
    dy = torch.randn((r, m), device=dev, generator=g) * 1e-3
    x = torch.randn((r, n), device=dev, generator=g).relu_()
    w = torch.randn((m, n), device=dev, generator=g) * 0.02
    mom = torch.randn((m, n), device=dev, generator=g) * 1e-3
    a, b = ops.to_bf16_slab(dy, transpose=True), ops.to_bf16_slab(x, transpose=True)
    lr = torch.tensor([3e-3], device=dev)
    lr_mult, wd, momentum, gpu_num = 1.0, 5e-4, 0.9, 4
    # gradient to memory, then the plane-writing SGD kernel over an arena that is this matrix
    w1, m1 = w.clone(), mom.clone()
    grad = ops.gemm_bf16_slab_nt(a, b)
    p1 = ops.to_bf16_slab(w1)
    reg = ops.SgdPlaneRegions([(0, m, n, m, p1, None, None, None)], L.PLANES_BF16)
    seg_end = torch.tensor([m * n], device=dev, dtype=torch.int64)
    ops.acm_sgd_update_planes(grad.view(-1), m1.view(-1), lr, w1.view(-1), seg_end,
                              torch.tensor([lr_mult], device=dev), torch.tensor([wd], device=dev),
                              momentum, 0, gpu_num, it, reg)
    # the update in the GEMM's epilogue
    w2, m2 = w.clone(), mom.clone()
    p2 = ops.to_bf16_slab(w2)
    for r0, r1 in (((0, 256), (256, m)) if blocks else ((0, m),)):
        ops.gemm_bf16_slab_nt_sgd(a[:, r0:r1], b, w2, m2, lr, lr_mult, wd, momentum, 0, gpu_num, it,
                                  p2, rows=(r0, r1))
    assert torch.equal(w1, w2) and torch.equal(m1, m2)
    assert torch.equal(p1.view(torch.int16), p2.view(torch.int16))
    assert torch.equal(p2.view(torch.int16), ops.to_bf16_slab(w2).view(torch.int16))
    assert not torch.equal(w2, w)
    with pytest.raises(L.NawsError):        # N not a multiple of 16
        ops.gemm_bf16_slab_nt_sgd(a, b[:, :n - 8], w2[:, :n - 8], m2[:, :n - 8], lr, lr_mult, wd,
                                  momentum, 0, gpu_num, it, p2)


@pytest.mark.parametrize('r', [37, 400])
def test_roi_pool_writes_the_bf16_slab_operand(dev, r):
    """naws_roi_pool_f_bf16_slab_mapped_fwd + naws_bf16_slab_transpose (the bf16 plan): the pooling
    kernel's one-plane operand equals to_bf16_slab of the fp32 RoIPoolF output, and its transposition
    equals the transposing conversion of the same features, bit for bit (reference operator:
    detectron/ops/roi_loop_pool_op.cu:31-101 + RoIFeatureBoost)."""
    from naws_hip import ops
    from detectron.datasets import synthetic
    mb = synthetic.make_minibatch(synthetic.make_roidb(2, (r + 1) // 2, 20, 160, 224, seed=4), 20)
    rois = torch.from_numpy(mb['rois'][:r]).to(dev).contiguous()
    boost = torch.from_numpy(mb['obn_scores'].reshape(-1)[:r]).to(dev).contiguous()
    x = torch.randn((2, 20, 28, 128), device=dev).relu_()
    want = ops.roi_pool_f(x, rois, 7, 7, 0.125, boost=boost, layout='NHWC', hier=True).view(r, -1)
    m2, m4 = torch.empty_like(x), torch.empty_like(x)
    ops.roi_maxmaps(x, m2, m4)
    slab = ops.roi_pool_f_bf16_slab(x, rois, (m2, m4), 7, 7, 0.125, boost=boost)
    assert slab.shape == (128 * 49 // 16, r, 16)
    assert torch.equal(slab.view(torch.int16), ops.to_bf16_slab(want).view(torch.int16))
    t = ops.bf16_slab_transpose(slab)
    assert torch.equal(t.view(torch.int16), ops.to_bf16_slab(want, transpose=True).view(torch.int16))
