"""GPU parity of the fp32-accurate "3 x bf16" GEMM (csrc/gemm_x3.hip).

The claim the kernel's header makes is pinned here: (1) the split is exact,
a == a1 + a2 + a3; (2) against a float64 product, the kernel's error is at the level of the
fp32-MFMA kernel's own (both are fp32 accumulations of K products) - checked against an absolute
bound AND against gemm_f32 on the same data; (3) same epilogues, same dropout stream.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _planes_sum(p):
    from naws_hip import ops
    return ops.planes_to_dense(p)


@pytest.mark.parametrize('transpose', [False, True])
def test_split_is_exact(dev, transpose):
    from naws_hip import ops
    rng = np.random.default_rng(31)
    x = (rng.standard_normal((2, 77, 52)) * np.exp(rng.uniform(-30, 30, (2, 77, 52)))).astype(np.float32)
    x[0, 0, :6] = [0.0, -0.0, 1.0, -1.0, 3.0e38, 1.0e-30]
    xd = _t(x, dev)
    p = ops.split_bf16x3(xd, transpose=transpose)
    k = 77 if transpose else 52
    kp = (k + 15) // 16 * 16
    assert p.shape == ((3, 2, kp // 16, 52, 16) if transpose else (3, 2, kp // 16, 77, 16))
    assert p.dtype == torch.bfloat16
    s = _planes_sum(p).cpu().numpy()
    ref = x.transpose(0, 2, 1) if transpose else x
    assert np.array_equal(s[..., :k], ref.astype(np.float64))       # bit-exact reconstruction
    assert not s[..., k:].any()                                      # zero-filled K pad
    # (residues below the smallest normal bf16, i.e. of |a| < 2^-109, are flushed: there the
    # split keeps 8..23 of the 24 bits - far below anything an activation or weight holds)
    # strided 2-D source (column slice of a wider matrix)
    xs = xd[1][:, 8:40]
    ps = ops.split_bf16x3(xs, transpose=transpose)
    rs = x[1][:, 8:40]
    rs = rs.T if transpose else rs
    assert np.array_equal(_planes_sum(ps).cpu().numpy()[..., :rs.shape[1]], rs.astype(np.float64))


def test_split_nonfinite(dev):
    from naws_hip import ops
    x = np.zeros((16, 16), np.float32)
    x[0, :3] = [np.inf, -np.inf, np.nan]
    p = ops.split_bf16x3(_t(x, dev)).float().cpu().numpy()      # [3, 1, 16, 16]
    assert p[0][0, 0, 0] == np.inf and p[0][0, 0, 1] == -np.inf and np.isnan(p[0][0, 0, 2])
    assert not p[1].any() and not p[2].any()


@pytest.mark.parametrize('m,n,k', [(256, 128, 16), (130, 72, 64), (517, 260, 1000), (64, 4000, 264),
                                   (2100, 140, 96), (300, 200, 25088)])
def test_gemm_x3_fp32_accurate(dev, m, n, k):
    from naws_hip import ops
    rng = np.random.default_rng(32)
    a = rng.uniform(-1, 1, (m, k)).astype(np.float32)
    b = rng.uniform(-1, 1, (n, k)).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64).T
    ad, bd = _t(a, dev), _t(b, dev)
    c = ops.gemm_f32x3_nt(ops.split_bf16x3(ad), ops.split_bf16x3(bd)).cpu().numpy()
    c32 = ops.gemm(ad, bd, False, True).cpu().numpy()
    scale = np.abs(ref).max()
    err = np.abs(c - ref).max() / scale
    err32 = np.abs(c32 - ref).max() / scale
    # the bound the fp32-MFMA kernel is held to (rounding of a K-long fp32 accumulation)
    assert err < 5e-6 * max(1.0, np.sqrt(k / 4096.0)), (err, err32)
    assert err <= 2.0 * err32 + 1e-7, (err, err32)
    # and in the root-mean-square sense
    rms, rms32 = np.sqrt(np.mean((c - ref) ** 2)), np.sqrt(np.mean((c32 - ref) ** 2))
    assert rms <= 2.0 * rms32 + 1e-7 * scale, (rms, rms32)


def test_gemm_x3_wide_dynamic_range(dev):
    """Operands spanning 12 orders of magnitude: the low-order planes carry real information."""
    from naws_hip import ops
    rng = np.random.default_rng(33)
    m, n, k = 192, 160, 512
    a = (rng.standard_normal((m, k)) * np.exp(rng.uniform(-14, 14, (m, 1)))).astype(np.float32)
    b = (rng.standard_normal((n, k)) * np.exp(rng.uniform(-14, 14, (n, 1)))).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64).T
    c = ops.gemm_f32x3_nt(ops.split_bf16x3(_t(a, dev)), ops.split_bf16x3(_t(b, dev))).cpu().numpy()
    # per-entry error relative to |a|.|b| (the natural fp32 bound for a length-k dot product)
    bound = (np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64).T)
    assert (np.abs(c - ref) <= 2e-6 * bound).all()


def test_gemm_x3_identity(dev):
    from naws_hip import ops
    n = 256
    b = (np.arange(n * n, dtype=np.float32).reshape(n, n) * 1.0009765625) % 1013.0
    c = ops.gemm_f32x3_nt(ops.split_bf16x3(_t(np.eye(n, dtype=np.float32), dev)),
                          ops.split_bf16x3(_t(b.T.copy(), dev))).cpu().numpy()
    assert np.array_equal(c, b)               # 1*b through three planes is exact


def test_gemm_x3_transposed_operands(dev):
    """dW = dY^T X: both operands come from transposing splits with a zero-padded K = rows."""
    from naws_hip import ops
    rng = np.random.default_rng(34)
    r, m, n = 203, 96, 300
    dy = rng.uniform(-1, 1, (r, m)).astype(np.float32)
    x = rng.uniform(-1, 1, (r, n)).astype(np.float32)
    ref = dy.astype(np.float64).T @ x.astype(np.float64)
    c = ops.gemm_f32x3_nt(ops.split_bf16x3(_t(dy, dev), transpose=True),
                          ops.split_bf16x3(_t(x, dev), transpose=True))
    assert np.abs(c.cpu().numpy() - ref).max() < 5e-6 * np.abs(ref).max()
    # row-chunked output (what the engine does to overlap the all-reduce)
    a3 = ops.split_bf16x3(_t(dy, dev), transpose=True)
    b3 = ops.split_bf16x3(_t(x, dev), transpose=True)
    out = torch.zeros((m, n), device=dev)
    for r0, r1 in ((0, 40), (40, 96)):
        ops.gemm_f32x3_nt(a3[:, :, r0:r1], b3, out=out[r0:r1])
    assert torch.equal(out, c)


def test_gemm_x3_epilogues_batched(dev):
    from naws_hip import ops, lib
    rng = np.random.default_rng(35)
    m, n, k = 260, 384, 160
    a = rng.uniform(-1, 1, (2, m, k)).astype(np.float32)
    w = rng.uniform(-1, 1, (2, n, k)).astype(np.float32)
    bias = rng.uniform(-1, 1, (2, n)).astype(np.float32)
    zz = np.stack([a[i].astype(np.float64) @ w[i].astype(np.float64).T for i in range(2)])
    z = zz + bias[:, None, :]
    a3, w3, bd = ops.split_bf16x3(_t(a, dev)), ops.split_bf16x3(_t(w, dev)), _t(bias, dev)
    y = ops.gemm_f32x3_nt(a3, w3, epilogue=lib.EPI_BIAS, bias=bd).cpu().numpy()
    np.testing.assert_allclose(y, z, rtol=1e-5, atol=1e-5)
    y = ops.gemm_f32x3_nt(a3, w3, epilogue=lib.EPI_BIAS_RELU, bias=bd).cpu().numpy()
    np.testing.assert_allclose(y, np.maximum(z, 0), rtol=1e-5, atol=1e-5)
    y = ops.gemm_f32x3_nt(a3, w3, epilogue=lib.EPI_BIAS_RELU_DROP, bias=bd, drop_ratio=0.5,
                          seed=77).cpu().numpy()
    mask = ops.dropout_mask(77, 0.5, 2 * m * n, dev).reshape(2, m, n).cpu().numpy()
    np.testing.assert_allclose(y, np.maximum(z, 0) * mask * 2.0, rtol=1e-5, atol=2e-5)
    aux = rng.standard_normal((2, m, n)).astype(np.float32)
    gt = ops.gemm_f32x3_nt(a3, w3, epilogue=lib.EPI_GATE_POS, aux=_t(aux, dev), alpha=2.0)
    np.testing.assert_allclose(gt.cpu().numpy(), np.where(aux > 0, zz * 2.0, 0.0), rtol=1e-5,
                               atol=2e-5)
    c0 = rng.standard_normal((2, m, n)).astype(np.float32)
    cd = _t(c0, dev)
    ops.gemm_f32x3_nt(a3, w3, out=cd, accumulate=True)
    np.testing.assert_allclose(cd.cpu().numpy(), c0 + zz, rtol=1e-5, atol=2e-5)
    with pytest.raises(lib.NawsError):      # K mismatch
        ops.gemm_f32x3_nt(a3, ops.split_bf16x3(_t(w[:, :, :96].copy(), dev)))


def test_gemm_x3_deterministic(dev):
    from naws_hip import ops
    rng = np.random.default_rng(36)
    a3 = ops.split_bf16x3(_t(rng.standard_normal((700, 1024)).astype(np.float32), dev))
    b3 = ops.split_bf16x3(_t(rng.standard_normal((900, 1024)).astype(np.float32), dev))
    c1 = ops.gemm_f32x3_nt(a3, b3)
    for _ in range(5):      # the LDS-DMA pipeline has no run-to-run variation (race screen)
        assert torch.equal(ops.gemm_f32x3_nt(a3, b3), c1)


@pytest.mark.parametrize('cin,cout,dil,h,w', [(64, 64, 1, 37, 53), (64, 128, 1, 40, 60),
                                              (128, 128, 1, 8, 32), (128, 128, 1, 67, 97),
                                              (64, 64, 2, 21, 40), (128, 256, 1, 19, 23),
                                              (512, 512, 2, 20, 31), (256, 512, 1, 75, 125)])
def test_conv3x3_f32x3(dev, cin, cout, dil, h, w):
    """Held to the tolerance of the fp32-MFMA convolution (tests/test_gpu_ops.py)."""
    from naws_hip import ops
    import torch.nn.functional as F
    rng = np.random.default_rng(37)
    n = 2
    x = rng.uniform(-1, 1, (n, cin, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, cout).astype(np.float32)
    ref = F.relu(F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(),
                          torch.from_numpy(b).double(), padding=dil, dilation=dil)).numpy()
    xd = ops.nchw_to_nhwc(_t(x, dev))
    wp = ops.conv3x3_pack_weight(_t(wt, dev))
    w3 = ops.split_bf16x3(wp.view(cout, 9 * cin))
    y = ops.nhwc_to_nchw(ops.conv3x3_nhwc_f32x3(xd, w3, _t(b, dev), dil, True)).cpu().numpy()
    y32 = ops.nhwc_to_nchw(ops.conv3x3_nhwc(xd, wp, _t(b, dev), dil, True)).cpu().numpy()
    scale = max(1.0, np.abs(ref).max())
    assert np.abs(y - ref).max() < 1e-5 * scale
    assert np.abs(y - ref).max() <= 2.0 * np.abs(y32 - ref).max() + 1e-6 * scale
    # no bias / no ReLU
    y2 = ops.conv3x3_nhwc_f32x3(xd, w3, None, dil, False)
    r2 = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), None, padding=dil,
                  dilation=dil).numpy()
    assert np.abs(ops.nhwc_to_nchw(y2).cpu().numpy() - r2).max() < 1e-5 * scale
    if dil == 1 and cout % 64 == 0 and cout <= 256 and h >= 2 and w >= 2:
        # the 2x2 / stride-2 max-pool in the epilogue == the pooling kernel on the un-pooled output
        yd = ops.conv3x3_nhwc_f32x3(xd, w3, _t(b, dev), 1, True)
        yp = ops.conv3x3_nhwc_f32x3(xd, w3, _t(b, dev), 1, True, pool2=True)
        assert torch.equal(yp, ops.maxpool2x2_nhwc(yd, 2))


@pytest.mark.parametrize('cin,cout,dil,h,w', [(128, 256, 1, 19, 23), (512, 512, 2, 20, 31),
                                              (256, 256, 1, 38, 63)])
def test_conv3x3_winograd_f32x3(dev, cin, cout, dil, h, w):
    from naws_hip import ops
    import torch.nn.functional as F
    rng = np.random.default_rng(38)
    n = 2
    x = rng.uniform(-1, 1, (n, cin, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, cout).astype(np.float32)
    ref = F.relu(F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(),
                          torch.from_numpy(b).double(), padding=dil, dilation=dil)).numpy()
    xd = ops.nchw_to_nhwc(_t(x, dev))
    u = ops.winograd_weight_transform(_t(wt, dev))
    y32 = ops.nhwc_to_nchw(ops.conv3x3_winograd_nhwc(xd, u, _t(b, dev), dil, True)).cpu().numpy()
    y = ops.nhwc_to_nchw(ops.conv3x3_winograd_nhwc_f32x3(xd, ops.split_bf16x3(u), _t(b, dev), dil,
                                                         True)).cpu().numpy()
    scale = max(1.0, np.abs(ref).max())
    assert np.abs(y - ref).max() < 1e-4 * scale          # the fp32 Winograd test's bound
    assert np.abs(y - ref).max() <= 2.0 * np.abs(y32 - ref).max() + 1e-6 * scale
