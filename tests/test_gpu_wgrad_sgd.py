"""naws_gemm_f32_f16x2_nt_xk_sgd: the fc6 weight-gradient GEMM whose epilogue applies the ACM SGD
update and writes the weights' operand planes (a run without a gradient exchange).  Checked
against the two-kernel route it replaces - naws_gemm_f32_f16x2_nt_xk, then
naws_acm_sgd_update_f16x2 - bit for bit (reference: FCGradient + detectron/ops/
acm_weightdecay_momentum_sgd_op.h:72-109; the oracle pins the SGD kernel in test_gpu_ops.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(dev, m, n, r, seed):
    from naws_hip import ops
    g = torch.Generator(device=dev).manual_seed(seed)
    dy = torch.randn((r, m), device=dev, generator=g) * 1e-3
    x = torch.randn((r, n), device=dev, generator=g).relu_()
    xp = ops.split_f16x2(x)
    a2 = ops.split_f16x2(dy, transpose=True, rowmul=xp.inv_scale)
    w = torch.randn((m, n), device=dev, generator=g) * 0.02
    w[3] *= 40.0                       # rows of very different magnitude
    w[5] *= 0.25
    mom = torch.randn((m, n), device=dev, generator=g) * 1e-3
    return ops, xp, a2, w, mom


def _two_kernel(ops, L, xp, a2, w, mom, lr, hyper, gpu_num, it, dev, ncols=None):
    """gradient to memory, then the planes-writing SGD kernel over an arena that is this matrix."""
    m, n = w.shape
    lr_mult, wd, momentum = hyper
    grad = torch.zeros_like(w)
    c0, c1 = ncols or (0, n)
    ops.gemm_f32_f16x2_nt_xk(a2, xp, ncols=(c0, c1), out=grad[:, c0:c1])
    planes = ops.split_f16x2(w)
    bound = planes.scales[0].view(torch.int32).clone()
    rowmax = torch.zeros((m,), device=dev, dtype=torch.int32)
    inv = torch.zeros((m,), device=dev)
    ovf = torch.zeros((1,), device=dev, dtype=torch.int32)
    reg = ops.SgdPlaneRegions([(0, m, n, m, planes.planes, bound, rowmax, inv)])
    seg_end = torch.tensor([m * n], device=dev, dtype=torch.int64)
    ops.acm_sgd_update_f16x2(grad.view(-1), mom.view(-1), lr, w.view(-1), seg_end,
                             torch.tensor([lr_mult], device=dev), torch.tensor([wd], device=dev),
                             momentum, 0, gpu_num, it, reg, ovf, 7)
    return planes.planes, rowmax, inv, ovf, bound


@pytest.mark.parametrize('m,n,r,it', [(256, 512, 96, 3),          # 128 x 128 tiles
                                      (4096, 4096, 160, 0),       # 256 x 256 tiles; first iteration
                                      (512, 768, 70, 2)])         # ragged K (rois padded to 96)
def test_wgrad_with_the_update_in_its_epilogue_equals_gemm_then_sgd(dev, m, n, r, it):
    from naws_hip import lib as L
    ops, xp, a2, w, mom = _case(dev, m, n, r, 11 + m)
    lr = torch.tensor([3e-3], device=dev)
    hyper = (1.0, 5e-4, 0.9)
    w1, m1 = w.clone(), mom.clone()
    p1, rm1, inv1, ovf1, bound = _two_kernel(ops, L, xp, a2, w1, m1, lr, hyper, 4, it, dev)
    w2, m2 = w.clone(), mom.clone()
    planes2 = ops.split_f16x2(w2)
    rm2 = torch.zeros((m,), device=dev, dtype=torch.int32)
    inv2 = torch.zeros((m,), device=dev)
    ovf2 = torch.zeros((1,), device=dev, dtype=torch.int32)
    ops.gemm_f32_f16x2_nt_xk_sgd(a2, xp, w2, m2, lr, hyper[0], hyper[1], hyper[2], 0, 4, it,
                                 planes2.planes, bound, rm2, inv2, ovf2, 7)
    torch.cuda.synchronize()
    assert torch.equal(w1, w2) and torch.equal(m1, m2)
    assert not torch.equal(w2, w)
    assert torch.equal(p1.view(torch.int16), planes2.planes.view(torch.int16))
    assert torch.equal(rm1, rm2) and torch.equal(inv1, inv2)
    assert int(ovf1.item()) == int(ovf2.item()) == 0
    # the planes are the updated weights: hi + lo reconstructs them to 2^-22 of the row scale
    dense = (planes2.planes[0].double() + planes2.planes[1].double())        # [n/16, m, 16]
    dense = dense.permute(1, 0, 2).reshape(m, n) * inv2.double()[:, None]
    rowmax = w2.double().abs().amax(dim=1, keepdim=True)
    assert bool(((dense - w2.double()).abs() <= rowmax * 2.0 ** -21).all())


def test_wgrad_sgd_column_and_row_blocks_and_overflow(dev):
    """The engine issues the product in column blocks (tile quantisation) and, with an exchange
    schedule, in row chunks: blocks compose to the whole-matrix call; a row that outgrows twice
    its bound raises the overflow word exactly as the SGD kernel does."""
    from naws_hip import lib as L
    m, n, r = 512, 1024, 64
    ops, xp, a2, w, mom = _case(dev, m, n, r, 5)
    lr = torch.tensor([1e-2], device=dev)
    hyper = (1.0, 0.0, 0.9)
    w1, m1 = w.clone(), mom.clone()
    planes1 = ops.split_f16x2(w1)
    bound = planes1.scales[0].view(torch.int32).clone()
    st1 = [torch.zeros((m,), device=dev, dtype=torch.int32), torch.zeros((m,), device=dev),
           torch.zeros((1,), device=dev, dtype=torch.int32)]
    ops.gemm_f32_f16x2_nt_xk_sgd(a2, xp, w1, m1, lr, *hyper, 0, 2, 1, planes1.planes, bound, *st1, 9)
    w2, m2 = w.clone(), mom.clone()
    planes2 = ops.split_f16x2(w2)
    st2 = [torch.zeros((m,), device=dev, dtype=torch.int32), torch.zeros((m,), device=dev),
           torch.zeros((1,), device=dev, dtype=torch.int32)]
    for r0, r1 in ((0, 256), (256, 512)):
        for c0, c1 in ((0, 768), (768, 1024)):
            ops.gemm_f32_f16x2_nt_xk_sgd(a2.rows(r0, r1), xp, w2, m2, lr, *hyper, 0, 2, 1,
                                         planes2.planes, bound, *st2, 9, ncols=(c0, c1),
                                         rows=(r0, r1))
    torch.cuda.synchronize()
    assert torch.equal(w1, w2) and torch.equal(m1, m2)
    assert torch.equal(planes1.planes.view(torch.int16), planes2.planes.view(torch.int16))
    assert torch.equal(st1[0], st2[0]) and torch.equal(st1[1], st2[1])
    # overflow: a bound far below the row's real size
    w3, m3 = w.clone(), mom.clone()
    planes3 = ops.split_f16x2(w3)
    small = bound.clone()
    small[7] = int(np.float32(1e-6).view(np.int32))
    st3 = [torch.zeros((m,), device=dev, dtype=torch.int32), torch.zeros((m,), device=dev),
           torch.zeros((1,), device=dev, dtype=torch.int32)]
    ops.gemm_f32_f16x2_nt_xk_sgd(a2, xp, w3, m3, lr, *hyper, 0, 2, 1, planes3.planes, small, *st3, 9)
    assert int(st3[2].item()) == 9
    assert torch.equal(w3, w1) and torch.equal(st3[0], st1[0])       # the update itself is unaffected


def test_wgrad_sgd_argument_checks(dev):
    import ctypes as C
    from naws_hip import lib as L
    lib = L.load()
    z = C.c_void_p(0)
    rc = lib.naws_gemm_f32_f16x2_nt_xk_sgd(256, 512, 96, z, 0, 0, z, z, 0, 0, 96, z, z, z, 512, z,
                                           1.0, 0.0, 0.9, 0, 1, 0, z, 0, 256, z, z, z, z, 1, z)
    assert rc == L.ERR_NULL
    rc = lib.naws_gemm_f32_f16x2_nt_xk_sgd(256, 512, 90, z, 0, 0, z, z, 0, 0, 90, z, z, z, 512, z,
                                           1.0, 0.0, 0.9, 0, 1, 0, z, 0, 256, z, z, z, z, 1, z)
    assert rc == L.ERR_UNSUPPORTED
    rc = lib.naws_gemm_f32_f16x2_nt_xk_sgd(256, 512, 96, z, 0, 0, z, z, 0, 0, 96, z, z, z, 512, z,
                                           1.0, 0.0, 0.9, 0, 0, 0, z, 0, 256, z, z, z, z, 1, z)
    assert rc == L.ERR_SHAPE
