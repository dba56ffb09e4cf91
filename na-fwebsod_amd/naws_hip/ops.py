"""Tensor-level wrappers over the C ABI (include/naws.h).

torch is plumbing only: it owns device memory and the current HIP stream; every
computation below is one or more hand-written gfx950 kernels in libnaws_hip.so.
Inputs must be CUDA(HIP) fp32 tensors; anything else raises (no CPU fallback).
"""
import ctypes as C

import numpy as np
import torch

from . import lib as L

_f32 = torch.float32


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, name, dtype=_f32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError('%s must be a HIP device tensor (the naws ops have no CPU path)' % name)
    if t.dtype != dtype:
        raise TypeError('%s must be %s, got %s' % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError('%s must be contiguous' % name)
    return t


def _ptr(t):
    return 0 if t is None else t.data_ptr()


# ----------------------------------------------------------------------------
# conv body
# ----------------------------------------------------------------------------
def conv3x3_c3_nchw_to_nhwc(x, w_oihw, bias, relu=True, out=None):
    _chk(x, 'x'); _chk(w_oihw, 'w')
    n, c, h, w = x.shape
    assert c == 3 and tuple(w_oihw.shape[1:]) == (3, 3, 3)
    cout = w_oihw.shape[0]
    y = out if out is not None else torch.empty((n, h, w, cout), device=x.device, dtype=_f32)
    L.call('naws_conv3x3_c3_nchw_to_nhwc_fwd', x.data_ptr(), w_oihw.data_ptr(), _ptr(bias),
           n, h, w, cout, int(relu), y.data_ptr(), _stream())
    return y


def conv3x3_pack_weight(w_oihw):
    _chk(w_oihw, 'w')
    cout, cin = w_oihw.shape[:2]
    wp = torch.empty((cout, 3, 3, cin), device=w_oihw.device, dtype=_f32)
    L.call('naws_conv3x3_pack_weight', w_oihw.data_ptr(), cout, cin, wp.data_ptr(), _stream())
    return wp


def conv3x3_nhwc(x, w_packed, bias, dilation=1, relu=True, out=None):
    _chk(x, 'x'); _chk(w_packed, 'w_packed')
    n, h, w, cin = x.shape
    cout = w_packed.shape[0]
    y = out if out is not None else torch.empty((n, h, w, cout), device=x.device, dtype=_f32)
    L.call('naws_conv3x3_nhwc_fwd', x.data_ptr(), w_packed.data_ptr(), _ptr(bias), n, h, w, cin,
           cout, dilation, int(relu), y.data_ptr(), _stream())
    return y


def winograd_weight_transform(w_oihw):
    _chk(w_oihw, 'w')
    cout, cin = w_oihw.shape[:2]
    u = torch.empty((16, cout, cin), device=w_oihw.device, dtype=_f32)
    L.call('naws_winograd_weight_transform', w_oihw.data_ptr(), cout, cin, u.data_ptr(), _stream())
    return u


def winograd4_weight_transform(w_oihw):
    """U [36, Cout, Cin] = G g G^T of F(4x4, 3x3) (csrc/winograd4.hip), frequency 6 i + j."""
    _chk(w_oihw, 'w')
    cout, cin = w_oihw.shape[:2]
    u = torch.empty((36, cout, cin), device=w_oihw.device, dtype=_f32)
    L.call('naws_winograd4_weight_transform', w_oihw.data_ptr(), cout, cin, u.data_ptr(), _stream())
    return u


def conv3x3_winograd_nhwc(x, u, bias, dilation=1, relu=True, out=None):
    _chk(x, 'x'); _chk(u, 'U')
    n, h, w, cin = x.shape
    cout = u.shape[1]
    y = out if out is not None else torch.empty((n, h, w, cout), device=x.device, dtype=_f32)
    nws = L.load().naws_winograd_workspace_floats(n, h, w, cin, cout, dilation)
    ws = torch.empty((nws,), device=x.device, dtype=_f32)
    L.call('naws_conv3x3_winograd_nhwc_fwd', x.data_ptr(), u.data_ptr(), _ptr(bias), n, h, w, cin,
           cout, dilation, int(relu), ws.data_ptr(), y.data_ptr(), _stream())
    return y


def maxpool2x2_nhwc(x, stride, out=None):
    _chk(x, 'x')
    n, h, w, c = x.shape
    ho, wo = (h - 2) // stride + 1, (w - 2) // stride + 1
    y = out if out is not None else torch.empty((n, ho, wo, c), device=x.device, dtype=_f32)
    L.call('naws_maxpool2x2_nhwc_fwd', x.data_ptr(), n, h, w, c, stride, y.data_ptr(), _stream())
    return y


def nchw_to_nhwc(x):
    _chk(x, 'x')
    n, c, h, w = x.shape
    y = torch.empty((n, h, w, c), device=x.device, dtype=_f32)
    L.call('naws_nchw_to_nhwc', x.data_ptr(), n, c, h, w, y.data_ptr(), _stream())
    return y


def nhwc_to_nchw(x):
    _chk(x, 'x')
    n, h, w, c = x.shape
    y = torch.empty((n, c, h, w), device=x.device, dtype=_f32)
    L.call('naws_nhwc_to_nchw', x.data_ptr(), n, h, w, c, y.data_ptr(), _stream())
    return y


# ----------------------------------------------------------------------------
# RoI ops
# ----------------------------------------------------------------------------
def roi_pool_f(x, rois, pooled_h=7, pooled_w=7, spatial_scale=0.125, boost=None,
               layout='NCHW', with_argmax=False, out=None, hier=False):
    _chk(x, 'x'); _chk(rois, 'rois')
    if rois.dim() != 2 or rois.shape[1] != 5:
        raise L.NawsError('naws_roi_pool_f_fwd', L.ERR_SHAPE)
    if layout == 'NCHW':
        n, c, h, w = x.shape
        lay = L.LAYOUT_NCHW
    elif layout == 'NHWC':
        n, h, w, c = x.shape
        lay = L.LAYOUT_NHWC
    else:
        raise L.NawsError('naws_roi_pool_f_fwd', L.ERR_ARG)
    r = rois.shape[0]
    if boost is not None:
        _chk(boost, 'boost')
        assert boost.numel() == r
    y = out if out is not None else torch.empty((r, c, pooled_h, pooled_w), device=x.device,
                                                dtype=_f32)
    if hier and layout == 'NHWC' and not with_argmax and c % 64 == 0 and r > 0:
        # bin maxima over precomputed 2x2 / 4x4 block maxima: same values, ~8x less gather
        ws = torch.empty((L.load().naws_roi_pool_workspace_floats(n, c, h, w),), device=x.device,
                         dtype=_f32)
        L.call('naws_roi_pool_f_nhwc_hier_fwd', x.data_ptr(), n, c, h, w, rois.data_ptr(), r,
               _ptr(boost), pooled_h, pooled_w, float(spatial_scale), ws.data_ptr(), y.data_ptr(),
               _stream())
        return y
    am = torch.empty((r, c, pooled_h, pooled_w), device=x.device, dtype=torch.int32) \
        if with_argmax else None
    L.call('naws_roi_pool_f_fwd', x.data_ptr(), lay, n, c, h, w, rois.data_ptr(), r, _ptr(boost),
           pooled_h, pooled_w, float(spatial_scale), y.data_ptr(), _ptr(am), _stream())
    return (y, am) if with_argmax else y


def roi_maxmaps(x, m2, m4):
    """The 2x2 / 4x4 block-maxima maps of NHWC features x into m2 / m4 (same shape as x), on the
    current stream (the first half of roi_pool_f_f16x2(hier=True))."""
    _chk(x, 'x'); _chk(m2, 'm2'); _chk(m4, 'm4')
    if m2.shape != x.shape or m4.shape != x.shape or not (x.is_contiguous() and m2.is_contiguous()
                                                          and m4.is_contiguous()):
        raise TypeError('roi_maxmaps: contiguous maps of x\'s shape')
    n, h, w, c = x.shape
    L.call('naws_roi_maxmaps_fwd', x.data_ptr(), n, c, h, w, m2.data_ptr(), m4.data_ptr(), _stream())


def roi_pool_operand(r, k, device):
    """An empty fp16x2 fc6 operand for r rois x k features (what roi_pool_f_f16x2 fills)."""
    return F16x2(torch.empty((2, k // 16, r, 16), device=device, dtype=torch.float16),
                 torch.empty((2, r), device=device, dtype=_f32))


def roi_pool_f_f16x2_range(x, rois, amax_words, out, r0, r1, maps, pooled_h=7, pooled_w=7,
                           spatial_scale=0.125, boost=None):
    """roi_pool_f_f16x2 over existing block-maxima maps for rois r0..r1 only, into rows r0..r1 of
    `out` (roi_pool_operand for ALL of rois): one image's proposals on that image's stream."""
    _chk(x, 'x'); _chk(rois, 'rois')
    n, h, w, c = x.shape
    r = rois.shape[0]
    m2, m4 = maps
    _chk(m2, 'm2'); _chk(m4, 'm4')
    if boost is not None:
        _chk(boost, 'boost')
        assert boost.numel() == r
    if out.planes.shape != (2, c * pooled_h * pooled_w // 16, r, 16) or m2.shape != x.shape:
        raise L.NawsError('naws_roi_pool_f_f16x2_mapped_range_fwd', L.ERR_SHAPE)
    L.call('naws_roi_pool_f_f16x2_mapped_range_fwd', x.data_ptr(), n, c, h, w, rois.data_ptr(), r,
           int(r0), int(r1 - r0), _ptr(boost), pooled_h, pooled_w, float(spatial_scale),
           amax_words.data_ptr(), amax_words.numel(), m2.data_ptr(), m4.data_ptr(),
           out.planes.data_ptr(), out.scales.data_ptr(), _stream())
    return out


def roi_pool_f_f16x2(x, rois, amax_words, pooled_h=7, pooled_w=7, spatial_scale=0.125, boost=None,
                     hier=True, maps=None):
    """RoIPoolF (+ boost) on NHWC features, written directly as the fp16x2 operand of the fc6
    GEMM: F16x2 with planes [2, K/16, R, 16], K = C*ph*pw, scaled per roi from the bound
    max|x| of the roi's image (`amax_words`: int32 [n] bit patterns) * |boost|."""
    _chk(x, 'x'); _chk(rois, 'rois')
    if rois.dim() != 2 or rois.shape[1] != 5:
        raise L.NawsError('naws_roi_pool_f_f16x2_fwd', L.ERR_SHAPE)
    n, h, w, c = x.shape
    r = rois.shape[0]
    k = c * pooled_h * pooled_w
    if boost is not None:
        _chk(boost, 'boost')
        assert boost.numel() == r
    if amax_words.dtype != torch.int32 or not amax_words.is_contiguous():
        raise TypeError('amax_words must be a contiguous int32 tensor')
    out = F16x2(torch.empty((2, k // 16, r, 16), device=x.device, dtype=torch.float16),
                torch.empty((2, r), device=x.device, dtype=_f32))
    if maps is not None:           # (m2, m4) already built by roi_maxmaps
        m2, m4 = maps
        _chk(m2, 'm2'); _chk(m4, 'm4')
        if m2.shape != x.shape or m4.shape != x.shape:
            raise TypeError('maps must have x\'s shape')
        L.call('naws_roi_pool_f_f16x2_mapped_fwd', x.data_ptr(), n, c, h, w, rois.data_ptr(), r,
               _ptr(boost), pooled_h, pooled_w, float(spatial_scale), amax_words.data_ptr(),
               amax_words.numel(), m2.data_ptr(), m4.data_ptr(), out.planes.data_ptr(),
               out.scales.data_ptr(), _stream())
        return out
    if hier:
        ws = torch.empty((L.load().naws_roi_pool_workspace_floats(n, c, h, w),), device=x.device,
                         dtype=_f32)
        L.call('naws_roi_pool_f_f16x2_hier_fwd', x.data_ptr(), n, c, h, w, rois.data_ptr(), r,
               _ptr(boost), pooled_h, pooled_w, float(spatial_scale), amax_words.data_ptr(),
               amax_words.numel(), ws.data_ptr(), out.planes.data_ptr(), out.scales.data_ptr(),
               _stream())
        return out
    L.call('naws_roi_pool_f_f16x2_fwd', x.data_ptr(), n, c, h, w, rois.data_ptr(), r, _ptr(boost),
           pooled_h, pooled_w, float(spatial_scale), amax_words.data_ptr(), amax_words.numel(),
           out.planes.data_ptr(), out.scales.data_ptr(), _stream())
    return out


def roi_pool_f_bf16_slab(x, rois, maps, pooled_h=7, pooled_w=7, spatial_scale=0.125, boost=None):
    """RoIPoolF (+ boost) on NHWC features over the block-maxima maps (m2, m4) of roi_maxmaps,
    written directly as the bf16 plan's fc6 operand: bf16 [K/16, R, 16], K = C*ph*pw (the layout
    of to_bf16_slab; same pooled values, rounded to nearest-even)."""
    _chk(x, 'x'); _chk(rois, 'rois')
    if rois.dim() != 2 or rois.shape[1] != 5:
        raise L.NawsError('naws_roi_pool_f_bf16_slab_mapped_fwd', L.ERR_SHAPE)
    n, h, w, c = x.shape
    r = rois.shape[0]
    k = c * pooled_h * pooled_w
    if boost is not None:
        _chk(boost, 'boost')
        assert boost.numel() == r
    m2, m4 = maps
    _chk(m2, 'm2'); _chk(m4, 'm4')
    if m2.shape != x.shape or m4.shape != x.shape:
        raise TypeError('maps must have x\'s shape')
    out = torch.empty((k // 16, r, 16), device=x.device, dtype=torch.bfloat16)
    L.call('naws_roi_pool_f_bf16_slab_mapped_fwd', x.data_ptr(), n, c, h, w, rois.data_ptr(), r,
           _ptr(boost), pooled_h, pooled_w, float(spatial_scale), m2.data_ptr(), m4.data_ptr(),
           out.data_ptr(), _stream())
    return out


def bf16_slab_transpose(p):
    """bf16 slab operand [K/16, R, 16] -> [Rpad/16, K, 16] (Rpad = R rounded up to 64, rows >= R
    zero): the same matrix with the other index K-contiguous (to_bf16_slab(x, transpose=True) of
    the matrix the slab was rounded from)."""
    if (not p.is_cuda or p.dtype != torch.bfloat16 or p.dim() != 3 or p.shape[-1] != 16
            or not p.is_contiguous()):
        raise TypeError('p must be a contiguous bf16 slab tensor [K/16, R, 16]')
    k, r = p.shape[0] * 16, p.shape[1]
    rp = (r + 63) // 64 * 64
    q = torch.empty((rp // 16, k, 16), device=p.device, dtype=torch.bfloat16)
    L.call('naws_bf16_slab_transpose', p.data_ptr(), r, k, rp, q.data_ptr(), _stream())
    return q


def f16_planes_transpose(op):
    """F16x2 with planes [2, K/16, R, 16] -> F16x2 with planes [2, Rpad/16, K, 16] holding the same
    scaled matrix K(=rows)-contiguous (Rpad = R rounded up to 32, zero rows) and scales of ones:
    the B operand of dW = dY^T X when dY is split with kmul = op.inv_scale."""
    p = op.planes
    if p.dim() != 4 or p.dtype != torch.float16 or not p.is_contiguous():
        raise TypeError('expects contiguous unbatched f16 planes [2, K/16, R, 16]')
    k, r = p.shape[1] * 16, p.shape[2]
    rpad = (r + 31) // 32 * 32
    q = torch.empty((2, rpad // 16, k, 16), device=p.device, dtype=torch.float16)
    L.call('naws_f16_planes_transpose', p.data_ptr(), r, k, rpad, q.data_ptr(), _stream())
    return F16x2(q, torch.ones((2, k), device=p.device, dtype=_f32))


def roi_feature_boost(x, s, out=None):
    _chk(x, 'X'); _chk(s, 'S')
    if s.shape[0] != s.numel() or x.shape[0] != s.shape[0]:
        raise L.NawsError('naws_roi_feature_boost_fwd', L.ERR_SHAPE)
    r = x.shape[0]
    f = x.numel() // max(r, 1)
    y = out if out is not None else torch.empty_like(x)
    L.call('naws_roi_feature_boost_fwd', x.data_ptr(), s.data_ptr(), r, f, y.data_ptr(), _stream())
    return y


def roi_feature_boost_grad(dy, s):
    _chk(dy, 'dY'); _chk(s, 'S')
    if s.shape[0] != s.numel() or dy.shape[0] != s.shape[0]:
        raise L.NawsError('naws_roi_feature_boost_bwd', L.ERR_SHAPE)
    r = dy.shape[0]
    f = dy.numel() // max(r, 1)
    dx = torch.empty_like(dy)
    L.call('naws_roi_feature_boost_bwd', dy.data_ptr(), s.data_ptr(), r, f, dx.data_ptr(),
           _stream())
    return dx


def roi_iou(rois):
    _chk(rois, 'rois')
    if rois.dim() != 2 or rois.shape[1] != 5:
        raise L.NawsError('naws_roi_iou_fwd', L.ERR_SHAPE)
    r = rois.shape[0]
    j = torch.empty((r, r), device=rois.device, dtype=_f32)
    L.call('naws_roi_iou_fwd', rois.data_ptr(), r, j.data_ptr(), _stream())
    return j


# ----------------------------------------------------------------------------
# GEMM / FC
# ----------------------------------------------------------------------------
def _amax_args(rowmax, rowmax_seg, colmax, colmax_rowmul):
    for t in (rowmax, colmax):
        if t is not None and (t.dtype != torch.int32 or not t.is_contiguous()):
            raise TypeError('rowmax / colmax must be contiguous int32 (bit pattern) tensors')
    if colmax_rowmul is not None:
        _chk(colmax_rowmul, 'colmax_rowmul')
    return (_ptr(rowmax), int(rowmax_seg), _ptr(colmax), _ptr(colmax_rowmul))


def gemm(a, b, trans_a=False, trans_b=False, out=None, epilogue=L.EPI_NONE, bias=None, aux=None,
         alpha=1.0, drop_ratio=0.0, seed=0, accumulate=False, rowmax=None, rowmax_seg=0,
         colmax=None, colmax_rowmul=None):
    """C[M,N] (+)= op(A) op(B) on row-major 2-D (or batched 3-D) tensors.  Inputs may be
    row-strided views (last dim contiguous).  rowmax / colmax (zeroed int32 [batch, nseg, M] /
    [batch, N]): receive the bit patterns of max|C| per row (per column segment of rowmax_seg
    columns) / per column (of diag(colmax_rowmul) C) - what split_f16x2_dual consumes."""
    batched = a.dim() == 3
    if batched:
        batch = a.shape[0]
        a2, b2 = a[0], b[0]
    else:
        batch, a2, b2 = 1, a, b
    for t, nm in ((a2, 'A'), (b2, 'B')):
        if not t.is_cuda or t.dtype != _f32 or t.stride(-1) != 1:
            raise TypeError('%s must be a HIP fp32 tensor with a contiguous last dim' % nm)
    m = a2.shape[1] if trans_a else a2.shape[0]
    k = a2.shape[0] if trans_a else a2.shape[1]
    kb = b2.shape[1] if trans_b else b2.shape[0]
    n = b2.shape[0] if trans_b else b2.shape[1]
    if k != kb:
        raise L.NawsError('naws_gemm_f32', L.ERR_SHAPE)
    if out is None:
        out = torch.empty(((batch, m, n) if batched else (m, n)), device=a.device, dtype=_f32)
    c2 = out[0] if batched else out
    sa = a.stride(0) if batched else 0
    sb = b.stride(0) if batched else 0
    sc = out.stride(0) if batched else 0
    sbias = 0
    if bias is not None and bias.dim() == 2:
        sbias = bias.stride(0)
    L.call('naws_gemm_f32_amax', int(trans_a), int(trans_b), m, n, k, a.data_ptr(), a2.stride(0),
           b.data_ptr(), b2.stride(0), out.data_ptr(), c2.stride(0), batch, sa, sb, sc,
           epilogue, _ptr(bias), sbias, _ptr(aux),
           (aux.stride(-2) if aux is not None else 0), float(alpha), float(drop_ratio),
           int(seed) & 0xFFFFFFFFFFFFFFFF, int(accumulate),
           *_amax_args(rowmax, rowmax_seg, colmax, colmax_rowmul), _stream())
    return out


def gemm_splitk(a, b, trans_a=False, trans_b=False, out=None, epilogue=L.EPI_NONE, bias=None,
                ksplit=4, workspace=None):
    """gemm() for a small output with a long inner dimension (fc8 forward / wgrad): K in `ksplit`
    slices, partial products in `workspace` (fp32, >= M * N * batch * ksplit floats; allocated when
    None), summed in slice order by a second pass that applies the epilogue (NONE / BIAS)."""
    batched = a.dim() == 3
    if batched:
        batch = a.shape[0]
        a2, b2 = a[0], b[0]
    else:
        batch, a2, b2 = 1, a, b
    for t, nm in ((a2, 'A'), (b2, 'B')):
        if not t.is_cuda or t.dtype != _f32 or t.stride(-1) != 1:
            raise TypeError('%s must be a HIP fp32 tensor with a contiguous last dim' % nm)
    m = a2.shape[1] if trans_a else a2.shape[0]
    k = a2.shape[0] if trans_a else a2.shape[1]
    kb = b2.shape[1] if trans_b else b2.shape[0]
    n = b2.shape[0] if trans_b else b2.shape[1]
    if k != kb:
        raise L.NawsError('naws_gemm_f32_splitk', L.ERR_SHAPE)
    if out is None:
        out = torch.empty(((batch, m, n) if batched else (m, n)), device=a.device, dtype=_f32)
    c2 = out[0] if batched else out
    need = m * n * batch * int(ksplit)
    if workspace is None:
        workspace = torch.empty((need,), device=a.device, dtype=_f32)
    if workspace.dtype != _f32 or workspace.numel() < need or not workspace.is_contiguous():
        raise TypeError('workspace: contiguous fp32, >= M * N * batch * ksplit floats')
    sbias = bias.stride(0) if (bias is not None and bias.dim() == 2) else 0
    L.call('naws_gemm_f32_splitk', int(trans_a), int(trans_b), m, n, k, a.data_ptr(), a2.stride(0),
           b.data_ptr(), b2.stride(0), out.data_ptr(), c2.stride(0), batch,
           (a.stride(0) if batched else 0), (b.stride(0) if batched else 0),
           (out.stride(0) if batched else 0), epilogue, _ptr(bias), sbias, int(ksplit),
           workspace.data_ptr(), _stream())
    return out


def gemm_bf16_nt(a, b, out=None, epilogue=L.EPI_NONE, bias=None, aux=None, alpha=1.0,
                 drop_ratio=0.0, seed=0, accumulate=False):
    """C[M,N] (+)= A[M,K] B[N,K]^T with bf16 MFMA / fp32 accumulate.  a and b are 2-D (or batched
    3-D) row-strided views, each fp32 (rounded in-kernel) or bf16."""
    batched = a.dim() == 3
    a2, b2 = (a[0], b[0]) if batched else (a, b)
    batch = a.shape[0] if batched else 1
    for t, nm in ((a2, 'A'), (b2, 'B')):
        if not t.is_cuda or t.dtype not in (_f32, torch.bfloat16) or t.stride(-1) != 1:
            raise TypeError('%s must be a HIP fp32/bf16 tensor with a contiguous last dim' % nm)
    m, k = a2.shape
    n, kb = b2.shape
    if k != kb:
        raise L.NawsError('naws_gemm_bf16_nt', L.ERR_SHAPE)
    if out is None:
        out = torch.empty(((batch, m, n) if batched else (m, n)), device=a.device, dtype=_f32)
    c2 = out[0] if batched else out
    sbias = bias.stride(0) if (bias is not None and bias.dim() == 2) else 0
    L.call('naws_gemm_bf16_nt', m, n, k, a.data_ptr(), int(a.dtype == torch.bfloat16),
           a2.stride(0), b.data_ptr(), int(b.dtype == torch.bfloat16), b2.stride(0),
           out.data_ptr(), c2.stride(0), batch, (a.stride(0) if batched else 0),
           (b.stride(0) if batched else 0), (out.stride(0) if batched else 0), epilogue,
           _ptr(bias), sbias, _ptr(aux), (aux.stride(-2) if aux is not None else 0),
           float(alpha), float(drop_ratio), int(seed) & 0xFFFFFFFFFFFFFFFF, int(accumulate),
           _stream())
    return out


def transpose_to_bf16(x, rows_pad=None, out=None):
    """fp32 [rows, cols] (or [b, rows, cols], batch-contiguous) -> bf16 [cols, rows_pad]."""
    batched = x.dim() == 3
    x2 = x[0] if batched else x
    batch = x.shape[0] if batched else 1
    rows, cols = x2.shape
    if x2.stride(1) != 1 or (batched and x.stride(0) != rows * x2.stride(0)):
        raise TypeError('x must be row-major with batch stride rows*ld')
    rp = rows_pad if rows_pad is not None else (rows + 7) // 8 * 8
    shape = (batch, cols, rp) if batched else (cols, rp)
    y = out if out is not None else torch.empty(shape, device=x.device, dtype=torch.bfloat16)
    L.call('naws_transpose_to_bf16', x.data_ptr(), batch, rows, cols, x2.stride(0), rp,
           y.data_ptr(), _stream())
    return y


def conv3x3_nhwc_bf16(x, w_packed, bias, dilation=1, relu=True, out=None):
    _chk(x, 'x'); _chk(w_packed, 'w_packed')
    n, h, w, cin = x.shape
    cout = w_packed.shape[0]
    y = out if out is not None else torch.empty((n, h, w, cout), device=x.device, dtype=_f32)
    L.call('naws_conv3x3_nhwc_bf16_fwd', x.data_ptr(), w_packed.data_ptr(), _ptr(bias), n, h, w,
           cin, cout, dilation, int(relu), y.data_ptr(), _stream())
    return y


def conv3x3_nhwc_bf16_wp(x, w_slab, bias, dilation=1, relu=True, out=None, pool2=False):
    """The bf16 plan's 3x3 convolution on the wave-private halo-tile kernel: w_slab =
    to_bf16_slab(conv3x3_pack_weight(w).view(Cout, -1)) (bf16 [9*Cin/16, Cout, 16]); fp32 NHWC in
    and out.  pool2: the 2x2 / stride-2 max-pool that follows, taken in the epilogue."""
    _chk(x, 'x')
    if (not w_slab.is_cuda or w_slab.dtype != torch.bfloat16 or w_slab.dim() != 3
            or not w_slab.is_contiguous() or w_slab.shape[-1] != 16):
        raise TypeError('w_slab must be a contiguous bf16 slab tensor [9*Cin/16, Cout, 16]')
    n, h, w, cin = x.shape
    cout = w_slab.shape[1]
    if w_slab.shape[0] * 16 != 9 * cin:
        raise ValueError('weight slab K does not match 9 * Cin')
    shape = (n, h // 2, w // 2, cout) if pool2 else (n, h, w, cout)
    y = out if out is not None else torch.empty(shape, device=x.device, dtype=_f32)
    if tuple(y.shape) != shape or y.dtype != _f32 or not y.is_contiguous():
        raise TypeError('out must be a contiguous fp32 %s tensor' % (shape,))
    L.call('naws_conv3x3_nhwc_bf16_wp_fwd', x.data_ptr(), w_slab.data_ptr(), _ptr(bias), n, h, w,
           cin, cout, dilation, int(relu), int(pool2), y.data_ptr(), _stream())
    return y


def split_bf16x3(x, transpose=False, out=None):
    """fp32 [rows, cols] or [b, rows, cols] (last dim contiguous) -> bf16 planes
    [3, (b,) K/16, outer, 16] ("K-slab-major"): outer/K = rows/cols, or cols/rows when
    `transpose`; K is rounded up to 16 (zero-filled).  x == P[0] + P[1] + P[2] exactly."""
    batched = x.dim() == 3
    x2 = x[0] if batched else x
    if not x.is_cuda or x.dtype != _f32 or x2.stride(1) != 1:
        raise TypeError('x must be a HIP fp32 tensor with a contiguous last dim')
    batch = x.shape[0] if batched else 1
    rows, cols = x2.shape
    outer, k = (cols, rows) if transpose else (rows, cols)
    kpad = (k + 15) // 16 * 16
    shape = (3, batch, kpad // 16, outer, 16) if batched else (3, kpad // 16, outer, 16)
    p = out if out is not None else torch.empty(shape, device=x.device, dtype=torch.bfloat16)
    L.call('naws_split_bf16x3', x.data_ptr(), batch, rows, cols, x2.stride(0),
           (x.stride(0) if batched else 0), int(transpose), kpad, p.data_ptr(), _stream())
    return p


def conv3x3_nhwc_f32x3(x, w3, bias, dilation=1, relu=True, out=None, pool2=False):
    """3x3 conv on NHWC fp32 activations with weight planes w3 = split_bf16x3(packed weight
    viewed [Cout, 9*Cin]); fp32-accurate, bf16 MFMA.  pool2: the 2x2 / stride-2 max-pool that
    follows is taken in the kernel's epilogue (dilation 1, Cout % 64 == 0, Cout <= 256)."""
    _chk(x, 'x')
    n, h, w, cin = x.shape
    cout = w3.shape[-2]
    if w3.dtype != torch.bfloat16 or w3.shape[0] != 3 or w3.shape[1] * 16 != 9 * cin:
        raise TypeError('w3 must be the bf16 planes [3, 9*Cin/16, Cout, 16] of the packed weight')
    if pool2:
        y = torch.empty((n, h // 2, w // 2, cout), device=x.device, dtype=_f32)
        L.call('naws_conv3x3_nhwc_f32x3_pool_fwd', x.data_ptr(), w3.data_ptr(), _ptr(bias), n, h, w,
               cin, cout, int(relu), y.data_ptr(), _stream())
        return y
    y = out if out is not None else torch.empty((n, h, w, cout), device=x.device, dtype=_f32)
    L.call('naws_conv3x3_nhwc_f32x3_fwd', x.data_ptr(), w3.data_ptr(), _ptr(bias), n, h, w, cin,
           cout, dilation, int(relu), y.data_ptr(), _stream())
    return y


def conv3x3_winograd_nhwc_f32x3(x, u3, bias, dilation=1, relu=True, out=None):
    """Winograd F(2x2,3x3) with fp32x3 GEMMs; u3 = split_bf16x3(winograd_weight_transform(w))
    = planes [3, 16, Cin/16, Cout, 16]."""
    _chk(x, 'x')
    n, h, w, cin = x.shape
    cout = u3.shape[-2]
    if u3.dtype != torch.bfloat16 or tuple(u3.shape[:3]) != (3, 16, cin // 16):
        raise TypeError('u3 must be the bf16 planes [3, 16, Cin/16, Cout, 16] of U')
    y = out if out is not None else torch.empty((n, h, w, cout), device=x.device, dtype=_f32)
    nws = L.load().naws_winograd_f32x3_workspace_floats(n, h, w, cin, cout, dilation)
    ws = torch.empty((nws,), device=x.device, dtype=_f32)
    L.call('naws_conv3x3_winograd_nhwc_f32x3_fwd', x.data_ptr(), u3.data_ptr(), _ptr(bias), n, h,
           w, cin, cout, dilation, int(relu), ws.data_ptr(), y.data_ptr(), _stream())
    return y


def emulate_exchange(src, dst, nbytes, cus, gbytes_per_sec):
    """MEASUREMENT AID (bench.py --emulate-exchange): `cus` workgroups copy nbytes from src to
    dst at the given aggregate pace on the current stream - see naws_emulate_exchange."""
    _chk(src, 'src'); _chk(dst, 'dst')
    nbytes = int(nbytes) // 16 * 16
    if nbytes > src.numel() * src.element_size() or nbytes > dst.numel() * dst.element_size():
        raise ValueError('emulate_exchange: nbytes exceeds a buffer')
    L.call('naws_emulate_exchange', src.data_ptr(), dst.data_ptr(), nbytes, int(cus),
           float(gbytes_per_sec), _stream())


def amax_word(x, out=None):
    """Bit pattern (int32 [1]) of max|x| over a contiguous fp32 tensor."""
    _chk(x, 'x')
    o = out if out is not None else torch.empty((1,), device=x.device, dtype=torch.int32)
    L.call('naws_amax_f32', x.data_ptr(), x.numel(), o.data_ptr(), _stream())
    return o


def conv3x3_nhwc_f16x2(x, w2, bias, relu=True, out=None, amax_in=None, in_mul=1.0, in_add=0.0,
                       amax_out=None, pool2=False, amax_out_zeroed=False, dilation=1):
    """3x3 / pad 1 conv (shallow layers) with w2 = split_f16x2(packed weight viewed
    [Cout, 9*Cin]); amax_in: int32 [1] bit pattern of a bound b, max|x| <= b * in_mul + in_add
    (measured here when None); amax_out: int32 [1] receiving the bit pattern of max|y|;
    pool2: return MaxPool 2x2 / stride 2 of the layer's output ([n, h//2, w//2, cout]) instead."""
    _chk(x, 'x')
    n, h, w, cin = x.shape
    cout = w2.planes.shape[-2]
    if w2.planes.dtype != torch.float16 or tuple(w2.planes.shape[:2]) != (2, 9 * cin // 16):
        raise TypeError('w2 must hold the f16 planes [2, 9*Cin/16, Cout, 16] of the packed weight')
    if amax_in is None:
        amax_in, in_mul, in_add = amax_word(x), 1.0, 0.0
    shape = (n, h // 2, w // 2, cout) if pool2 else (n, h, w, cout)
    y = out if out is not None else torch.empty(shape, device=x.device, dtype=_f32)
    L.call('naws_conv3x3_nhwc_f16x2_fwd', x.data_ptr(), w2.planes.data_ptr(),
           w2.inv_scale.data_ptr(), _ptr(bias), n, h, w, cin, cout, int(dilation), int(relu),
           y.data_ptr(),
           amax_in.data_ptr(), float(in_mul), float(in_add), _ptr(amax_out),
           int(bool(amax_out_zeroed)), int(pool2), _stream())
    return y


def winograd_weight_columns(u):
    """The transformed weight U [16 = (row i, column j)][Cout][Cin] regrouped for the
    frequency-column batch GEMM and split: F16x2 with planes [2, 4 (j), 4 Cin/16, Cout, 16] -
    row (j, cout) = the four row blocks i of column j side by side, one scale per row."""
    _chk(u, 'u')
    sixteen, cout, cin = u.shape
    assert sixteen == 16
    cols = u.view(4, 4, cout, cin).permute(1, 2, 0, 3).reshape(4, cout, 4 * cin).contiguous()
    return split_f16x2(cols)


def conv3x3_winograd_nhwc_f16x2(x, u2, bias, dilation=1, relu=True, out=None, amax_in=None,
                                amax_out=None):
    """Winograd F(2x2,3x3) with fp16x2 GEMMs; u2 = split_f16x2(winograd_weight_transform(w))
    (F16x2: planes [2, 16, Cin/16, Cout, 16], scales [2, 16, Cout]) - or F(4x4,3x3) when u2 =
    split_f16x2(winograd4_weight_transform(w)) (planes [2, 36, Cin/16, Cout, 16]).  amax_in / amax_out: optional
    one-element int32 tensors (bit pattern of an upper bound of max|x| / receives that of max|y|)."""
    _chk(x, 'x')
    n, h, w, cin = x.shape
    cout = u2.planes.shape[-2]
    col = tuple(u2.planes.shape[:3]) == (2, 4, 4 * cin // 16)     # winograd_weight_columns
    if u2.planes.dtype != torch.float16 or not (
            col or tuple(u2.planes.shape[:3]) in ((2, 16, cin // 16), (2, 36, cin // 16))):
        raise TypeError('u2 must hold the f16 planes [2, 16, Cin/16, Cout, 16] of U (or the '
                        'column form [2, 4, 4 Cin/16, Cout, 16])')
    y = out if out is not None else torch.empty((n, h, w, cout), device=x.device, dtype=_f32)
    if tuple(u2.planes.shape[:3]) == (2, 36, cin // 16):          # winograd4_weight_transform
        nws = L.load().naws_winograd4_f16x2_workspace_floats(n, h, w, cin, cout, dilation)
        ws = torch.empty((nws,), device=x.device, dtype=_f32)
        if amax_in is None:
            amax_in = amax_word(x)
        L.call('naws_conv3x3_winograd4_nhwc_f16x2_fwd', x.data_ptr(), u2.planes.data_ptr(),
               u2.inv_scale.data_ptr(), _ptr(bias), n, h, w, cin, cout, dilation, int(relu),
               ws.data_ptr(), y.data_ptr(), _ptr(amax_in), _ptr(amax_out), _stream())
        return y
    nws = L.load().naws_winograd_f16x2_workspace_floats(n, h, w, cin, cout, dilation)
    ws = torch.empty((nws,), device=x.device, dtype=_f32)
    if col and not hasattr(L.load(), 'naws_conv3x3_winograd_nhwc_f16x2_col_fwd'):
        raise RuntimeError('the frequency-column Winograd form lives in the A/B build only '
                           '(make -C na-fwebsod_amd/csrc AB=1; NAWS_LIB=.../libnaws_hip_ab.so): it '
                           'measured slower than the 16-plane form, profiles/r04_wino_column_pmc.md')
    if col and amax_in is None:
        amax_in = amax_word(x)
    L.call('naws_conv3x3_winograd_nhwc_f16x2_col_fwd' if col else
           'naws_conv3x3_winograd_nhwc_f16x2_fwd', x.data_ptr(), u2.planes.data_ptr(),
           u2.inv_scale.data_ptr(), _ptr(bias), n, h, w, cin, cout, dilation, int(relu),
           ws.data_ptr(), y.data_ptr(), _ptr(amax_in), _ptr(amax_out), _stream())
    return y


SOFT_NMS_MAX = 5111         # naws_soft_nms_fwd keeps a class's list in LDS


def soft_nms_per_class(dets, counts, sigma, overlap_thresh, score_thresh, method):
    """dets fp32 [C, n_max, 5], counts int32 [C] (device) -> (out_dets [C, n_max, 5], keep int32
    [C, n_max], out_counts int32 [C]): cython_nms.soft_nms of every class in one launch, results in
    the reference's output order."""
    _chk(dets, 'dets')
    c, n_max, five = dets.shape
    if five != 5 or counts.dtype != torch.int32 or counts.numel() != c:
        raise L.NawsError('naws_soft_nms_fwd', L.ERR_SHAPE)
    out = torch.empty_like(dets)
    keep = torch.empty((c, n_max), device=dets.device, dtype=torch.int32)
    oc = torch.empty((c,), device=dets.device, dtype=torch.int32)
    L.call('naws_soft_nms_fwd', dets.data_ptr(), counts.data_ptr(), c, n_max, float(sigma),
           float(overlap_thresh), float(score_thresh), int(method), out.data_ptr(), keep.data_ptr(),
           oc.data_ptr(), _stream())
    return out, keep, oc


def nms_per_class(boxes, scores, score_thresh, nms_thresh):
    """Per-class greedy NMS of one image on the GPU (cython_nms.pyx `nms` semantics).
    boxes [R,4] (or [R,4*C] class-tiled, the reference's pred_boxes), scores [R,C] (fg classes).
    -> keep [C,R] bool: box r survives for class c (score > score_thresh and not suppressed).
    Visiting order per class = stable descending score."""
    r, c = scores.shape
    dev = scores.device
    if r == 0:
        return torch.zeros((c, 0), dtype=torch.bool, device=dev)
    st = scores.t().contiguous()                                  # [C,R]
    valid = st > score_thresh
    key = torch.where(valid, st, torch.full_like(st, -float('inf')))
    order = torch.sort(key, dim=1, descending=True, stable=True).indices     # candidates first
    counts = valid.sum(dim=1).to(torch.int32).contiguous()
    if boxes.shape[1] == 4:
        sb = boxes[order]                                         # [C,R,4]
    else:
        b3 = boxes.view(r, -1, 4)
        off = b3.shape[1] - c                                     # skip the background column(s)
        sb = b3[order, (torch.arange(c, device=dev) + off)[:, None]]
    sb = sb.to(_f32).contiguous()
    ws = torch.empty((L.load().naws_nms_workspace_bytes(c, r) // 8,), dtype=torch.int64, device=dev)
    keep_sorted = torch.empty((c, r), dtype=torch.int32, device=dev)
    L.call('naws_nms_sorted_fwd', sb.data_ptr(), counts.data_ptr(), c, r, float(nms_thresh),
           ws.data_ptr(), keep_sorted.data_ptr(), _stream())
    keep = torch.zeros((c, r), dtype=torch.bool, device=dev)
    keep.scatter_(1, order, keep_sorted.bool())
    return keep


def prep_image(im_u8, out, im_scale, flip=False, crop=None, means=(0.0, 0.0, 0.0),
               stds=(1.0, 1.0, 1.0), distort=None):
    """Loader image preparation on the GPU.  im_u8: uint8 [H,W,3] BGR device tensor;
    crop = (y0, x0, y1, x1) inclusive on the (flipped) image; `out` = this image's [3,Hp,Wp]
    slice of the NCHW batch blob (zero-filled by the caller).  Returns (out_h, out_w)."""
    import ctypes
    if im_u8.dtype != torch.uint8 or im_u8.dim() != 3 or im_u8.shape[2] != 3 or not im_u8.is_contiguous():
        raise TypeError('im_u8 must be a contiguous uint8 [H,W,3] device tensor')
    h, w = int(im_u8.shape[0]), int(im_u8.shape[1])
    y0, x0, y1, x1 = crop if crop is not None else (0, 0, h - 1, w - 1)
    ch, cw = y1 - y0 + 1, x1 - x0 + 1
    oh, ow = int(np.round(ch * im_scale)), int(np.round(cw * im_scale))   # cvRound: half to even
    if out.dim() != 3 or out.shape[0] != 3 or out.shape[1] < oh or out.shape[2] < ow or \
            out.stride(2) != 1 or out.dtype != _f32:
        raise TypeError('out must be a [3,Hp,Wp] fp32 view with Hp >= %d, Wp >= %d' % (oh, ow))
    m = (ctypes.c_float * 3)(*[float(v) for v in means])
    sd = (ctypes.c_float * 3)(*[float(v) for v in stds])
    L.call('naws_prep_image_fwd', im_u8.data_ptr(), h, w, int(bool(flip)), int(y0), int(x0), ch, cw,
           ctypes.cast(m, ctypes.c_void_p), ctypes.cast(sd, ctypes.c_void_p), float(im_scale),
           int(distort is not None), float(distort[0]) if distort else 1.0,
           float(distort[1]) if distort else 1.0, oh, ow, out.stride(0), out.stride(1),
           out.data_ptr(), _stream())
    return oh, ow


def min_entropy_loss(x, l):
    _chk(x, 'X'); _chk(l, 'L')
    if x.dim() != 2 or l.dim() != 2 or l.shape[0] != 1 or l.shape[1] != x.shape[1]:
        raise L.NawsError('naws_min_entropy_loss_fwd', L.ERR_SHAPE)    # CAFFE_ENFORCE sites :11-13,30
    y = torch.empty((1,), device=x.device, dtype=_f32)
    L.call('naws_min_entropy_loss_fwd', x.data_ptr(), l.data_ptr(), x.shape[0], x.shape[1],
           y.data_ptr(), _stream())
    return y


def min_entropy_loss_grad(x, l, dy):
    _chk(x, 'X'); _chk(l, 'L'); _chk(dy, 'dY')
    dx = torch.empty_like(x)
    L.call('naws_min_entropy_loss_bwd', x.data_ptr(), l.data_ptr(), dy.data_ptr(), x.shape[0],
           x.shape[1], dx.data_ptr(), _stream())
    return dx


def to_bf16_slab(x, transpose=False, out=None):
    """fp32 [rows, cols] or [b, rows, cols] -> bf16 [(b,) K/16, outer, 16] (K-slab-major, K
    rounded up to 64 and zero-filled): the operand layout of gemm_bf16_slab_nt."""
    batched = x.dim() == 3
    x2 = x[0] if batched else x
    if not x.is_cuda or x.dtype != _f32 or x2.stride(1) != 1:
        raise TypeError('x must be a HIP fp32 tensor with a contiguous last dim')
    batch = x.shape[0] if batched else 1
    rows, cols = x2.shape
    outer, k = (cols, rows) if transpose else (rows, cols)
    kpad = (k + 63) // 64 * 64
    shape = (batch, kpad // 16, outer, 16) if batched else (kpad // 16, outer, 16)
    p = out if out is not None else torch.empty(shape, device=x.device, dtype=torch.bfloat16)
    L.call('naws_to_bf16_slab', x.data_ptr(), batch, rows, cols, x2.stride(0),
           (x.stride(0) if batched else 0), int(transpose), kpad, p.data_ptr(), _stream())
    return p


def gemm_bf16_slab_nt(a, b, out=None, epilogue=L.EPI_NONE, bias=None, aux=None, alpha=1.0,
                      drop_ratio=0.0, seed=0, accumulate=False):
    """C[M,N] (+)= A B^T for bf16 slab operands a [(b,)K/16,M,16], b [(b,)K/16,N,16]
    (to_bf16_slab); bf16 MFMA, fp32 accumulate.  Row-sliced views (a[..., r0:r1, :]) are fine."""
    batched = a.dim() == 4
    for t in (a, b):
        if (not t.is_cuda or t.dtype != torch.bfloat16 or t.shape[-1] != 16 or t.stride(-1) != 1
                or t.stride(-2) != 16):
            raise TypeError('operands must be bf16 slab tensors [..., K/16, rows, 16]')
    batch = a.shape[0] if batched else 1
    mm, k = a.shape[-2], a.shape[-3] * 16
    nn, kb = b.shape[-2], b.shape[-3] * 16
    if k != kb:
        raise L.NawsError('naws_gemm_bf16_slab_nt', L.ERR_SHAPE)
    if out is None:
        out = torch.empty(((batch, mm, nn) if batched else (mm, nn)), device=a.device, dtype=_f32)
    c2 = out[0] if batched else out
    sbias = bias.stride(0) if (bias is not None and bias.dim() == 2) else 0
    L.call('naws_gemm_bf16_slab_nt', mm, nn, k, a.data_ptr(), a.stride(-3), b.data_ptr(),
           b.stride(-3), out.data_ptr(), c2.stride(0), batch, (a.stride(0) if batched else 0),
           (b.stride(0) if batched else 0), (out.stride(0) if batched else 0), epilogue,
           _ptr(bias), sbias, _ptr(aux), (aux.stride(-2) if aux is not None else 0), float(alpha),
           float(drop_ratio), int(seed) & 0xFFFFFFFFFFFFFFFF, int(accumulate), _stream())
    return out


def planes_to_dense(p):
    """[3, (b,) K/16, outer, 16] planes -> float64 [(b,) outer, K] (test / debug helper)."""
    s = p[0].double() + p[1].double() + p[2].double()
    s = s.transpose(-3, -2)
    return s.reshape(*s.shape[:-2], -1)


def gemm_f32x3_nt(a3, b3, out=None, epilogue=L.EPI_NONE, bias=None, aux=None, alpha=1.0,
                  drop_ratio=0.0, seed=0, accumulate=False):
    """C[M,N] (+)= A B^T from split planes a3 [3,(b,)K/16,M,16], b3 [3,(b,)K/16,N,16]
    (split_bf16x3); fp32-accurate, runs on the bf16 MFMA.  Row-sliced plane views
    (a3[..., r0:r1, :]) are fine."""
    batched = a3.dim() == 5
    for t in (a3, b3):
        if (not t.is_cuda or t.dtype != torch.bfloat16 or t.shape[0] != 3 or t.shape[-1] != 16
                or t.stride(-1) != 1 or t.stride(-2) != 16):
            raise TypeError('operands must be bf16 split planes [3, ..., K/16, rows, 16]')
    batch = a3.shape[1] if batched else 1
    mm, k = a3.shape[-2], a3.shape[-3] * 16
    nn, kb = b3.shape[-2], b3.shape[-3] * 16
    if k != kb:
        raise L.NawsError('naws_gemm_f32x3_nt', L.ERR_SHAPE)
    if out is None:
        out = torch.empty(((batch, mm, nn) if batched else (mm, nn)), device=a3.device, dtype=_f32)
    c2 = out[0] if batched else out
    sbias = bias.stride(0) if (bias is not None and bias.dim() == 2) else 0
    L.call('naws_gemm_f32x3_nt', mm, nn, k, a3.data_ptr(), a3.stride(-3), a3.stride(0),
           b3.data_ptr(), b3.stride(-3), b3.stride(0), out.data_ptr(), c2.stride(0), batch,
           (a3.stride(1) if batched else 0), (b3.stride(1) if batched else 0),
           (out.stride(0) if batched else 0), epilogue, _ptr(bias), sbias, _ptr(aux),
           (aux.stride(-2) if aux is not None else 0), float(alpha), float(drop_ratio),
           int(seed) & 0xFFFFFFFFFFFFFFFF, int(accumulate), _stream())
    return out


class F16x2(object):
    """An fp32 matrix split for the fp16x2 GEMM: `planes` [2, (b,) K/32*2, outer, 16] f16
    (hi, lo of x * scale_row) and `scales` [2, (b,) outer] fp32 ([1] = 1/scale_row, [0] scratch)."""
    __slots__ = ('planes', 'scales')

    def __init__(self, planes, scales):
        self.planes, self.scales = planes, scales

    def rows(self, r0, r1):
        """The operand restricted to outer indices r0..r1 (views, no copy)."""
        return F16x2(self.planes[..., r0:r1, :], self.scales[..., r0:r1])

    def batches(self, nb):
        return F16x2(self.planes[:, :nb], self.scales[:, :nb])

    @property
    def inv_scale(self):
        return self.scales[1]


def split_f16x2(x, transpose=False, out=None, rowmul=None):
    """fp32 [rows, cols] or [b, rows, cols] (last dim contiguous) -> F16x2 with planes
    [2, (b,) K/16, outer, 16] (K rounded up to 32, zero-filled) scaled per outer index by a
    power of two that puts the row maximum in [2^14, 2^15):  x * s = hi + lo to 22+ bits.
    rowmul (fp32 [rows], optional): split diag(rowmul) x instead."""
    batched = x.dim() == 3
    x2 = x[0] if batched else x
    if not x.is_cuda or x.dtype != _f32 or x2.stride(1) != 1:
        raise TypeError('x must be a HIP fp32 tensor with a contiguous last dim')
    batch = x.shape[0] if batched else 1
    rows, cols = x2.shape
    outer, k = (cols, rows) if transpose else (rows, cols)
    kpad = (k + 31) // 32 * 32
    if out is None:
        shape = (2, batch, kpad // 16, outer, 16) if batched else (2, kpad // 16, outer, 16)
        out = F16x2(torch.empty(shape, device=x.device, dtype=torch.float16),
                    torch.empty((2, batch, outer) if batched else (2, outer), device=x.device,
                                dtype=_f32))
    if rowmul is not None and (rowmul.numel() != rows or rowmul.dtype != _f32
                               or not rowmul.is_contiguous()):
        raise TypeError('rowmul must be a contiguous fp32 vector with one entry per source row')
    L.call('naws_split_f16x2_kscaled', x.data_ptr(), batch, rows, cols, x2.stride(0),
           (x.stride(0) if batched else 0), int(transpose), kpad, out.planes.data_ptr(),
           out.scales.data_ptr(), _ptr(rowmul), _stream())
    return out


def gemm_f32_f16x2_nt(a, b, out=None, epilogue=L.EPI_NONE, bias=None, aux=None, alpha=1.0,
                      drop_ratio=0.0, seed=0, accumulate=False, rowmax=None, rowmax_seg=0,
                      colmax=None, colmax_rowmul=None):
    """C[M,N] (+)= A B^T from F16x2 operands (split_f16x2); fp32 in / fp32 accumulate / fp32 out
    on the f16 MFMA with three products per K-slab.  Row-sliced operands (`.rows`) are fine.
    rowmax / rowmax_seg / colmax / colmax_rowmul: as `gemm`."""
    a3, b3 = a.planes, b.planes
    batched = a3.dim() == 5
    for t in (a3, b3):
        if (not t.is_cuda or t.dtype != torch.float16 or t.shape[0] != 2 or t.shape[-1] != 16
                or t.stride(-1) != 1 or t.stride(-2) != 16):
            raise TypeError('operands must be f16 split planes [2, ..., K/16, rows, 16]')
    sa, sb = a.inv_scale, b.inv_scale
    if sa.stride(-1) != 1 or sb.stride(-1) != 1:
        raise TypeError('scale vectors must be contiguous')
    batch = a3.shape[1] if batched else 1
    mm, k = a3.shape[-2], a3.shape[-3] * 16
    nn, kb = b3.shape[-2], b3.shape[-3] * 16
    if k != kb:
        raise L.NawsError('naws_gemm_f32_f16x2_nt', L.ERR_SHAPE)
    if out is None:
        out = torch.empty(((batch, mm, nn) if batched else (mm, nn)), device=a3.device, dtype=_f32)
    c2 = out[0] if batched else out
    sbias = bias.stride(0) if (bias is not None and bias.dim() == 2) else 0
    L.call('naws_gemm_f32_f16x2_nt_amax', mm, nn, k, a3.data_ptr(), a3.stride(-3), a3.stride(0),
           sa.data_ptr(), b3.data_ptr(), b3.stride(-3), b3.stride(0), sb.data_ptr(),
           out.data_ptr(), c2.stride(0), batch,
           (a3.stride(1) if batched else 0), (b3.stride(1) if batched else 0),
           (out.stride(0) if batched else 0), (sa.stride(0) if batched else 0),
           (sb.stride(0) if batched else 0), epilogue, _ptr(bias), sbias, _ptr(aux),
           (aux.stride(-2) if aux is not None else 0), float(alpha), float(drop_ratio),
           int(seed) & 0xFFFFFFFFFFFFFFFF, int(accumulate),
           *_amax_args(rowmax, rowmax_seg, colmax, colmax_rowmul), _stream())
    return out


def gemm_f32_f16x2_nt_cols(a, b, out, col0, width, epilogue=L.EPI_NONE, bias=None, drop_ratio=0.0,
                           seed=0, rowmax=None, rowmax_seg=0, colmax=None):
    """Columns [col0, col0 + n) of the `width`-wide product A B_all^T, with b = the rows
    col0 .. col0 + n of B_all's operand (`.rows(col0, col0 + n)`), `out` = that column range of the
    [M, width] result (a row-strided view), bias / colmax = the range's entries, rowmax = the words
    of the range's rowmax segment.  Element (m, c) draws the Dropout counter m * width + col0 + c,
    as the full-width gemm_f32_f16x2_nt does: the pieces of one activation are bit-identical to
    the single launch (naws_gemm_f32_f16x2_nt_cols)."""
    a3, b3 = a.planes, b.planes
    for t in (a3, b3):
        if (not t.is_cuda or t.dtype != torch.float16 or t.dim() != 4 or t.shape[0] != 2
                or t.shape[-1] != 16 or t.stride(-1) != 1 or t.stride(-2) != 16):
            raise TypeError('operands must be unbatched f16 split planes [2, K/16, rows, 16]')
    sa, sb = a.inv_scale, b.inv_scale
    mm, k = a3.shape[-2], a3.shape[-3] * 16
    nn = b3.shape[-2]
    if k != b3.shape[-3] * 16 or out.shape != (mm, nn) or out.stride(1) != 1:
        raise L.NawsError('naws_gemm_f32_f16x2_nt_cols', L.ERR_SHAPE)
    L.call('naws_gemm_f32_f16x2_nt_cols', mm, nn, k, a3.data_ptr(), a3.stride(-3), a3.stride(0),
           sa.data_ptr(), b3.data_ptr(), b3.stride(-3), b3.stride(0), sb.data_ptr(), out.data_ptr(),
           out.stride(0), epilogue, _ptr(bias), float(drop_ratio), int(seed) & 0xFFFFFFFFFFFFFFFF,
           *_amax_args(rowmax, rowmax_seg, colmax, None)[:3], int(width), int(col0), _stream())
    return out


def gemm_f32_f16x2_nt_xk(a, x, out=None, ncols=None, scale_x=None):
    """C[M, N] = sum_k A[k][m] X[k][n]: a = F16x2 with planes [2, K/16, M, 16] (a transposing split:
    K = the source's rows, zero beyond x's row count), x = F16x2 with planes [2, N/16, R, 16] - the
    FORWARD operand of an [R, N] matrix (roi_pool_f_f16x2 / split_f16x2), read K(=row)-wise by the
    kernel's transposing LDS reads, so no transposed copy of x is made.  x's per-row scales are
    NOT applied (the caller folds them into a, as the fc6 wgrad does with rowmul); scale_x:
    optional per-column factors.  ncols: (c0, c1) restricts to x's columns c0..c1 (multiples of 16)."""
    a3, x3 = a.planes, x.planes
    for t in (a3, x3):
        if (not t.is_cuda or t.dtype != torch.float16 or t.dim() != 4 or t.shape[0] != 2
                or t.shape[-1] != 16 or t.stride(-1) != 1 or t.stride(-2) != 16):
            raise TypeError('operands must be unbatched f16 split planes [2, K/16, rows, 16]')
    mm, k = a3.shape[-2], a3.shape[-3] * 16
    r, n_all = x3.shape[-2], x3.shape[-3] * 16
    c0, c1 = (0, n_all) if ncols is None else ncols
    if c0 % 16 or c1 % 16 or not 0 <= c0 < c1 <= n_all or r > k:
        raise L.NawsError('naws_gemm_f32_f16x2_nt_xk', L.ERR_SHAPE)
    nn = c1 - c0
    if out is None:
        out = torch.empty((mm, nn), device=a3.device, dtype=_f32)
    xs = x3[:, c0 // 16:]
    L.call('naws_gemm_f32_f16x2_nt_xk', mm, nn, k, a3.data_ptr(), a3.stride(-3), a3.stride(0),
           a.inv_scale.data_ptr(), xs.data_ptr(), x3.stride(-3), x3.stride(0), r,
           _ptr(scale_x), out.data_ptr(), out.stride(0), _stream())
    return out


def gemm_f32_f16x2_nt_xk_sgd(a, x, param, momentum_buf, lr, lr_mult, weight_decay, momentum, nesterov,
                             gpu_num, iter_count, wplanes, bound, rowmax, inv_scale, overflow,
                             overflow_tag, ncols=None, rows=None):
    """gemm_f32_f16x2_nt_xk whose epilogue applies the ACM SGD update to `param` (fp32 [M_all, N_all]
    row-major view of the arena; `momentum_buf` likewise) instead of storing the gradient, and
    writes param's own operand planes `wplanes` (F16x2 planes [2, N_all/16, M_all, 16]).  rows =
    (r0, r1): a's rows are rows r0..r1 of param; ncols as in gemm_f32_f16x2_nt_xk.  bound / rowmax
    (int32 [M_all]) / inv_scale (fp32 [M_all]) / overflow as acm_sgd_update_f16x2."""
    a3, x3 = a.planes, x.planes
    for t in (a3, x3, wplanes):
        if (not t.is_cuda or t.dtype != torch.float16 or t.dim() != 4 or t.shape[0] != 2
                or t.shape[-1] != 16 or t.stride(-1) != 1 or t.stride(-2) != 16):
            raise TypeError('operands must be unbatched f16 split planes [2, K/16, rows, 16]')
    mm, k = a3.shape[-2], a3.shape[-3] * 16
    r, n_all = x3.shape[-2], x3.shape[-3] * 16
    c0, c1 = (0, n_all) if ncols is None else ncols
    r0, r1 = (0, mm) if rows is None else rows
    if (c0 % 16 or c1 % 16 or not 0 <= c0 < c1 <= n_all or r > k or r1 - r0 != mm
            or param.dim() != 2 or param.shape[1] != n_all or param.stride(1) != 1
            or momentum_buf.shape != param.shape or momentum_buf.stride() != param.stride()
            or wplanes.shape[1] * 16 != n_all or wplanes.shape[2] != param.shape[0]
            or not 0 <= r0 < r1 <= param.shape[0]):
        raise L.NawsError('naws_gemm_f32_f16x2_nt_xk_sgd', L.ERR_SHAPE)
    for t, dt in ((bound, torch.int32), (rowmax, torch.int32), (inv_scale, _f32)):
        if t.dtype != dt or t.numel() != param.shape[0] or not t.is_contiguous():
            raise TypeError('bound / rowmax / inv_scale: contiguous [rows] int32 / int32 / fp32')
    xs = x3[:, c0 // 16:]
    ws = wplanes[:, c0 // 16:, r0:]
    L.call('naws_gemm_f32_f16x2_nt_xk_sgd', mm, c1 - c0, k, a3.data_ptr(), a3.stride(-3), a3.stride(0),
           a.inv_scale.data_ptr(), xs.data_ptr(), x3.stride(-3), x3.stride(0), r, None,
           param[r0:r1, c0:c1].data_ptr(), momentum_buf[r0:r1, c0:c1].data_ptr(), param.stride(0),
           lr.data_ptr(), float(lr_mult), float(weight_decay), float(momentum), int(nesterov),
           int(gpu_num), int(iter_count), ws.data_ptr(), wplanes.stride(0), wplanes.shape[2],
           bound[r0:].data_ptr(), rowmax[r0:].data_ptr(), inv_scale[r0:].data_ptr(),
           overflow.data_ptr(), int(overflow_tag), _stream())


def gemm_bf16_slab_nt_sgd(a, b, param, momentum_buf, lr, lr_mult, weight_decay, momentum, nesterov,
                          gpu_num, iter_count, wplane, rows=None):
    """gemm_bf16_slab_nt whose epilogue applies the ACM SGD update to rows r0..r1 of `param` (fp32
    [M_all, N] row-major view of the arena; `momentum_buf` likewise) instead of storing the
    gradient a b^T, and rounds the updated rows into param's own bf16 operand plane `wplane`
    ([N/16, M_all, 16], to_bf16_slab(param))."""
    for t in (a, b, wplane):
        if (not t.is_cuda or t.dtype != torch.bfloat16 or t.dim() != 3 or t.shape[-1] != 16
                or t.stride(-1) != 1 or t.stride(-2) != 16):
            raise TypeError('operands must be unbatched bf16 slab tensors [K/16, rows, 16]')
    mm, k = a.shape[-2], a.shape[-3] * 16
    n = b.shape[-2]
    r0, r1 = (0, mm) if rows is None else rows
    if (b.shape[-3] * 16 != k or r1 - r0 != mm or param.dim() != 2 or param.shape[1] != n
            or param.stride(1) != 1 or momentum_buf.shape != param.shape
            or momentum_buf.stride() != param.stride() or wplane.shape[0] * 16 < n
            or wplane.shape[1] != param.shape[0] or not wplane.is_contiguous()
            or not 0 <= r0 < r1 <= param.shape[0]):
        raise L.NawsError('naws_gemm_bf16_slab_nt_sgd', L.ERR_SHAPE)
    L.call('naws_gemm_bf16_slab_nt_sgd', mm, n, k, a.data_ptr(), a.stride(-3), b.data_ptr(),
           b.stride(-3), momentum_buf[r0:r1].data_ptr(), param[r0:r1].data_ptr(), param.stride(0),
           lr.data_ptr(), float(lr_mult), float(weight_decay), float(momentum), int(nesterov),
           int(gpu_num), int(iter_count), wplane[:, r0:].data_ptr(), wplane.shape[1], _stream())


def amax_scales(batch, outer, device):
    """Zeroed scale block [2, (batch,) outer] of an F16x2 whose maxima a GEMM epilogue will report:
    [0] (viewed as int32 bit patterns: `amax_words`) is the rowmax / colmax accumulator, [1]
    receives 1/scale from split_f16x2_dual."""
    return torch.zeros((2, batch, outer) if batch else (2, outer), device=device, dtype=_f32)


def amax_words(scales):
    return scales[0].view(torch.int32)


def split_f16x2_dual(x, scales_n=None, scales_t=None, rowmul=None, out_n=None, out_t=None):
    """One pass over x (fp32 [rows, cols] or [b, rows, cols], last dim contiguous) -> the
    row-scaled planes (F16x2, outer = rows, K = cols) when `scales_n` is given and / or the
    column-scaled transposed planes of diag(rowmul) x (outer = cols, K = rows) when `scales_t` is
    given.  scales_* are `amax_scales` blocks whose [0] already holds the maxima (bit patterns)
    reported by the GEMM that produced x.  Returns (normal or None, transposed or None)."""
    batched = x.dim() == 3
    x2 = x[0] if batched else x
    if not x.is_cuda or x.dtype != _f32 or x2.stride(1) != 1:
        raise TypeError('x must be a HIP fp32 tensor with a contiguous last dim')
    batch = x.shape[0] if batched else 1
    rows, cols = x2.shape
    kn, kt = (cols + 31) // 32 * 32, (rows + 31) // 32 * 32
    bs = (batch,) if batched else ()
    pn = pt = None
    if scales_n is not None:
        pn = out_n if out_n is not None else F16x2(
            torch.empty((2, *bs, kn // 16, rows, 16), device=x.device, dtype=torch.float16), scales_n)
    if scales_t is not None:
        pt = out_t if out_t is not None else F16x2(
            torch.empty((2, *bs, kt // 16, cols, 16), device=x.device, dtype=torch.float16), scales_t)
    for o, sc, shape in ((pn, scales_n, (2, *bs, kn // 16, rows, 16)),
                         (pt, scales_t, (2, *bs, kt // 16, cols, 16))):
        if o is not None and (tuple(o.planes.shape) != shape or not o.planes.is_contiguous()
                              or o.scales.data_ptr() != sc.data_ptr()):
            raise TypeError('out planes must be contiguous %s with the given scales block' % (shape,))
    for sc, outer in ((scales_n, rows), (scales_t, cols)):
        if sc is not None and (tuple(sc.shape) != (2, *bs, outer) or sc.dtype != _f32
                               or not sc.is_contiguous()):
            raise TypeError('scales must be contiguous fp32 [2, (batch,) outer] blocks')
    if rowmul is not None and (rowmul.numel() != rows or rowmul.dtype != _f32
                               or not rowmul.is_contiguous()):
        raise TypeError('rowmul must be a contiguous fp32 vector with one entry per source row')
    L.call('naws_split_f16x2_dual', x.data_ptr(), batch, rows, cols, x2.stride(0),
           (x.stride(0) if batched else 0), _ptr(scales_n), _ptr(scales_t), _ptr(rowmul),
           _ptr(pn.planes if pn else None), _ptr(scales_n), kn,
           _ptr(pt.planes if pt else None), _ptr(scales_t), kt, _stream())
    return pn, pt


def dropout_mask(seed, ratio, n, device):
    m = torch.empty((n,), device=device, dtype=_f32)
    L.call('naws_dropout_mask', int(seed) & 0xFFFFFFFFFFFFFFFF, float(ratio), n, m.data_ptr(),
           _stream())
    return m


def colsum(x, out=None, accumulate=False):
    assert x.dim() == 2 and x.stride(1) == 1
    m, n = x.shape
    y = out if out is not None else torch.empty((n,), device=x.device, dtype=_f32)
    L.call('naws_colsum_f32', x.data_ptr(), m, n, x.stride(0), y.data_ptr(), int(accumulate),
           _stream())
    return y


# ----------------------------------------------------------------------------
# WSDDN head tail
# ----------------------------------------------------------------------------
def wsddn_outputs(fc8c, fc8d, noisy_fc8c, noisy_fc8d, seg_off):
    """-> alpha_cls, alpha_det, rois_pred [nb,Rt,C], cls_prob [nb,nseg,C].  The four logit
    matrices may be column slices of one wider row-major buffer (same row stride)."""
    rt, c = fc8c.shape
    ld = fc8c.stride(0)
    for t in (fc8d, noisy_fc8c, noisy_fc8d):
        if t is not None and (t.stride(0) != ld or t.stride(1) != 1):
            raise ValueError('logit matrices must share one row stride')
    nseg = seg_off.numel() - 1
    nb = 2 if noisy_fc8c is not None else 1
    dev = fc8c.device
    ac = torch.empty((nb, rt, c), device=dev, dtype=_f32)
    ad = torch.empty_like(ac)
    rp = torch.empty_like(ac)
    cp = torch.empty((nb, nseg, c), device=dev, dtype=_f32)
    L.call('naws_wsddn_outputs_fwd', fc8c.data_ptr(), fc8d.data_ptr(), _ptr(noisy_fc8c),
           _ptr(noisy_fc8d), ld, seg_off.data_ptr(), nseg, rt, c, ac.data_ptr(), ad.data_ptr(),
           rp.data_ptr(), cp.data_ptr(), _stream())
    return ac, ad, rp, cp


def wsddn_outputs_grad(alpha_cls, alpha_det, rois_pred, cls_prob, d_cls_prob, seg_off, out=None,
                       col_offsets=None):
    """-> d_fc8c, d_fc8d, d_noisy_fc8c, d_noisy_fc8d as column slices of `out`
    ([Rt, 4C] at columns 0, C, 2C, 3C; allocated when None - or at `col_offsets` of a wider
    row-major `out`)."""
    nb, rt, c = alpha_cls.shape
    nseg = seg_off.numel() - 1
    if out is None:
        out = torch.empty((rt, 4 * c), device=alpha_cls.device, dtype=_f32)
    ld = out.stride(0)
    base = out.data_ptr()
    offs = col_offsets if col_offsets is not None else [c * i for i in range(4)]
    ptrs = [base + 4 * o for o in offs]
    L.call('naws_wsddn_outputs_bwd', alpha_cls.data_ptr(), alpha_det.data_ptr(),
           rois_pred.data_ptr(), cls_prob.data_ptr(), d_cls_prob.data_ptr(), seg_off.data_ptr(),
           nseg, rt, c, nb, ptrs[0], ptrs[1], ptrs[2] if nb == 2 else 0,
           ptrs[3] if nb == 2 else 0, ld, _stream())
    return out


def entropy_gate(rois, rois_pred, cls_prob, labels_oh, seg_off, max_seg_len):
    """-> class_weight, class_weight_noise, hatE_sum, hatE_sum_norm, each [nseg,C]."""
    _chk(rois, 'rois'); _chk(rois_pred, 'rois_pred'); _chk(cls_prob, 'cls_prob')
    _chk(labels_oh, 'labels_oh')
    rt, c = rois_pred.shape
    nseg = seg_off.numel() - 1
    dev = rois.device
    nws = L.load().naws_entropy_gate_workspace_floats(rt, c, nseg, max_seg_len)
    ws = torch.empty((max(nws, 1),), device=dev, dtype=_f32)
    # one buffer: [class_weight | class_weight_noise] is then the [2, nseg, C] weight operand of the
    # two branches' cross entropy as it stands (no stack)
    buf = torch.empty((4, nseg, c), device=dev, dtype=_f32)
    outs = [buf[i] for i in range(4)]
    L.call('naws_entropy_gate_fwd', rois.data_ptr(), rois_pred.data_ptr(), cls_prob.data_ptr(),
           labels_oh.data_ptr(), seg_off.data_ptr(), nseg, rt, c, max_seg_len, ws.data_ptr(),
           outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(),
           _stream())
    return tuple(outs)


def weighted_ce_shared(x, l, w, is_mean):
    """x, w: [nb, nseg, C] (nb branches), l: [nseg, C] shared by the branches -> losses [nb * nseg]
    (naws_weighted_ce_shared_fwd, N = 1 row per problem)."""
    _chk(x, 'X'); _chk(l, 'L'); _chk(w, 'W')
    nb, nseg, c = x.shape
    if l.shape != (nseg, c) or w.shape != x.shape:
        raise L.NawsError('naws_weighted_ce_shared_fwd', L.ERR_SHAPE)
    y = torch.empty((nb * nseg,), device=x.device, dtype=_f32)
    L.call('naws_weighted_ce_shared_fwd', x.data_ptr(), l.data_ptr(), w.data_ptr(), 1, c, int(is_mean),
           nb * nseg, nseg, y.data_ptr(), _stream())
    return y


def weighted_ce_shared_grad(x, l, w, is_mean, dy=None, dy_const=1.0):
    """Gradient of weighted_ce_shared w.r.t. x; dy None: the constant loss seed dy_const."""
    _chk(x, 'X'); _chk(l, 'L'); _chk(w, 'W')
    nb, nseg, c = x.shape
    if l.shape != (nseg, c) or w.shape != x.shape or (dy is not None and dy.numel() != nb * nseg):
        raise L.NawsError('naws_weighted_ce_shared_bwd', L.ERR_SHAPE)
    dx = torch.empty_like(x)
    L.call('naws_weighted_ce_shared_bwd', x.data_ptr(), l.data_ptr(), w.data_ptr(), _ptr(dy),
           float(dy_const), 1, c, int(is_mean), nb * nseg, nseg, dx.data_ptr(), _stream())
    return dx


def weighted_ce(x, l, w, is_mean, nprob=1):
    _chk(x, 'X'); _chk(l, 'L')
    if x.shape != l.shape or (w is not None and w.shape != x.shape):
        raise L.NawsError('naws_weighted_ce_fwd', L.ERR_SHAPE)
    c = x.shape[-1]
    n = x.numel() // (c * nprob)
    y = torch.empty((nprob,), device=x.device, dtype=_f32)
    L.call('naws_weighted_ce_fwd', x.data_ptr(), l.data_ptr(), _ptr(w), n, c, int(is_mean), nprob,
           y.data_ptr(), _stream())
    return y


def weighted_ce_grad(x, l, w, dy, is_mean, nprob=1):
    _chk(x, 'X'); _chk(l, 'L'); _chk(dy, 'dY')
    if x.shape != l.shape or (w is not None and w.shape != x.shape) or dy.numel() != nprob:
        raise L.NawsError('naws_weighted_ce_bwd', L.ERR_SHAPE)
    c = x.shape[-1]
    n = x.numel() // (c * nprob)
    dx = torch.empty_like(x)
    L.call('naws_weighted_ce_bwd', x.data_ptr(), l.data_ptr(), _ptr(w), dy.data_ptr(), n, c,
           int(is_mean), nprob, dx.data_ptr(), _stream())
    return dx


def acm_sgd_update(grad, momentum_buf, lr, param, acmgrad, seg_end, seg_lr_mult, seg_wd,
                   momentum, nesterov, iter_size, gpu_num, iter_count, rowmax=None, rm_table=None):
    """rowmax (zeroed int32 words) + rm_table (RowmaxTable): also report max|updated parameter|
    per matrix row of the table's arena regions."""
    total = param.numel()
    if rowmax is not None:
        L.call('naws_acm_sgd_update_rowmax', grad.data_ptr(), momentum_buf.data_ptr(),
               lr.data_ptr(), param.data_ptr(), _ptr(acmgrad), total, seg_end.data_ptr(),
               seg_lr_mult.data_ptr(), seg_wd.data_ptr(), seg_end.numel(), float(momentum),
               int(nesterov), int(iter_size), int(gpu_num), int(iter_count), rowmax.data_ptr(),
               rm_table.host.ctypes.data, rm_table.n, _stream())
        return
    L.call('naws_acm_sgd_update', grad.data_ptr(), momentum_buf.data_ptr(), lr.data_ptr(),
           param.data_ptr(), _ptr(acmgrad), total, seg_end.data_ptr(), seg_lr_mult.data_ptr(),
           seg_wd.data_ptr(), seg_end.numel(), float(momentum), int(nesterov), int(iter_size),
           int(gpu_num), int(iter_count), _stream())


class _SgdPlaneRegion(C.Structure):
    """naws_sgd_plane_region (include/naws.h)."""
    _fields_ = [('start', C.c_int64), ('rows', C.c_int32), ('cols', C.c_int32),
                ('rows_per_batch', C.c_int32), ('reserved', C.c_int32), ('planes', C.c_void_p),
                ('plane_stride', C.c_int64), ('bound', C.c_void_p), ('rowmax', C.c_void_p),
                ('inv_scale', C.c_void_p), ('colmax', C.c_void_p)]


class SgdPlaneRegions(object):
    """The weight matrices whose operand planes `acm_sgd_update_planes` writes itself:
    [(first arena element, rows, cols, rows_per_batch, planes, bound int32 [rows], rowmax int32
      [rows], inv_scale fp32 [rows][, colmax int32 [rows / rows_per_batch, cols]])], ascending;
    planes None = leave this matrix untouched.  planes: contiguous f16 [2, ...] (fp16x2), bf16
    [3, ...] (fp32x3) or bf16 [...] (bf16 plan); bound / rowmax / inv_scale only for fp16x2 (None
    otherwise)."""

    def __init__(self, regions, fmt=L.PLANES_F16X2):
        self.keep = regions                       # the tensors stay alive with the table
        self.fmt = fmt
        self.n = len(regions)
        self.host = (_SgdPlaneRegion * self.n)()
        want = {L.PLANES_F16X2: (torch.float16, 2), L.PLANES_BF16X3: (torch.bfloat16, 3),
                L.PLANES_BF16: (torch.bfloat16, None)}[fmt]
        for i, reg in enumerate(regions):
            start, rows, cols, rpb, planes, bound, rowmax, inv = reg[:8]
            colmax = reg[8] if len(reg) > 8 else None      # int32 [rows / rpb, cols] (optional)
            if planes is None:         # the matrix is left alone (updated by gemm_f32_f16x2_nt_xk_sgd)
                self.host[i] = _SgdPlaneRegion(int(start), int(rows), int(cols), int(rpb), 0, None, 0,
                                               None, None, None, None)
                continue
            if colmax is not None and (colmax.dtype != torch.int32 or not colmax.is_contiguous()
                                       or colmax.numel() != rows // rpb * cols):
                raise TypeError('colmax: contiguous int32 [rows / rows_per_batch, cols]')
            raw = None
            if isinstance(planes, tuple):
                # (tensor holding the planes of the larger matrix, first row of this block): a row
                # block of one batch item - rows < rows_per_batch (the sharded update)
                whole, first_row = planes
                if whole.dtype != want[0] or not whole.is_contiguous() or want[1] is None or \
                        whole.shape[0] != want[1] or rows >= rpb or first_row + rows > rpb:
                    raise TypeError('row-block planes: (contiguous planes of the whole matrix, first row)')
                raw = (whole.data_ptr() + first_row * 16 * whole.element_size(), whole.stride(0))
            elif planes.dtype != want[0] or not planes.is_contiguous() or \
                    (want[1] is not None and planes.shape[0] != want[1]) or \
                    planes.numel() != (want[1] or 1) * rows * cols:
                raise TypeError('planes must be the contiguous operand planes of a [rows, cols] matrix')
            if fmt == L.PLANES_F16X2:
                for t, dt in ((bound, torch.int32), (rowmax, torch.int32), (inv, _f32)):
                    if t.dtype != dt or t.numel() != rows or not t.is_contiguous():
                        raise TypeError('bound / rowmax / inv_scale: contiguous [rows] int32 / int32 / fp32')
            self.host[i] = _SgdPlaneRegion(int(start), int(rows), int(cols), int(rpb), 0,
                                           raw[0] if raw else planes.data_ptr(),
                                           raw[1] if raw else
                                           (planes.stride(0) if want[1] is not None else 0),
                                           _ptr(bound), _ptr(rowmax), _ptr(inv), _ptr(colmax))


def acm_sgd_update_f16x2(grad, momentum_buf, lr, param, seg_end, seg_lr_mult, seg_wd, momentum,
                         nesterov, gpu_num, iter_count, regions, overflow, overflow_tag):
    """ITER_SIZE 1.  The fused update that also emits the operand planes of `regions`
    (SgdPlaneRegions, fp16x2); `overflow` (int32 [1]) receives overflow_tag when a row outgrew
    its bound."""
    assert regions.fmt == L.PLANES_F16X2
    acm_sgd_update_planes(grad, momentum_buf, lr, param, seg_end, seg_lr_mult, seg_wd, momentum,
                          nesterov, gpu_num, iter_count, regions, overflow, overflow_tag)


def acm_sgd_update_planes(grad, momentum_buf, lr, param, seg_end, seg_lr_mult, seg_wd, momentum,
                          nesterov, gpu_num, iter_count, regions, overflow=None, overflow_tag=0):
    """The fused update writing the planes of `regions` in the regions' format (bf16x3 / bf16:
    exact / rounded planes, no scales, no overflow word)."""
    L.call('naws_acm_sgd_update_planes', regions.fmt, grad.data_ptr(), momentum_buf.data_ptr(),
           lr.data_ptr(), param.data_ptr(), param.numel(), seg_end.data_ptr(),
           seg_lr_mult.data_ptr(), seg_wd.data_ptr(), seg_end.numel(), float(momentum),
           int(nesterov), int(gpu_num), int(iter_count), C.addressof(regions.host), regions.n,
           _ptr(overflow), int(overflow_tag), _stream())


def split_f16x2_rows_if(x, rowmax, out, cond, cond_value):
    """Redo the row-scaled planes of x (fp32 [rows, cols] or [b, rows, cols]) into the F16x2 `out`
    from the maxima `rowmax` (int32 bit patterns, [b*rows]) - only if cond[0] == cond_value when
    the kernel runs."""
    batched = x.dim() == 3
    x2 = x[0] if batched else x
    batch = x.shape[0] if batched else 1
    rows, cols = x2.shape
    L.call('naws_split_f16x2_rows_if', x.data_ptr(), batch, rows, cols, x2.stride(0),
           (x.stride(0) if batched else 0), rowmax.data_ptr(), out.planes.data_ptr(),
           out.inv_scale.data_ptr(), (cols + 31) // 32 * 32, _ptr(cond), int(cond_value), _stream())
    return out


def split_f16x2_row_range_if(x, rowmax, out, row0, row1, cond=None, cond_value=0):
    """split_f16x2_rows_if for rows row0..row1 of an unbatched fp32 [rows, cols] matrix x: rowmax /
    out's planes and scales are those of the WHOLE matrix, only the range's entries are read /
    written."""
    if x.dim() != 2 or x.stride(1) != 1 or out.planes.dim() != 4:
        raise TypeError('split_f16x2_row_range_if: an unbatched matrix and its planes')
    rows, cols = x.shape
    L.call('naws_split_f16x2_row_range_if', x.data_ptr(), rows, int(row0), int(row1 - row0), cols,
           x.stride(0), rowmax.data_ptr(), out.planes.data_ptr(), out.inv_scale.data_ptr(),
           (cols + 31) // 32 * 32, _ptr(cond), int(cond_value), _stream())
    return out


# ----------------------------------------------------------------------------
# OICR refinement operators
# ----------------------------------------------------------------------------
def roi_label(s, u, l, cw=None, fg_thresh=0.5, bg_thresh_hi=0.5, bg_thresh_lo=-1.0, top_k=1,
              num_pos=9999, num_neg=9999, stats=None):
    """-> (RL int32 [n], RW fp32 [n]); stats (fp32 [4], optional) accumulates the op's counters."""
    _chk(s, 'S'); _chk(u, 'U'); _chk(l, 'L')
    if s.dim() != 2 or u.dim() != 2 or l.dim() != 2 or l.shape[0] != 1 or \
            u.shape[0] != u.shape[1] or s.shape[0] != u.shape[0]:
        raise L.NawsError('naws_roi_label_fwd', L.ERR_SHAPE)        # ENFORCE sites :13-22
    n, cs = s.shape
    c = l.shape[1]
    if cw is not None:
        _chk(cw, 'CW')
    dev = s.device
    ws = torch.empty((max(L.load().naws_roi_label_workspace_bytes(n, c, top_k), 16) // 4,),
                     device=dev, dtype=torch.int32)
    rl = torch.empty((n,), device=dev, dtype=torch.int32)
    rw = torch.empty((n,), device=dev, dtype=_f32)
    st = stats if stats is not None else torch.zeros((4,), device=dev, dtype=_f32)
    L.call('naws_roi_label_fwd', s.data_ptr(), u.data_ptr(), l.data_ptr(), _ptr(cw), n, cs, c,
           float(fg_thresh), float(bg_thresh_hi), float(bg_thresh_lo), int(top_k), int(num_pos),
           int(num_neg), ws.data_ptr(), rl.data_ptr(), rw.data_ptr(), st.data_ptr(), _stream())
    return rl, rw


def softmax_with_loss_n(x, t, w=None, scale=1.0):
    """-> (P [N,D], loss [1])."""
    _chk(x, 'X'); _chk(t, 'T', torch.int32)
    if x.dim() != 2 or t.numel() != x.shape[0] or (w is not None and w.numel() != x.shape[0]):
        raise L.NawsError('naws_softmax_with_loss_n_fwd', L.ERR_SHAPE)
    if w is not None:
        _chk(w, 'W')
    n, d = x.shape
    ws = torch.empty((L.load().naws_softmax_with_loss_n_workspace_floats(n),), device=x.device,
                     dtype=_f32)
    p = torch.empty_like(x)
    loss = torch.empty((1,), device=x.device, dtype=_f32)
    L.call('naws_softmax_with_loss_n_fwd', x.data_ptr(), t.data_ptr(), _ptr(w), n, d, float(scale),
           ws.data_ptr(), p.data_ptr(), loss.data_ptr(), _stream())
    return p, loss


def softmax_with_loss_n_grad(t, w, p, dloss, scale=1.0):
    _chk(p, 'P'); _chk(t, 'T', torch.int32); _chk(dloss, 'd_avg_loss')
    if w is not None:
        _chk(w, 'W')
    n, d = p.shape
    ws = torch.empty((L.load().naws_softmax_with_loss_n_workspace_floats(n),), device=p.device,
                     dtype=_f32)
    dx = torch.empty_like(p)
    L.call('naws_softmax_with_loss_n_bwd', t.data_ptr(), _ptr(w), p.data_ptr(), dloss.data_ptr(), n,
           d, float(scale), ws.data_ptr(), dx.data_ptr(), _stream())
    return dx


def roi_entropy(s, c, num_classes, rm_bg=True, mean=None, init=True):
    """-> E [1, num_classes]; `mean` (fp32 [num_classes]) is the op's running accumulator."""
    _chk(s, 'S'); _chk(c, 'C')
    if s.dim() != 1 or c.dim() != 1 or s.shape[0] != c.shape[0]:
        raise L.NawsError('naws_roi_entropy_fwd', L.ERR_SHAPE)       # ENFORCE sites :73-75
    e = torch.empty((1, int(num_classes)), device=s.device, dtype=_f32)
    L.call('naws_roi_entropy_fwd', s.data_ptr(), c.data_ptr(), s.shape[0], int(num_classes),
           int(bool(rm_bg)), e.data_ptr(), _ptr(mean), int(bool(init)), _stream())
    return e


# ----------------------------------------------------------------------------
# inference post-processing on the device
# ----------------------------------------------------------------------------
DEDUP_PASS_DTYPE = np.dtype([('im_scale', '<f8'), ('im_width', '<f4'), ('flip', '<i4'),
                             ('batch_index', '<f4')], align=True)      # = struct naws_dedup_pass


def roi_dedup(boxes, obn_scores, passes, dedup_boxes):
    """boxes [n,4], obn_scores [n] (device fp32); passes: list of (im_scale, im_width, flip,
    batch_index).  -> dict(rois [P,n,5], obn [P,n], index [P,n], inv [P,n], count [P]) on the device
    (rows >= count[p] of rois / obn / index are unspecified)."""
    _chk(boxes, 'boxes'); _chk(obn_scores, 'obn_scores')
    n = boxes.shape[0]
    if boxes.dim() != 2 or boxes.shape[1] != 4 or obn_scores.numel() != n:
        raise L.NawsError('naws_roi_dedup_fwd', L.ERR_SHAPE)
    rec = np.zeros((len(passes),), DEDUP_PASS_DTYPE)
    for i, (sc, w, fl, b) in enumerate(passes):
        rec[i] = (float(sc), float(w), int(bool(fl)), float(b))
    dev = boxes.device
    pd = torch.from_numpy(rec.view(np.uint8).reshape(-1)).to(dev)
    npass = len(passes)
    out = dict(rois=torch.empty((npass, n, 5), device=dev, dtype=_f32),
               obn=torch.empty((npass, n), device=dev, dtype=_f32),
               index=torch.empty((npass, n), device=dev, dtype=torch.int32),
               inv=torch.empty((npass, n), device=dev, dtype=torch.int32),
               count=torch.empty((npass,), device=dev, dtype=torch.int32))
    L.call('naws_roi_dedup_fwd', boxes.data_ptr(), obn_scores.data_ptr(), n, npass, pd.data_ptr(),
           float(dedup_boxes), out['rois'].data_ptr(), out['obn'].data_ptr(),
           out['index'].data_ptr(), out['inv'].data_ptr(), out['count'].data_ptr(), _stream())
    return out


def tta_accumulate(scores, inv_index, acc, first):
    """acc[n,k] (=|+=) scores[inv_index]."""
    _chk(scores, 'scores'); _chk(acc, 'acc')
    n, k = acc.shape
    L.call('naws_tta_accumulate', scores.data_ptr(), _ptr(inv_index), n, k, int(bool(first)),
           acc.data_ptr(), _stream())
    return acc


def tta_finish(acc, npass):
    L.call('naws_tta_finish', acc.data_ptr(), acc.numel(), int(npass), _stream())
    return acc


def det_limit(scores, keep, limit, cap):
    """scores [R,K] fp32, keep [C,R] bool -> (count [1], cls [cap], row [cap], score [cap]), one
    packed int32 / fp32 device buffer each; see naws_det_limit_fwd."""
    _chk(scores, 'scores')
    r, k = scores.shape
    c = keep.shape[0]
    if keep.dtype != torch.bool or tuple(keep.shape) != (c, r) or not keep.is_contiguous():
        raise TypeError('keep must be a contiguous bool [C, R] tensor')
    dev = scores.device
    ints = torch.empty((1 + 2 * cap,), device=dev, dtype=torch.int32)
    sc = torch.empty((cap,), device=dev, dtype=_f32)
    L.call('naws_det_limit_fwd', scores.data_ptr(), keep.data_ptr(), c, r, k, int(limit), int(cap),
           ints.data_ptr(), ints[1:].data_ptr(), ints[1 + cap:].data_ptr(), sc.data_ptr(), _stream())
    return ints, sc


class RowmaxTable(object):
    """[(first element, end element, row length, first rowmax index)] for acm_sgd_update."""

    def __init__(self, rows, device):
        self.host = np.ascontiguousarray(np.asarray(rows, dtype=np.int64).reshape(-1, 4))
        self.n = self.host.shape[0]


def stat_accumulate(i, l, ai, al, init):
    L.call('naws_stat_accumulate', i.data_ptr(), l.data_ptr(), i.numel(), int(init),
           ai.data_ptr(), al.data_ptr(), _stream())


# ----------------------------------------------------------------------------
# small built-ins
# ----------------------------------------------------------------------------
def unary(op, x, a=0.0, b=0.0, out=None):
    _chk(x, 'X')
    y = out if out is not None else torch.empty_like(x)
    L.call('naws_unary_f32', op, x.data_ptr(), x.numel(), float(a), float(b), y.data_ptr(),
           _stream())
    return y


def binary(op, a, b, out=None):
    _chk(a, 'A'); _chk(b, 'B')
    a2 = a.reshape(-1, a.shape[-1]) if a.dim() >= 1 else a.reshape(1, 1)
    b2 = b.reshape(-1, b.shape[-1]) if b.dim() >= 1 else b.reshape(1, 1)
    rows, cols = max(a2.shape[0], b2.shape[0]), max(a2.shape[1], b2.shape[1])
    y = out if out is not None else torch.empty((rows, cols), device=a.device, dtype=_f32)
    L.call('naws_binary_f32', op, a2.data_ptr(), a2.shape[0], a2.shape[1], b2.data_ptr(),
           b2.shape[0], b2.shape[1], y.data_ptr(), rows, cols, _stream())
    return y


def softmax_rows(x):
    _chk(x, 'X')
    y = torch.empty_like(x)
    L.call('naws_softmax_rows_fwd', x.data_ptr(), x.shape[0], x.shape[1], y.data_ptr(), _stream())
    return y


def softmax_rows_grad(y, dy):
    _chk(y, 'Y'); _chk(dy, 'dY')
    dx = torch.empty_like(y)
    L.call('naws_softmax_rows_bwd', y.data_ptr(), dy.data_ptr(), y.shape[0], y.shape[1],
           dx.data_ptr(), _stream())
    return dx


def transpose2d(x):
    _chk(x, 'X')
    y = torch.empty((x.shape[1], x.shape[0]), device=x.device, dtype=_f32)
    L.call('naws_transpose2d_f32', x.data_ptr(), x.shape[0], x.shape[1], y.data_ptr(), _stream())
    return y


def reduce_sum_axis0(x):
    _chk(x, 'X')
    y = torch.empty((1, x.shape[1]), device=x.device, dtype=_f32)
    L.call('naws_reduce_sum_axis0', x.data_ptr(), x.shape[0], x.shape[1], y.data_ptr(), _stream())
    return y


class BackgroundStream(object):
    """naws_stream_create: a HIP stream with a priority or a compute-unit mask, usable wherever
    torch takes a stream (`.stream` is a torch.cuda.ExternalStream over the same handle)."""

    def __init__(self, device, priority=0, cu_mask=None):
        handle = C.c_void_p()
        words = None
        if cu_mask is not None:
            words = (C.c_uint32 * len(cu_mask))(*[int(w) & 0xFFFFFFFF for w in cu_mask])
        with torch.cuda.device(device):
            L.call('naws_stream_create', int(priority), words, 0 if words is None else len(words),
                   C.byref(handle))
        self.handle = handle
        self.stream = torch.cuda.ExternalStream(handle.value, device=device)

    def close(self):
        if self.handle is not None and self.handle.value:
            L.call('naws_stream_destroy', self.handle)
        self.handle = None


def cu_mask_every(n_cu, stride, offset=0):
    """32-bit mask words selecting compute units offset, offset + stride, ... below n_cu."""
    words = [0] * ((n_cu + 31) // 32)
    for i in range(offset, n_cu, stride):
        words[i // 32] |= 1 << (i % 32)
    return words
