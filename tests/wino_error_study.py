#!/usr/bin/env python3
"""CPU study behind csrc/winograd4.hip (no GPU; it imports the ORACLE's layer table, so it lives under
tests/, not in the product package): the per-channel error of conv5_3 when conv4_1..conv5_3 run as Winograd F(2x2,3x3) or
F(4x4,3x3) emulated in numpy fp32 (transforms fp32, products by fp32 BLAS; split=1 also rounds V
under one power-of-two scale per tensor and U under one per (frequency, output channel) to the
f16 hi + lo pair), against the torch fp32 direct convolution (the oracle) and a float64 one.
Yardstick = tests/test_gpu_fullsize_oracle.py's: max error per channel / the channel's
pre-activation RMS; plus max error / max|conv5_3| and the dark-third measure.

    python tests/wino_error_study.py 600 1000 kaiming|skewed

Measured (600 x 1000): F(2x2) 2.1e-5 / 2.1e-5, F(4x4) 3.8e-5 / 3.6e-5 (kaiming / skewed, split=1);
the fp32 direct sum itself is 1.6e-5 / 1.9e-5 from float64."""
import sys, time
import numpy as np, torch, torch.nn.functional as F
import os
_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(_HERE, '..')); sys.path.insert(0, os.path.join(_HERE, '..', 'na-fwebsod_amd'))
from detectron.datasets import synthetic
from oracle import oracle

def mats(m):
    if m == 2:
        BT = np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], np.float64)
        G = np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], np.float64)
        AT = np.array([[1,1,1,0],[0,1,-1,-1]], np.float64)
    elif m == 4:
        BT = np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]], np.float64)
        G = np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]], np.float64)
        AT = np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]], np.float64)
    elif m == 5:   # F(4x4,3x3) on the points 0, 1, -1, 2, -1/2, inf (csrc/winograd4.hip since round 6)
        BT = np.array([[2,3,-4,-3,2,0],[0,2,5,1,-2,0],[0,2,1,-5,2,0],[0,-1,-2,1,2,0],[0,-2,1,2,-1,0],[0,2,3,-4,-3,2]], np.float64)
        G = np.array([[1/2,0,0],[1/6,1/6,1/6],[1/6,-1/6,1/6],[1/30,1/15,2/15],[16/15,-8/15,4/15],[0,0,1/2]], np.float64)
        AT = np.array([[1,1,1,1,1,0],[0,1,-1,2,-1/2,0],[0,1,1,4,1/4,0],[0,1,-1,8,-1/8,1]], np.float64)
    elif m == 3:   # F(3x3,3x3): points 0, 1, -1, 2, inf  (5x5 tiles)
        BT = np.array([[2,-1,-2,1,0],[0,-2,-1,1,0],[0,2,-3,1,0],[0,-1,0,1,0],[0,2,-1,-2,1]], np.float64)
        G = np.array([[1/2,0,0],[-1/2,-1/2,-1/2],[-1/6,1/6,-1/6],[1/6,1/3,2/3],[0,0,1]], np.float64)
        AT = np.array([[1,1,1,1,0],[0,1,-1,2,0],[0,1,1,4,1]], np.float64)
    return BT, G, AT

def round22(a, bound=None):
    """f16 hi + lo under one power-of-two scale: bound (>= max|a|) maps below 2^15"""
    a = np.asarray(a, np.float32)
    if bound is None: bound = float(np.abs(a).max())
    sc = np.float32(2.0 ** (14 - np.floor(np.log2(bound))))
    t = a * sc
    hi = t.astype(np.float16).astype(np.float32)
    lo = (t - hi).astype(np.float16).astype(np.float32)
    return (hi + lo) / sc

def wino_conv(x, w, b, m, split=False, mixed=(None, None)):
    """x [C,H,W] fp32, w [O,C,3,3], pad 1, dilation 1.  F(m x m, 3x3), transforms and products fp32.
    mixed = (mh, mw) for rectangular forms."""
    mh, mw = (m, m) if mixed[0] is None else mixed
    BTh, Gh, ATh = mats(mh); BTw, Gw, ATw = mats(mw)
    mh, mw = ATh.shape[0], ATw.shape[0]          # output tile side (m = 5 names a point set of F(4x4))
    C, H, W = x.shape; O = w.shape[0]
    th, tw = -(-H // mh), -(-W // mw)
    ah, aw = mh + 2, mw + 2
    xp = np.zeros((C, th * mh + 2, tw * mw + 2), np.float32)
    xp[:, 1:H + 1, 1:W + 1] = x
    # tiles [C, th, tw, ah, aw]
    s = xp.strides
    tiles = np.lib.stride_tricks.as_strided(xp, (C, th, tw, ah, aw), (s[0], s[1] * mh, s[2] * mw, s[1], s[2]))
    U = np.einsum('ai,ocij,bj->abco', Gh, w.astype(np.float64), Gw).astype(np.float32)        # [ah,aw,C,O]
    V = np.einsum('ai,ctuij->actuj', BTh.astype(np.float32), tiles, optimize=False)
    V = np.einsum('bj,actuj->abctu', BTw.astype(np.float32), V, optimize=False).astype(np.float32)   # [ah,aw,C,th,tw]
    if split:
        amp = float(np.abs(BTh).sum(1).max() * np.abs(BTw).sum(1).max())
        V = round22(V, amp * max(float(np.abs(x).max()), 1e-30))
        for a_ in range(ah):
            for b_ in range(aw):
                for o_ in range(O): U[a_, b_][:, o_] = round22(U[a_, b_][:, o_], max(float(np.abs(U[a_, b_][:, o_]).max()), 1e-30))
    V2 = V.reshape(ah, aw, C, th * tw)
    M = np.empty((ah, aw, O, th * tw), np.float32)
    for a in range(ah):
        for bb in range(aw):
            M[a, bb] = U[a, bb].T @ V2[a, bb]
    M = M.reshape(ah, aw, O, th, tw)
    Y = np.einsum('ia,abotu->ibotu', ATh.astype(np.float32), M, optimize=False)
    Y = np.einsum('jb,ibotu->otiuj', ATw.astype(np.float32), Y, optimize=False).astype(np.float32)   # [O,th,mh,tw,mw]
    Y = Y.reshape(O, th * mh, tw * mw)[:, :H, :W] + b[:, None, None].astype(np.float32)
    return Y.astype(np.float32)

def wino_layer(x, w, b, dil, m, split, mixed=(None, None)):
    """x [1,C,H,W] torch; dilation by parity sub-images"""
    xn = x[0].numpy(); C, H, W = xn.shape
    out = np.empty((w.shape[0], H, W), np.float32)
    for py in range(dil):
        for px in range(dil):
            out[:, py::dil, px::dil] = wino_conv(np.ascontiguousarray(xn[:, py::dil, px::dil]), w.numpy(), b.numpy(), m, split, mixed)
    return torch.from_numpy(out)[None]

def body(data, blobs, plan, stats=None, split=False, f64=False):
    """plan: dict layer -> m (0 = direct)"""
    x = data.double() if f64 else data
    for item in oracle.VGG16_LAYERS:
        if item == 'P2': x = F.max_pool2d(x, 2, 2)
        elif item == 'P1': x = F.max_pool2d(x, 2, 1)
        else:
            name, _, _, dil = item
            w, b = blobs[name + '_w'], blobs[name + '_b']
            m = plan.get(name, 0)
            if m and not f64:
                if isinstance(m, tuple): x = wino_layer(x, w, b, dil, 0, split, m)
                else: x = wino_layer(x, w, b, dil, m, split)
            else:
                x = F.conv2d(x, w.double() if f64 else w, b.double() if f64 else b, padding=dil, dilation=dil)
            if stats is not None: stats[name] = x.double().pow(2).mean(dim=(0, 2, 3)).sqrt().numpy()
            x = F.relu(x)
    return x

def measure(got, want, rms):
    err = (got.double() - want.double()).abs()
    live = rms > 0
    pc = float((err.amax(dim=(0, 2, 3)).numpy()[live] / rms[live]).max())
    tot = float(err.max() / want.abs().max())
    wd = want.shape[3] // 3 - 4
    global DARK
    DARK = float(err[..., :wd].max() / want[..., :wd].abs().max())
    return pc, tot

if __name__ == '__main__':
    H, W = int(sys.argv[1]), int(sys.argv[2]); statsname = sys.argv[3]
    torch.set_num_threads(8)
    blobs = synthetic.init_blobs(20, seed=11)
    mb = synthetic.make_minibatch(synthetic.make_roidb(1, 100, 20, H, W, seed=11), 20)
    if statsname == 'skewed':
        blobs = synthetic.skew_blobs(blobs, seed=11); mb['data'] = synthetic.skew_images(mb['data'])
    blobs = {k: torch.from_numpy(np.asarray(v)) if not isinstance(v, torch.Tensor) else v for k, v in blobs.items() if k.startswith('conv')}
    data = torch.from_numpy(mb['data'])
    st = {}
    t0 = time.time(); ref = body(data, blobs, {}, st); print('direct fp32 %.1fs' % (time.time() - t0), flush=True)
    ref64 = body(data, blobs, {}, f64=True).float()
    rms = st['conv5_3']
    print('direct fp32 vs f64: per-channel %.2e  tensor %.2e' % measure(ref, ref64, rms), flush=True)
    deep = ['conv4_1', 'conv4_2', 'conv4_3', 'conv5_1', 'conv5_2', 'conv5_3']
    for label, plan in (('F2 all deep', {k: 2 for k in deep}),
                        ('F4 all deep', {k: 4 for k in deep}),
                        ('F4 new points', {k: 5 for k in deep}),

                        ):
        for split in (False, True):
            t0 = time.time(); y = body(data, blobs, plan, split=split)
            b = measure(y, ref64, rms); a = measure(y, ref, rms); print('   dark third vs fp32 oracle %.2e' % DARK)
            print('%-18s split=%d: vs fp32 oracle per-channel %.2e tensor %.2e | vs f64 per-channel %.2e tensor %.2e  (%.0fs)' % ((label, split) + a + b + (time.time() - t0,)), flush=True)
