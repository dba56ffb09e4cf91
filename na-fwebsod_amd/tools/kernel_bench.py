#!/usr/bin/env python3
"""Per-kernel timing of the hot-path kernels at the BASELINE config-2 shapes
(2 images 600x1000, 2000 proposals each, fp32).  Prints one line per kernel with
achieved TFLOP/s or GB/s.  Kernels run on torch's current stream, so
torch.cuda.Event brackets them correctly."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from naws_hip import ops, lib  # noqa: E402


def timeit(fn, iters=5, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters  # ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--images', type=int, default=2)
    ap.add_argument('--rois', type=int, default=2000)
    ap.add_argument('--what', default='gemm,conv,roi')
    ap.add_argument('--iters', type=int, default=5)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    R = a.images * a.rois
    what = a.what.split(',')

    def rnd(*shape):
        return torch.empty(shape, device=dev).uniform_(-1, 1)

    if 'gemm' in what:
        shapes = [
            ('fc6 fwd  NT', R, 8192, 25088, False, True, 1),
            ('fc7 fwd  NT', R, 4096, 4096, False, True, 2),
            ('fc7 dgrad NN', R, 4096, 4096, False, False, 2),
            ('fc7 wgrad TN', 4096, 4096, R, True, False, 2),
            ('fc6 wgrad TN', 8192, 25088, R, True, False, 1),
        ]
        for name, m, n, k, ta, tb, batch in shapes:
            A = rnd(*((batch,) if batch > 1 else ()), *((k, m) if ta else (m, k)))
            B = rnd(*((batch,) if batch > 1 else ()), *((n, k) if tb else (k, n)))
            Cc = torch.empty(((batch, m, n) if batch > 1 else (m, n)), device=dev)
            ms = timeit(lambda: ops.gemm(A, B, ta, tb, out=Cc), a.iters)
            fl = 2.0 * m * n * k * batch
            print('%-14s M=%6d N=%6d K=%6d b=%d  %8.3f ms  %7.1f TFLOP/s (%.1f%% of 157.3)' % (
                name, m, n, k, batch, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100))
            del A, B, Cc
    if 'x3' in what:
        shapes = [('fc6 fwd ', R, 8192, 25088, 1), ('fc7 fwd ', R, 4096, 4096, 2),
                  ('fc7 wgrad', 4096, 4096, (R + 15) // 16 * 16, 2),
                  ('fc6 wgrad', 8192, 25088, (R + 15) // 16 * 16, 1)]
        for name, m, n, k, batch in shapes:
            bs = (batch,) if batch > 1 else ()
            A3 = torch.empty((3, *bs, k // 16, m, 16), device=dev).uniform_(-1, 1).bfloat16()
            B3 = torch.empty((3, *bs, k // 16, n, 16), device=dev).uniform_(-1, 1).bfloat16()
            Cc = torch.empty((*bs, m, n), device=dev)
            ms = timeit(lambda: ops.gemm_f32x3_nt(A3, B3, out=Cc), a.iters)
            fl = 2.0 * m * n * k * batch
            print('x3 %-10s M=%6d N=%6d K=%6d b=%d %8.3f ms %7.1f TFLOP/s fp32-equivalent '
                  '(%.1f%% of 416.7 = 2500/6; bf16 MFMA %.0f TFLOP/s)' % (
                      name, m, n, k, batch, ms, fl / ms / 1e9, fl / ms / 1e9 / 416.7 * 100,
                      6 * fl / ms / 1e9))
            del A3, B3, Cc
        tot = 0.0
        for cin, cout, h, w, dil, mult in [(64, 64, 600, 1000, 1, 1), (64, 128, 300, 500, 1, 1),
                                           (128, 128, 300, 500, 1, 1), (128, 256, 150, 250, 1, 1),
                                           (256, 256, 150, 250, 1, 2), (256, 512, 75, 125, 1, 1),
                                           (512, 512, 75, 125, 1, 2), (512, 512, 74, 124, 2, 3)]:
            xx = rnd(a.images, h, w, cin)
            w3 = torch.empty((3, 9 * cin // 16, cout, 16), device=dev).uniform_(-1, 1).bfloat16()
            b = rnd(cout)
            y = torch.empty((a.images, h, w, cout), device=dev)
            ms = timeit(lambda: ops.conv3x3_nhwc_f32x3(xx, w3, b, dil, True, out=y), a.iters)
            fl = 2.0 * a.images * h * w * cout * 9 * cin
            tot += ms * mult
            extra = ''
            if dil == 1:
                w2 = ops.split_f16x2(torch.empty((cout, 9 * cin), device=dev).uniform_(-1, 1))
                am = ops.amax_word(xx)
                msh = timeit(lambda: ops.conv3x3_nhwc_f16x2(xx, w2, b, True, out=y, amax_in=am), a.iters)
                extra = '   fp16x2 halo: %8.3f ms %7.1f TFLOP/s' % (msh, fl / msh / 1e9)
            print('x3 conv %3d->%3d %4dx%4d d%d  %8.3f ms  %7.1f TFLOP/s fp32-equivalent%s' % (
                cin, cout, h, w, dil, ms, fl / ms / 1e9, extra))
            del xx, w3, b, y
        print('x3 conv stack conv1_2..conv5_3 (%d images, one launch per layer): %.3f ms' % (a.images, tot))
        x = rnd(R, 25088)
        ms = timeit(lambda: ops.split_bf16x3(x), a.iters)
        print('split_bf16x3      [%d,25088]  %8.3f ms  %6.0f GB/s' % (R, ms, x.numel() * 10 / ms / 1e6))
        ms = timeit(lambda: ops.split_bf16x3(x, transpose=True), a.iters)
        print('split_bf16x3 (T)  [%d,25088]  %8.3f ms  %6.0f GB/s' % (R, ms, x.numel() * 10 / ms / 1e6))
        del x
        w = rnd(8192, 25088)
        ms = timeit(lambda: ops.split_bf16x3(w), a.iters)
        print('split_bf16x3   [8192,25088]  %8.3f ms  %6.0f GB/s' % (ms, w.numel() * 10 / ms / 1e6))
        del w
    if 'h2' in what:
        # fp16x2: fp32 GEMM as 3 f16 MFMA products per K-slab (ceiling 2500/3 = 833 TFLOP/s)
        shapes = [('fc6 fwd ', R, 8192, 25088, 1), ('fc7 fwd ', R, 4096, 4096, 2),
                  ('fc7 wgrad', 4096, 4096, (R + 31) // 32 * 32, 2),
                  ('fc6 wgrad', 8192, 25088, (R + 31) // 32 * 32, 1)]
        for name, m, n, k, batch in shapes:
            bs = (batch,) if batch > 1 else ()
            A2 = ops.split_f16x2(torch.empty((*bs, m, k), device=dev).normal_())
            B2 = ops.split_f16x2(torch.empty((*bs, n, k), device=dev).normal_())
            Cc = torch.empty((*bs, m, n), device=dev)
            ms = timeit(lambda: ops.gemm_f32_f16x2_nt(A2, B2, out=Cc), a.iters)
            fl = 2.0 * m * n * k * batch
            print('h2 %-10s M=%6d N=%6d K=%6d b=%d %8.3f ms %7.1f TFLOP/s fp32-equivalent '
                  '(%.1f%% of 833.3 = 2500/3; f16 MFMA %.0f TFLOP/s)' % (
                      name, m, n, k, batch, ms, fl / ms / 1e9, fl / ms / 1e9 / 833.3 * 100,
                      3 * fl / ms / 1e9))
            del A2, B2, Cc
        x = rnd(R, 25088)
        ms = timeit(lambda: ops.split_f16x2(x), a.iters)
        print('split_f16x2      [%d,25088]  %8.3f ms  %6.0f GB/s' % (R, ms, x.numel() * 12 / ms / 1e6))
        ms = timeit(lambda: ops.split_f16x2(x, transpose=True), a.iters)
        print('split_f16x2 (T)  [%d,25088]  %8.3f ms  %6.0f GB/s' % (R, ms, x.numel() * 12 / ms / 1e6))
        del x
        w = rnd(8192, 25088)
        ms = timeit(lambda: ops.split_f16x2(w), a.iters)
        print('split_f16x2   [8192,25088]  %8.3f ms  %6.0f GB/s' % (ms, w.numel() * 12 / ms / 1e6))
        del w
    if 'bf16' in what:
        PEAK = 2500.0
        shapes = [('fc6 fwd ', R, 8192, 25088, 1, False, False),
                  ('fc7 fwd ', R, 4096, 4096, 2, False, False),
                  ('fc7 dgrad', R, 4096, 4096, 2, False, True),
                  ('fc7 wgrad', 4096, 4096, R, 2, True, True),
                  ('fc6 wgrad', 8192, 25088, R, 1, True, True)]
        for name, m, n, k, batch, a16, b16 in shapes:
            bs = (batch,) if batch > 1 else ()
            A = rnd(*bs, m, k)
            B = rnd(*bs, n, k)
            if a16:
                A = A.bfloat16()
            if b16:
                B = B.bfloat16()
            Cc = torch.empty((*bs, m, n), device=dev)
            ms = timeit(lambda: ops.gemm_bf16_nt(A, B, out=Cc), a.iters)
            fl = 2.0 * m * n * k * batch
            by = A.numel() * A.element_size() + B.numel() * B.element_size() + Cc.numel() * 4
            print('bf16 %-10s M=%6d N=%6d K=%6d b=%d A16=%d B16=%d %8.3f ms %7.1f TFLOP/s (%.1f%% of 2500) '
                  '%6.0f GB/s min-traffic' % (name, m, n, k, batch, a16, b16, ms, fl / ms / 1e9,
                                              fl / ms / 1e9 / PEAK * 100, by / ms / 1e6))
            del A, B, Cc
        for name, m, n, k, batch in [('fc6 fwd ', R, 8192, 25088, 1), ('fc7 fwd ', R, 4096, 4096, 2),
                                     ('fc6 wgrad', 8192, 25088, (R + 63) // 64 * 64, 1)]:
            bs = (batch,) if batch > 1 else ()
            A1 = torch.empty((*bs, k // 16, m, 16), device=dev).uniform_(-1, 1).bfloat16()
            B1 = torch.empty((*bs, k // 16, n, 16), device=dev).uniform_(-1, 1).bfloat16()
            Cc = torch.empty((*bs, m, n), device=dev)
            ms = timeit(lambda: ops.gemm_bf16_slab_nt(A1, B1, out=Cc), a.iters)
            fl = 2.0 * m * n * k * batch
            print('bf16 slab %-10s M=%6d N=%6d K=%6d b=%d %8.3f ms %7.1f TFLOP/s (%.1f%% of 2500)' % (
                name, m, n, k, batch, ms, fl / ms / 1e9, fl / ms / 1e9 / PEAK * 100))
            del A1, B1, Cc
        x = rnd(2, R, 4096)
        ms = timeit(lambda: ops.transpose_to_bf16(x, rows_pad=(R + 7) // 8 * 8), a.iters)
        print('transpose_to_bf16 [2,%d,4096]  %8.3f ms  %6.0f GB/s' % (R, ms, x.numel() * 6 / ms / 1e6))
        del x
        for cin, cout, h, w, dil in [(64, 64, 600, 1000, 1), (128, 128, 300, 500, 1),
                                     (256, 256, 150, 250, 1), (512, 512, 75, 125, 1),
                                     (512, 512, 74, 124, 2)]:
            x = rnd(a.images, h, w, cin)
            wp = rnd(cout, 3, 3, cin)
            b = rnd(cout)
            y = torch.empty((a.images, h, w, cout), device=dev)
            ms = timeit(lambda: ops.conv3x3_nhwc_bf16(x, wp, b, dil, True, out=y), a.iters)
            fl = 2.0 * a.images * h * w * cout * 9 * cin
            print('bf16 conv %3d->%3d %4dx%4d d%d  %8.3f ms  %7.1f TFLOP/s' % (
                cin, cout, h, w, dil, ms, fl / ms / 1e9))
            del x, wp, b, y
    if 'conv' in what:
        layers = [(64, 64, 600, 1000, 1), (64, 128, 300, 500, 1), (128, 128, 300, 500, 1),
                  (128, 256, 150, 250, 1), (256, 256, 150, 250, 1), (256, 512, 75, 125, 1),
                  (512, 512, 75, 125, 1), (512, 512, 74, 124, 2)]
        tot_ms, tot_fl = 0.0, 0.0
        mult = {(256, 256, 150, 250, 1): 2, (512, 512, 75, 125, 1): 2, (512, 512, 74, 124, 2): 3}
        for cin, cout, h, w, dil in layers:
            x = rnd(a.images, h, w, cin)
            wp = rnd(cout, 3, 3, cin)
            b = rnd(cout)
            y = torch.empty((a.images, h, w, cout), device=dev)
            ms = timeit(lambda: ops.conv3x3_nhwc(x, wp, b, dil, True, out=y), a.iters)
            if cin >= 128:
                u = rnd(16, cout, cin)
                msw = timeit(lambda: ops.conv3x3_winograd_nhwc(x, u, b, dil, True, out=y), a.iters)
                u3 = ops.split_bf16x3(u)
                msx = timeit(lambda: ops.conv3x3_winograd_nhwc_f32x3(x, u3, b, dil, True, out=y), a.iters)
                u2 = ops.split_f16x2(u)
                msh = timeit(lambda: ops.conv3x3_winograd_nhwc_f16x2(x, u2, b, dil, True, out=y), a.iters)
                print('   winograd F(2x2,3x3): %8.3f ms (direct %8.3f ms, winograd fp32x3 %8.3f ms, '
                      'winograd fp16x2 %8.3f ms)' % (msw, ms, msx, msh))
            fl = 2.0 * a.images * h * w * cout * 9 * cin
            k = mult.get((cin, cout, h, w, dil), 1)
            tot_ms += ms * k
            tot_fl += fl * k
            print('conv %3d->%3d %4dx%4d d%d  %8.3f ms  %7.1f TFLOP/s (%.1f%%)' % (
                cin, cout, h, w, dil, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100))
            del x, wp, b, y
        x = rnd(a.images, 3, 600, 1000)
        w1 = rnd(64, 3, 3, 3)
        b1 = rnd(64)
        ms = timeit(lambda: ops.conv3x3_c3_nchw_to_nhwc(x, w1, b1, True), a.iters)
        print('conv1_1 direct            %8.3f ms' % ms)
        tot_ms += ms
        tot_fl += 2.0 * a.images * 600 * 1000 * 64 * 27
        print('conv stack (MFMA layers + conv1_1, no pools): %.3f ms, %.1f TFLOP/s (%.1f%%)' % (
            tot_ms, tot_fl / tot_ms / 1e9, tot_fl / tot_ms / 1e9 / 157.3 * 100))
    if 'infer' in what:
        # BASELINE configs[4]: multi-scale TTA inference, 4000 proposals per image, forward only
        import numpy as np
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
        from detectron.datasets import synthetic
        from naws_hip.engine import WsddnEngine
        c = 20
        blobs = synthetic.init_blobs(c, seed=11)
        eng = WsddnEngine(c + 1, dev, gpu_num=1, seed=11)
        eng.set_conv_blobs(blobs)
        eng.set_head_blobs(blobs)
        tot = tot_pair = 0.0
        for short in (480, 576, 688, 864, 1200):
            hh, ww = short, int(round(short * 1000 / 600))
            mb = synthetic.make_minibatch(synthetic.make_roidb(1, 4000, c, hh, ww, seed=13), c,
                                          max_rois=4000)
            t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
            seg = [0, t['rois'].shape[0]]
            ms_all = timeit(lambda: eng.infer(t['data'], t['rois'], t['obn_scores'], seg=seg), a.iters)
            ms_conv = timeit(lambda: eng.conv_body(t['data']), a.iters)
            conv5 = eng.conv_body(t['data'])
            ms_roi = timeit(lambda: ops.roi_pool_f(conv5, t['rois'], 7, 7, 0.125,
                                                   boost=t['obn_scores'].reshape(-1), layout='NHWC'),
                            a.iters)
            # plain + mirrored pass as one batch of two images (core/test_wsl.im_detect_bbox_pair)
            d2 = torch.cat([t['data'], t['data'].flip(-1)], 0)
            r2 = torch.cat([t['rois'], t['rois']], 0)
            r2[t['rois'].shape[0]:, 0] = 1
            o2 = torch.cat([t['obn_scores'], t['obn_scores']], 0)
            seg2 = [0, seg[1], 2 * seg[1]]
            ms_pair = timeit(lambda: eng.infer(d2, r2, o2, seg=seg2), a.iters)
            tot += 2 * ms_all      # + horizontal flip
            tot_pair += ms_pair
            print('infer %4dx%4d R=%d: total %7.3f ms (conv %6.3f, RoIPool %6.3f, head %6.3f); '
                  'plain+flip as one batch of 2: %7.3f ms' % (
                      hh, ww, t['rois'].shape[0], ms_all, ms_conv, ms_roi,
                      ms_all - ms_conv - ms_roi, ms_pair))
            del d2, r2, o2
            del conv5, t
        print('10-pass TTA (5 scales x flip): %.1f ms per image forward, %.1f ms with paired flips' % (
            tot, tot_pair))
    if 'roi' in what:
        import numpy as np
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..',
                                        'tests'))
        from helpers import make_rois
        rois = torch.from_numpy(make_rois(np.random.default_rng(11), a.images, a.rois, 600, 1000,
                                          degenerate=False)).to(dev)
        feat = rnd(a.images, 74, 124, 512).abs_()
        boost = torch.empty((R,), device=dev).uniform_(1, 2)
        out = torch.empty((R, 512, 7, 7), device=dev)
        ms = timeit(lambda: ops.roi_pool_f(feat, rois, 7, 7, 0.125, boost=boost, layout='NHWC',
                                           out=out), a.iters)
        by = out.numel() * 4 + feat.numel() * 4
        print('roi_pool+boost NHWC R=%d  %8.3f ms  %7.1f GB/s algorithmic' % (R, ms, by / ms / 1e6))


if __name__ == '__main__':
    main()
