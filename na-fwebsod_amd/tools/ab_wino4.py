#!/usr/bin/env python3
"""Interleaved in-process A/B of Winograd F(4x4,3x3) (csrc/winograd4.hip) against F(2x2,3x3)
(csrc/winograd.hip) in the fp16x2 plan: (1) layer by layer on the conv4 / conv5 shapes of a
600 x 1000 image, one image per launch as the engine runs them, each against a float64
convolution; (2) the whole conv body (engine.conv_body on the bench images, two engines that
differ only in WINO_F4_MIN_CIN).

    python tools/ab_wino4.py [--rounds 9] [--skip-body] [--h2 V ...]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from naws_hip import lib as L, ops  # noqa: E402


def timed(fn, rounds):
    ts = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts) // 2]


def layers(a, dev):
    import torch.nn.functional as F
    g = torch.Generator(device=dev).manual_seed(1)
    shapes = [('conv4_1', 256, 512, 75, 125, 1), ('conv4_2', 512, 512, 75, 125, 1),
              ('conv5_1', 512, 512, 74, 124, 2), ('conv4_2@1200x2000', 512, 512, 150, 250, 1)]
    for name, cin, cout, h, w, dil in shapes:
        x = torch.randn((a.images, h, w, cin), device=dev, generator=g).relu_()
        wt = torch.randn((cout, cin, 3, 3), device=dev, generator=g) * (2.0 / (9 * cin)) ** 0.5
        b = torch.randn((cout,), device=dev, generator=g)
        u2 = ops.split_f16x2(ops.winograd_weight_transform(wt))
        u4 = ops.split_f16x2(ops.winograd4_weight_transform(wt))
        am = ops.amax_word(x)
        ref = F.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), b.double(), padding=dil,
                              dilation=dil)).permute(0, 2, 3, 1)
        res = {}
        for label, u in (('F(2x2)', u2), ('F(4x4)', u4)):
            amo = torch.zeros((1,), device=dev, dtype=torch.int32)
            run = lambda: ops.conv3x3_winograd_nhwc_f16x2(x, u, b, dil, True, amax_in=am, amax_out=amo)  # noqa: E731
            y = run()
            torch.cuda.synchronize()
            err = float((y.double() - ref).abs().max() / ref.abs().max())
            res[label] = (None, err)
        # interleaved timing
        t = {k: [] for k in res}
        for _ in range(a.rounds):
            for label, u in (('F(2x2)', u2), ('F(4x4)', u4)):
                amo = torch.zeros((1,), device=dev, dtype=torch.int32)
                t[label].append(timed(lambda: ops.conv3x3_winograd_nhwc_f16x2(
                    x, u, b, dil, True, amax_in=am, amax_out=amo), 1))
        med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
        print('%s %d->%d %dx%d d%d x%d: F(2x2) %.3f ms (err %.1e), F(4x4) %.3f ms (err %.1e)' % (
            name, cin, cout, h, w, dil, a.images, med['F(2x2)'], res['F(2x2)'][1], med['F(4x4)'],
            res['F(4x4)'][1]), flush=True)


def body(a, dev):
    from detectron.datasets import synthetic
    from naws_hip.engine import WsddnEngine
    blobs = synthetic.init_blobs(20, seed=11)
    mb = synthetic.make_minibatch(synthetic.make_roidb(a.images_body, 16, 20, 600, 1000, seed=11), 20)
    data = torch.from_numpy(mb['data']).to(dev)
    engines = {}
    for label, mincin in (('F(2x2) deep layers', 0), ('F(4x4) from Cin 512', 512), ('F(4x4) from Cin 256', 256)):
        eng = WsddnEngine(21, dev, gpu_num=a.images_body, seed=11)
        eng.WINO_F4_MIN_CIN = mincin
        eng.set_conv_blobs(blobs)
        engines[label] = eng
    times = {k: [] for k in engines}
    outs = {}
    for r in range(a.rounds + 1):
        for label, eng in engines.items():
            y = eng.conv_body(data)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(a.iters):
                y = eng.conv_body(data)
            e.record()
            torch.cuda.synchronize()
            if r == 0:
                outs[label] = y.clone()
            else:
                times[label].append(s.elapsed_time(e) / a.iters)
    first = next(iter(outs))
    for label in engines:
        ts = sorted(times[label])
        d = float((outs[label] - outs[first]).abs().max() / outs[first].abs().max())
        print('conv body, %-22s median %.3f ms (min %.3f max %.3f); max |diff| / max vs %s %.1e' % (
            label, ts[len(ts) // 2], ts[0], ts[-1], first, d), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=9)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--images', type=int, default=1)
    ap.add_argument('--images-body', type=int, default=2)
    ap.add_argument('--skip-body', action='store_true')
    ap.add_argument('--skip-layers', action='store_true')
    ap.add_argument('--h2', type=int, nargs='*', default=[0],
                    help='gemm knob "h2" values to run the layer A/B under (A/B build for most)')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    if not a.skip_layers:
        for v in a.h2:
            L.set_variant('h2', v)
            print('--- h2 variant %d' % v)
            layers(a, dev)
        L.set_variant('h2', 0)
    if not a.skip_body:
        body(a, dev)


if __name__ == '__main__':
    main()
