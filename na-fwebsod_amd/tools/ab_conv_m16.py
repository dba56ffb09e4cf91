#!/usr/bin/env python3
"""Interleaved in-process A/B of the direct fp16x2 convolution, layer by layer on the shapes of a
600 x 1000 image (one image per launch): the 16x16x32 kernel (conv_h2_m16_kernel, knob conv_ring =
13; A/B build only: measured, not adopted - csrc/conv_x3.hip at its dispatch) against the 32x32x16
kernel in use (conv_h2_wp_kernel, conv_ring = 11).

    make -C csrc AB=1 && NAWS_LIB=$PWD/lib/libnaws_hip_ab.so python tools/ab_conv_m16.py [--rounds 15]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from naws_hip import lib as L, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=15)
    ap.add_argument('--images', type=int, default=1)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(1)
    for name, cin, cout, h, w, pool in [('conv1_2', 64, 64, 600, 1000, True), ('conv2_1', 64, 128, 300, 500, False),
                                        ('conv2_2', 128, 128, 300, 500, True), ('conv3_1', 128, 256, 150, 250, False),
                                        ('conv3_2', 256, 256, 150, 250, False), ('conv3_3', 256, 256, 150, 250, True)]:
        x = torch.randn((a.images, h, w, cin), device=dev, generator=g).relu_()
        wt = torch.randn((cout, cin, 3, 3), device=dev, generator=g) * (2.0 / (9 * cin)) ** 0.5
        b = torch.randn((cout,), device=dev, generator=g)
        w2 = ops.split_f16x2(ops.conv3x3_pack_weight(wt).view(cout, 9 * cin))
        am = ops.amax_word(x)
        outs, times = {}, {13: [], 11: []}
        for r in range(a.rounds + 1):
            for v in (13, 11):
                L.set_variant('conv_ring', v)
                amo = torch.zeros((1,), device=dev, dtype=torch.int32)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                y = ops.conv3x3_nhwc_f16x2(x, w2, b, True, amax_in=am, amax_out=amo, pool2=pool,
                                           amax_out_zeroed=True)
                e.record()
                torch.cuda.synchronize()
                if r == 0:
                    outs[v] = y.clone()
                else:
                    times[v].append(s.elapsed_time(e))
        L.set_variant('conv_ring', 11)
        d = float((outs[13] - outs[11]).abs().max() / outs[11].abs().max())
        med = {v: sorted(t)[len(t) // 2] for v, t in times.items()}
        print('%s %d->%d %dx%d%s x%d: 16x16x32 %.3f ms, 32x32x16 %.3f ms; max |diff| / max %.1e' % (
            name, cin, cout, h, w, ' +pool' if pool else '', a.images, med[13], med[11], d), flush=True)


if __name__ == '__main__':
    main()
