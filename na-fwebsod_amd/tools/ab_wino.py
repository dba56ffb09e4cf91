#!/usr/bin/env python3
"""Interleaved in-process A/B of the fp16x2 Winograd convolution: the fused GEMM + output-transform
kernel (knob wino = 0, default) against the three-kernel route (wino = 1), on the conv4 / conv5
shapes of a 600 x 1000 image, one image per launch as the engine runs them; results compared.

    python tools/ab_wino.py [--rounds 9] [--images 1]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from naws_hip import lib as L, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=9)
    ap.add_argument('--images', type=int, default=1)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(1)
    for name, cin, cout, h, w, dil in [('conv4_1', 256, 512, 75, 125, 1), ('conv4_2', 512, 512, 75, 125, 1),
                                       ('conv5_1', 512, 512, 74, 124, 2), ('conv3_2', 256, 256, 150, 250, 1)]:
        x = torch.randn((a.images, h, w, cin), device=dev, generator=g).relu_()
        wt = torch.randn((cout, cin, 3, 3), device=dev, generator=g) * (2.0 / (9 * cin)) ** 0.5
        b = torch.randn((cout,), device=dev, generator=g)
        u2 = ops.split_f16x2(ops.winograd_weight_transform(wt))
        am = ops.amax_word(x)
        variants = (0, 1, 2, 4, 5, 6)
        outs, times = {}, {v: [] for v in variants}
        for r in range(a.rounds + 1):
            for v in variants:
                L.set_variant('wino', v)
                amo = torch.zeros((1,), device=dev, dtype=torch.int32)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                y = ops.conv3x3_winograd_nhwc_f16x2(x, u2, b, dil, True, amax_in=am, amax_out=amo)
                e.record()
                torch.cuda.synchronize()
                if r == 0:
                    outs[v] = (y.clone(), int(amo.item()))
                else:
                    times[v].append(s.elapsed_time(e))
        L.set_variant('wino', 0)
        d = max(float((outs[v][0] - outs[1][0]).abs().max() / outs[1][0].abs().max()) for v in variants)
        med = {v: sorted(t)[len(t) // 2] for v, t in times.items()}
        print('%s %d->%d %dx%d d%d x%d: auto %.3f ms, three kernels %.3f, fused: LDS-DMA %.3f, '
              'wave-private 32 tiles x K32 %.3f, 64 x K32 %.3f, 32 x K64 %.3f; max |diff| / max %.1e' % (
                  name, cin, cout, h, w, dil, a.images, med[0], med[1], med[2], med[4], med[5], med[6], d),
              flush=True)


if __name__ == '__main__':
    main()
