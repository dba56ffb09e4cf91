#!/usr/bin/env python3
"""Interleaved in-process A/B of the fp16x2 Winograd convolution of the 512 -> 512 layers: the
16-plane three-kernel route (input transform, 16 batched GEMMs, output transform) against the
frequency-column form (naws_conv3x3_winograd_nhwc_f16x2_col_fwd: 4 batched GEMMs of K = 4 Cin with
two accumulator sets, 8-plane output transform), one image per launch as the engine runs them, and
two images on two streams (how they overlap inside a training step).

    make -C csrc AB=1 && NAWS_LIB=$PWD/lib/libnaws_hip_ab.so python tools/ab_wino_col.py [--rounds 11]

Round 4's result (profiles/r04_wino_column_pmc.md): the column form LOSES, 0.200 vs 0.162 ms per
layer-image alone and 0.269 vs 0.236 with two images on two streams; it is kept in the A/B build.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from naws_hip import lib as L, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=11)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(1)
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    L.set_variant('wino', 1)             # the 16-plane entry never fuses here
    for name, cin, cout, h, w, dil in [('conv4_2', 512, 512, 75, 125, 1), ('conv5_1', 512, 512, 74, 124, 2)]:
        xs = [torch.randn((1, h, w, cin), device=dev, generator=g).relu_() for _ in range(2)]
        wt = torch.randn((cout, cin, 3, 3), device=dev, generator=g) * (2.0 / (9 * cin)) ** 0.5
        b = torch.randn((cout,), device=dev, generator=g)
        u = ops.winograd_weight_transform(wt)
        ucol = ops.winograd_weight_columns(u)
        forms = {'16 planes': (ops.split_f16x2(u), 1), 'columns': (ucol, 1), 'columns 64x128': (ucol, 8),
                 'columns 128x64': (ucol, 9)}
        am = [ops.amax_word(x) for x in xs]
        outs = {}
        times = {(k, m): [] for k in forms for m in ('one image', 'two streams')}
        for r in range(a.rounds + 1):
            for k, (u2, knob) in forms.items():
                L.set_variant('wino', knob)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                y = ops.conv3x3_winograd_nhwc_f16x2(xs[0], u2, b, dil, True, amax_in=am[0])
                e.record()
                torch.cuda.synchronize()
                if r == 0:
                    outs[k] = y.clone()
                else:
                    times[(k, 'one image')].append(s.elapsed_time(e))
                main_s = torch.cuda.current_stream(dev)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for i, st in enumerate(streams):
                    st.wait_event(s)
                    with torch.cuda.stream(st):
                        for _ in range(3):          # three layers per chain, as conv5_1..conv5_3
                            ops.conv3x3_winograd_nhwc_f16x2(xs[i], u2, b, dil, True, amax_in=am[i])
                        main_s.wait_event(st.record_event())
                e.record()
                torch.cuda.synchronize()
                if r:
                    times[(k, 'two streams')].append(s.elapsed_time(e) / 3)
        d = float((outs['columns'] - outs['16 planes']).abs().max() / outs['16 planes'].abs().max())
        med = {k: sorted(t)[len(t) // 2] for k, t in times.items()}
        print('%s %d->%d %dx%d d%d (ms per layer; one image per launch / two images on two streams): %s; '
              'max |diff| / max %.1e' % (name, cin, cout, h, w, dil, ', '.join(
                  '%s %.3f / %.3f' % (k, med[(k, 'one image')], med[(k, 'two streams')]) for k in forms), d),
              flush=True)
    L.set_variant('wino', 0)


if __name__ == '__main__':
    main()
