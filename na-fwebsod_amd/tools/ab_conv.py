#!/usr/bin/env python3
"""(Needs the A/B build: `make -C na-fwebsod_amd/csrc AB=1`, run with
NAWS_LIB=na-fwebsod_amd/lib/libnaws_hip_ab.so.)  Interleaved in-process A/B of the fp16x2 halo-tile conv pipelines (knob conv_ring: 0 = one-step
weight prefetch, 3 / 4 / 6 = weight ring of that depth) and channel-tile widths (NAWS_CONV_BN) on
the VGG-16 layer shapes of a 600 x 1000 image.  Both knobs are read per call, so all variants run
round-robin in ONE process on the same operands; every variant's output must be bit-identical to
the first one's (same MFMA order).  Also times the Winograd route of the deep layers beside them.

    python tools/ab_conv.py --rings 0 4 [--images 1] [--rounds 7] [--bn 0]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from naws_hip import lib as L, ops  # noqa: E402

LAYERS = [('conv1_2', 64, 64, 600, 1000, 1), ('conv2_1', 64, 128, 300, 500, 1),
          ('conv2_2', 128, 128, 300, 500, 1), ('conv3_1', 128, 256, 150, 250, 1),
          ('conv3_2', 256, 256, 150, 250, 1), ('conv4_1', 256, 512, 75, 125, 1),
          ('conv4_2', 512, 512, 75, 125, 1), ('conv5_1', 512, 512, 74, 124, 2)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rings', type=int, nargs='+', default=[0, 4])
    ap.add_argument('--bn', type=int, nargs='+', default=[0], help='0 = the heuristic, 64, 128')
    ap.add_argument('--images', type=int, default=1)
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--only', default='')
    ap.add_argument('--wino', action='store_true')
    ap.add_argument('--stamp', action='store_true', help='phase shares of the stamped ring-4 build')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(3)
    variants = [(r, b) for r in a.rings for b in a.bn]
    for name, cin, cout, h, w, dil in LAYERS:
        if a.only and a.only not in name:
            continue
        x = torch.randn((a.images, h, w, cin), device=dev, generator=g).relu_()
        wt = torch.randn((cout, 9 * cin), device=dev, generator=g) * 0.05
        b = torch.randn((cout,), device=dev, generator=g)
        w2 = ops.split_f16x2(wt)
        am = ops.amax_word(x)
        outs = [torch.empty((a.images, h, w, cout), device=dev) for _ in variants]
        times = [[] for _ in variants]
        for r in range(a.rounds + 1):
            for i, (ring, bn) in enumerate(variants):
                L.set_variant('conv_ring', ring)
                L.set_variant('conv_bn', bn or 0)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                ops.conv3x3_nhwc_f16x2(x, w2, b, True, out=outs[i], amax_in=am, dilation=dil)
                e.record()
                torch.cuda.synchronize()
                if r > 0:
                    times[i].append(s.elapsed_time(e))
        fl = 2.0 * a.images * h * w * cout * 9 * cin
        msg = []
        for i, (ring, bn) in enumerate(variants):
            med = sorted(times[i])[len(times[i]) // 2]
            same = bool(torch.equal(outs[i], outs[0]))
            msg.append('ring%d/bn%d: %.3f ms %.0f TF (x3 = %.2f PF)%s' % (
                ring, bn, med, fl / med / 1e9, 3 * fl / med / 1e12, '' if same else '  MISMATCH'))
        if a.wino:
            u = ops.winograd_weight_transform(wt.view(cout, 3, 3, cin).permute(0, 3, 1, 2).contiguous())
            u2 = ops.split_f16x2(u)
            y = torch.empty((a.images, h, w, cout), device=dev)
            tw = []
            for r in range(a.rounds + 1):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                ops.conv3x3_winograd_nhwc_f16x2(x, u2, b, dil, True, out=y, amax_in=am)
                e.record()
                torch.cuda.synchronize()
                if r > 0:
                    tw.append(s.elapsed_time(e))
            msg.append('winograd: %.3f ms' % sorted(tw)[len(tw) // 2])
        print('%s %d->%d %dx%d d%d x%d  ' % (name, cin, cout, h, w, dil, a.images) + '   '.join(msg),
              flush=True)
        if a.stamp and dil == 1:
            nwg = a.images * ((h + 7) // 8) * ((w + 31) // 32) * (cout // 64)
            dbg = torch.zeros((nwg * 4, 8), device=dev, dtype=torch.int64)
            L.set_variant('conv_ring', 4)
            L.set_variant('conv_bn', 0)
            stamp = L.load().naws_debug_conv_stamp_buffer      # only in the AB build
            stamp.argtypes, stamp.restype = [L.p], L.i32
            stamp(dbg.data_ptr())
            ops.conv3x3_nhwc_f16x2(x, w2, b, True, out=outs[0], amax_in=am, dilation=dil)
            torch.cuda.synchronize()
            stamp(None)
            d = dbg[dbg[:, 4] > 0].double()
            tot = d[:, :6].sum(1, keepdim=True)
            sh = (d[:, :6] / tot).mean(0).tolist()
            bn = cout * a.images * ((h + 7) // 8) * ((w + 31) // 32) // (d.shape[0] // 4)
            floor = 9 * cin // 16 * 3 * 2 * (bn // 32) * 32
            print('    stamped build, %d waves, BN %d: wait %.1f %%  barrier %.1f %%  DMA issue %.1f %%  '
                  'LDS reads %.1f %%  MFMA %.1f %%  halo refill %.1f %%   cycles/wave %.0f (MFMA floor %d)'
                  % (d.shape[0], bn, *[100 * v for v in sh], float(tot.mean()), floor), flush=True)
        del x, wt, w2, outs
    L.set_variant('conv_ring', 11)
    L.set_variant('conv_bn', 0)


if __name__ == '__main__':
    main()
