#!/usr/bin/env python3
"""The bf16 plan's conv chain layer by layer: the wave-private one-plane kernel
(`conv3x3_nhwc_bf16_wp`) against round 1's implicit GEMM (`conv3x3_nhwc_bf16`) on the same input,
and both chains' distance to the fp32 chain - every number relative to the layer's max|y|.
(Round 4: same-input difference 0.3-1.4e-6; chain error 2.5e-3 at conv1_2 .. 8.5e-3 at conv5_3 for
either kernel.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from naws_hip import ops
from naws_hip.engine import VGG16_CONVS
from detectron.datasets import synthetic
dev = torch.device('cuda:0')
blobs = synthetic.init_blobs(20, seed=5)
torch.manual_seed(1)
for (h, w) in ((203, 317), (480, 640)):
    x = torch.rand((1, 3, h, w), device=dev) * 255 - 120
    wp = blobs['conv1_1_w'].to(dev).float().contiguous(); b = blobs['conv1_1_b'].to(dev).float()
    x0 = ops.conv3x3_c3_nchw_to_nhwc(x, wp, b, True)
    xa = xb = xr = x0
    for item in VGG16_CONVS[1:]:
        if item[0] == 'pool':
            xa = ops.maxpool2x2_nhwc(xa, 2); xb = ops.maxpool2x2_nhwc(xb, 2); xr = ops.maxpool2x2_nhwc(xr, 2)
            continue
        if item[0] == 'pool4':
            xa = ops.maxpool2x2_nhwc(xa, 1); xb = ops.maxpool2x2_nhwc(xb, 1); xr = ops.maxpool2x2_nhwc(xr, 1)
            continue
        name, cin, cout, dil = item
        d = 2 if dil is None else dil
        wt = blobs[name + '_w'].to(dev).float().contiguous(); bb = blobs[name + '_b'].to(dev).float()
        pk = ops.conv3x3_pack_weight(wt)
        ya = ops.conv3x3_nhwc_bf16(xa, pk, bb, d, True)                       # old kernel chain
        yb = ops.conv3x3_nhwc_bf16_wp(xb, ops.to_bf16_slab(pk.view(cout, -1)), bb, d, True)   # new kernel chain
        yb_same = ops.conv3x3_nhwc_bf16_wp(xa, ops.to_bf16_slab(pk.view(cout, -1)), bb, d, True)   # new kernel on old chain's input
        yr = ops.conv3x3_nhwc(xr, pk, bb, d, True)                            # fp32 chain
        sc = float(yr.abs().max())
        print('%dx%d %-8s same-input old-vs-new %.2e   chain old-vs-fp32 %.2e  new-vs-fp32 %.2e  (max %.3g)' % (
            h, w, name, float((ya - yb_same).abs().max()) / sc, float((ya - yr).abs().max()) / sc,
            float((yb - yr).abs().max()) / sc, sc))
        xa, xb, xr = ya, yb, yr
