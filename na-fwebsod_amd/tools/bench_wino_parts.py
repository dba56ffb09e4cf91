#!/usr/bin/env python3
"""Time the three stages of the Winograd convolution (input transform, batch-16 GEMM, output
transform) in isolation at the VGG deep-layer shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from naws_hip import ops

dev = torch.device('cuda:0')


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for n, h, w, cin, cout, d in [(2, 150, 250, 256, 256, 1), (1, 150, 250, 256, 256, 1),
                              (2, 75, 125, 512, 512, 1), (1, 75, 125, 512, 512, 1),
                              (2, 74, 124, 512, 512, 2)]:
    hs, ws = (h + d - 1) // d, (w + d - 1) // d
    P = n * d * d * ((hs + 1) // 2) * ((ws + 1) // 2)
    V = torch.empty((16, P, cin), device=dev).uniform_(-1, 1)
    U = torch.empty((16, cout, cin), device=dev).uniform_(-1, 1)
    M = torch.empty((16, P, cout), device=dev)
    x = torch.empty((n, h, w, cin), device=dev).uniform_(-1, 1)
    b = torch.zeros((cout,), device=dev)
    y = torch.empty((n, h, w, cout), device=dev)
    tg = timeit(lambda: ops.gemm(V, U, False, True, out=M))
    tall = timeit(lambda: ops.conv3x3_winograd_nhwc(x, U, b, d, True, out=y))
    fl = 2.0 * P * 16 * cin * cout
    print('n=%d %dx%d %d->%d d%d  P=%d  gemm %.3f ms (%.1f TF)  whole %.3f ms  transforms %.3f ms '
          '(%.0f MB -> %.2f TB/s)' % (n, h, w, cin, cout, d, P, tg, fl / tg / 1e9, tall, tall - tg,
                                      (V.numel() + M.numel() + x.numel() + y.numel()) * 4 / 1e6,
                                      (V.numel() + M.numel() + x.numel() + y.numel()) * 4 / (tall - tg) / 1e9))
