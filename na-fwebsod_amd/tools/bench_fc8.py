#!/usr/bin/env python3
"""fc8 forward (logits = H7 W8^T + b: [4000 x 40] from K = 4096, two branches) and fc8 wgrad
(dW8 = dL^T H7: [40 x 4096] from K = 4000): the one-pass fp32-MFMA GEMM against the split-K form,
interleaved in one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from naws_hip import lib as L, ops  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(1)
    rt, c2, n = 4000, 40, 4096
    h7 = torch.randn((rt, 2 * n), device=dev, generator=g)
    h7v = h7.view(rt, 2, n).permute(1, 0, 2)
    w8 = torch.randn((2, c2, n), device=dev, generator=g) * 0.02
    b8 = torch.randn((2, c2), device=dev, generator=g)
    dl = torch.randn((rt, 2 * c2), device=dev, generator=g) * 1e-3
    dlv = dl.view(rt, 2, c2).permute(1, 0, 2)
    lg = torch.empty((rt, 2 * c2), device=dev)
    lgv = lg.view(rt, 2, c2).permute(1, 0, 2)
    gw8 = torch.empty((2, c2, n), device=dev)
    ws = torch.empty((max(rt, n) * c2 * 2 * 16,), device=dev)
    cases = [('fwd one pass', lambda: ops.gemm(h7v, w8, False, True, out=lgv, epilogue=L.EPI_BIAS, bias=b8)),
             ('wgrad one pass', lambda: ops.gemm(dlv, h7v, True, False, out=gw8))]
    for ks in (2, 4, 8, 16):
        cases.append(('fwd split %d' % ks, lambda ks=ks: ops.gemm_splitk(
            h7v, w8, False, True, out=lgv, epilogue=L.EPI_BIAS, bias=b8, ksplit=ks, workspace=ws)))
        cases.append(('wgrad split %d' % ks, lambda ks=ks: ops.gemm_splitk(
            dlv, h7v, True, False, out=gw8, ksplit=ks, workspace=ws)))
    res = {}
    for rnd in range(7):
        for name, fn in cases:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fn()
            e.record()
            torch.cuda.synchronize()
            if rnd:
                res.setdefault(name, []).append(s.elapsed_time(e))
    for name, ts in res.items():
        ts = sorted(ts)
        print('%-16s median %.1f us (min %.1f)' % (name, ts[len(ts) // 2] * 1e3, ts[0] * 1e3))
    print('floor: %.0f us (131 MB of H7 at 4.5 TB/s)' % (h7.numel() * 4 / 4.5e6))


if __name__ == '__main__':
    main()
