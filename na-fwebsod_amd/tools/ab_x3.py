#!/usr/bin/env python3
"""Interleaved in-process A/B of the exact 3 x bf16 GEMM kernel forms (knob "x3") on the fc6 / fc7
shapes of the bench: same operands, round-robin in one process, every form checked against form 0
(and form 0 against a float64 product on a sample of entries).

    python tools/ab_x3.py --variants 0 8 9 [--rounds 7]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from naws_hip import lib as L, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--variants', type=int, nargs='+', default=[0, 8, 9])
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--rows', type=int, default=4000)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(2)
    R = a.rows
    cases = [('fc6 fwd  ', (R, 25088), (8192, 25088)), ('fc6 wgrad', (8192, R), (25088, R)),
             ('fc7 fwd  ', (2, R, 4096), (2, 4096, 4096)), ('fc7 wgrad', (2, 4096, R), (2, 4096, R))]
    for name, sa, sb in cases:
        x = torch.randn(sa, device=dev, generator=g)
        w = torch.randn(sb, device=dev, generator=g)
        a3, b3 = ops.split_bf16x3(x), ops.split_bf16x3(w)
        flops = 2.0 * x.numel() * (sb[-2])
        outs, times = {}, {v: [] for v in a.variants}
        for r in range(a.rounds + 1):
            for v in a.variants:
                L.set_variant('x3', v)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                y = ops.gemm_f32x3_nt(a3, b3)
                e.record()
                torch.cuda.synchronize()
                if r == 0:
                    outs[v] = y.clone()
                else:
                    times[v].append(s.elapsed_time(e))
        L.set_variant('x3', 0)
        ref = outs[a.variants[0]]
        rows = torch.randint(0, ref.shape[-2], (64,), device=dev)
        if x.dim() == 2:
            want = x[rows].double() @ w.double().t()
            got = ref[rows].double()
        else:
            want = torch.einsum('brk,bnk->brn', x[:, rows].double(), w.double())
            got = ref[:, rows].double()
        err = float((got - want).abs().max() / want.abs().max())
        msg = []
        for v in a.variants:
            t = sorted(times[v])[len(times[v]) // 2]
            d = float((outs[v] - ref).abs().max() / ref.abs().max())
            msg.append('x3=%d %.3f ms %.0f TF (diff %.0e)' % (v, t, flops / t / 1e9, d))
        print('%s %s: %s; form %d vs float64 %.1e' % (name, 'x'.join(map(str, sa)), '   '.join(msg),
                                                     a.variants[0], err), flush=True)


if __name__ == '__main__':
    main()
