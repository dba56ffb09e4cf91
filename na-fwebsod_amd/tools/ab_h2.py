#!/usr/bin/env python3
"""Interleaved in-process A/B of the fp16x2 GEMM kernel variants (NAWS_H2_VARIANT) on the fc6 / fc7
shapes: the form is selected per call
with naws_set_variant("h2", v), all variants run round-robin in ONE process on the same random operands, and every
variant's result is checked against a float64 product of the split operands on a sample of
output entries.  (Perf deltas between separate runs or boxes are not comparable: DVFS, device
spread - cdna_hip_programming.md rule 24.)

    python tools/ab_h2.py --variants 0 7 [--rounds 9]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from naws_hip import lib as L, ops  # noqa: E402

FN = 'naws_gemm_f32_f16x2_nt'


def call(lib, a, b, out, st):
    a3, b3 = a.planes, b.planes
    mm, k = a3.shape[-2], a3.shape[-3] * 16
    nn = b3.shape[-2]
    batched = a3.dim() == 5
    batch = a3.shape[1] if batched else 1
    c2 = out[0] if batched else out
    rc = getattr(lib, FN)(mm, nn, k, a3.data_ptr(), a3.stride(-3), a3.stride(0),
                          a.inv_scale.data_ptr(), b3.data_ptr(), b3.stride(-3), b3.stride(0),
                          b.inv_scale.data_ptr(), out.data_ptr(), c2.stride(0), batch,
                          (a3.stride(1) if batched else 0), (b3.stride(1) if batched else 0),
                          (out.stride(0) if batched else 0),
                          (a.inv_scale.stride(0) if batched else 0),
                          (b.inv_scale.stride(0) if batched else 0), 0, None, 0, None, 0, 1.0, 0.0,
                          0, 0, st)
    assert rc == 0, rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--variants', type=int, nargs='+', default=[0, 7])
    ap.add_argument('--rounds', type=int, default=9)
    ap.add_argument('--rows', type=int, default=4000)
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    # one library; the form is chosen per call through naws_set_variant("h2", v)
    class _Variant(object):
        def __init__(self, v):
            self.v = v

        def __getattr__(self, name):
            L.set_variant('h2', self.v)
            return getattr(L.load(), name)
    libs = [_Variant(v) for v in a.variants]
    R = a.rows
    kr = (R + 31) // 32 * 32
    cases = [('fc6 fwd  ', (R, 25088), (8192, 25088), None),
             ('fc6 wgrad', (8192, kr), (24576, kr), None),
             ('fc7 fwd  ', (2, R, 4096), (2, 4096, 4096), None),
             ('fc7 wgrad', (2, 4096, kr), (2, 4096, kr), None),
             ('wino conv4_2', (16, 2394, 512), (16, 512, 512), None),
             ('wino conv4_1', (16, 2394, 256), (16, 512, 256), None)]
    g = torch.Generator(device=dev).manual_seed(1)
    for name, sa, sb, _ in cases:
        if a.only and a.only not in name:
            continue
        A = torch.randn(sa, device=dev, generator=g).relu_() if 'fwd' in name else \
            torch.randn(sa, device=dev, generator=g)
        B = torch.randn(sb, device=dev, generator=g) * 0.01
        a2, b2 = ops.split_f16x2(A), ops.split_f16x2(B)
        batched = A.dim() == 3
        m, n, k = sa[-2], sb[-2], sa[-1]
        outs = [torch.empty(((sa[0], m, n) if batched else (m, n)), device=dev) for _ in libs]
        times = [[] for _ in libs]
        for r in range(a.rounds + 1):
            for i, lib in enumerate(libs):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                call(lib, a2, b2, outs[i], st)
                e.record()
                torch.cuda.synchronize()
                if r > 0:
                    times[i].append(s.elapsed_time(e))
        # accuracy on a sample of rows against float64 of the original operands
        A0, B0 = (A[0], B[0]) if batched else (A, B)
        rows = torch.randperm(m, device=dev, generator=g)[:64]
        ref = A0[rows].double() @ B0.double().t()
        mag = A0[rows].double().abs() @ B0.double().abs().t()
        fl = 2.0 * m * n * k * (sa[0] if batched else 1)
        msg = []
        for i, v in enumerate(a.variants):
            o = outs[i][0] if batched else outs[i]
            err = float(((o[rows].double() - ref).abs() / mag).max())
            med = sorted(times[i])[len(times[i]) // 2]
            msg.append('v%d: med %.3f ms %.0f TF (min %.3f) err/|a||b| %.1e' % (
                v, med, fl / med / 1e9, min(times[i]), err))
        print('%s M=%d N=%d K=%d  ' % (name, m, n, k) + '   '.join(msg))
        del A, B, a2, b2, outs


if __name__ == '__main__':
    main()
